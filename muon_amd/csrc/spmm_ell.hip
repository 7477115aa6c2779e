// SpMM for a NARROW dense block on a sliced-ELL operand (f32, r04):  Y[n x 16] = X[n x d] * Q[d x 16].
//
// MOFA's sparse views multiply by a factor block of <= 16 columns twice per ELBO iteration - A = Y (tau o W) and
// B = Y^T Z, mofapy2's Z / W node updates reached from /root/reference/muon/_core/tools.py:583-585 - and the
// operand does not change for the hundreds of iterations of a fit.  r03's kernel for it (csrc/spmm_narrow.hip)
// walks the row stream with a window protocol per (row, slab) visit - cursor, compare, ballot, count, re-request -
// and r03's own ablation showed that protocol to be the bound: 1.03 of 1.21 ms per product without a single
// gather (profiles/r03_spmm_narrow_b16.txt).  A static operand can be laid out ONCE so that nothing of that is
// left:
//
//   * 16 rows x 4 dense columns fill a wave exactly: quad r (lanes 4r .. 4r+3) owns row r of a group of 16
//     rows, lane c of the quad dense columns 4c .. 4c+3.  One step = ONE stored entry of each of the 16 rows:
//     `ds_read_b128` of the entry's Q row (64 bytes, four lanes) and two `v_pk_fma_f32`, with two DPP quad
//     broadcasts that hand the entry's (LDS offset, value) to its four lanes - the offset's broadcast is the
//     address add itself (`v_add_u32_dpp`);
//   * the operand ("sliced ELL"): the rows in launch order (sorted by length, so the 16 rows of a group are
//     alike), the columns in slabs of 1024 (a Q slab = 64 KiB of LDS, double buffered by LDS-DMA straight from
//     the row-major block); the entries of (group, slab) as steps padded to the longest of the 16 rows, stored
//     in WINDOWS of 4 steps = 384 bytes (64 f32 values, then 64 u16 byte offsets of the Q row inside the slab:
//     6 bytes a slot against the 8 of a CSR entry): slot 4r + j = row r's step 4w + j.  A wave
//     owns one group and reads its windows - slab after slab - as ONE sequential stream, eight windows ahead; a
//     count per (group, slab) says how many.  No cursor, no compare, no ballot;
//   * padded slots are (offset 0, value 0): row 0 of the slab times zero.
// The price is the padding (slots / stored entries = 1.36 on the bench view: Poisson(31) entries per row and
// slab, the longest of 16 rows, rounded up to 4) - bytes the kernel streams instead of instructions it cannot
// issue.  What bounds it: with four waves per SIMD the kernel is ISSUE-bound (an instruction of a 64-lane wave
// occupies its 16-lane SIMD for four cycles), so the loop is written for instruction count: the ring slot is a
// register NAME (eight unrolled copies, left by a counted branch, no dispatch), 18 vector instructions, four LDS
// reads and one global load per window of 64 slots.  Sums: a row's entries are added in stored (column) order,
// slab after slab, in one accumulator: bit-reproducible and independent of the launch shape.
//
// DT = double: MOFA's default precision (tools.py:308 use_float32=False).  The stored values stay f32 (an f64
// matrix is hi + lo, two operands and two launches, the second one accumulating - as for the row stream,
// mu_spmm_stream_f64); Q rows are 128 bytes, so a slab is 512 columns, a lane's four columns two `ds_read_b128`
// and four `v_fma_f64` per entry.
#include <type_traits>
#include <utility>

#include "common.hpp"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"  // 32-bit LDS addresses made from integers

namespace {

constexpr int kESlabBytes = 65536;        // a Q slab: 1024 rows of 16 f32 / 512 rows of 16 f64, double buffered
template <typename DT> constexpr int slab_rows() { return kESlabBytes / (16 * (int)sizeof(DT)); }
constexpr int kEMaxWaves = 15;            // row-owning waves of a workgroup (+ the producer wave)
constexpr int kEDepth = 8;                // windows a wave keeps in flight (3 KiB)
constexpr int kEWin = 384;                // bytes of a window: value[64] f32 | offset[64] u16

typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void e_dma_piece(const void* base, unsigned byte_off, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(byte_off), "s"(base), "s"(lds_dst)
      : "memory");
}

// The window ring: slot D = a[2 D], a[2 D + 1] = (LDS offset, value bits) of this lane's slot.  Loads from asm,
// exact counted waits (hipcc answers a register ring reloaded inside a loop with `s_waitcnt vmcnt(0)`), registers
// the compiler does not know about except through the clobber lists (which make the kernel descriptor count them).
#define MU_EL_CLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
static_assert(kEDepth == 8, "the clobber list names 2 kEDepth registers");
template <int D>
__device__ __forceinline__ void e_request0(const unsigned char* wp, unsigned lane4, unsigned lane2) {  // (prologue)
  asm volatile(
      "global_load_ushort a%c0, %3, %4 offset:%c6\n\t"
      "global_load_dword a%c1, %2, %4 offset:%c5" ::"i"(2 * D),
      "i"(2 * D + 1), "v"(lane4), "v"(lane2), "s"(wp), "i"(kEWin * D), "i"(kEWin * D + 256)
      : MU_EL_CLOB, "memory");
}
// Take slot D (all but the kEDepth - 1 younger windows' two requests each have returned), request window
// `this + kEDepth` into it, and hand every lane of a quad the four (LDS address, value) pairs of its row: lane j of
// the quad holds step j.  (v_accvgpr_read -> DPP read of the same register: two wait states - the loads.)
template <int D>
__device__ __forceinline__ void e_take(const unsigned char* ahead, unsigned lane4, unsigned lane2, unsigned base,
                                       unsigned (&adr)[4], float (&val)[4]) {
  unsigned off;
  float v;
  asm volatile(
      "s_waitcnt vmcnt(%c[n])\n\t"
      "v_accvgpr_read_b32 %[off], a%c[r0]\n\t"
      "v_accvgpr_read_b32 %[v], a%c[r1]\n\t"
      "global_load_ushort a%c[r0], %[lane2], %[ahead] offset:256\n\t"
      "global_load_dword a%c[r1], %[lane4], %[ahead]\n\t"
      "v_add_u32_dpp %[a0], %[off], %[base] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_u32_dpp %[a1], %[off], %[base] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_u32_dpp %[a2], %[off], %[base] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_u32_dpp %[a3], %[off], %[base] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %[v0], %[v] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %[v1], %[v] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %[v2], %[v] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %[v3], %[v] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
      : [off] "=&v"(off), [v] "=&v"(v), [a0] "=&v"(adr[0]), [a1] "=&v"(adr[1]), [a2] "=&v"(adr[2]), [a3] "=&v"(adr[3]),
        [v0] "=&v"(val[0]), [v1] "=&v"(val[1]), [v2] "=&v"(val[2]), [v3] "=&v"(val[3])
      : [base] "v"(base), [lane4] "v"(lane4), [lane2] "v"(lane2), [ahead] "s"(ahead), [n] "i"(2 * (kEDepth - 1)),
        [r0] "i"(2 * D), [r1] "i"(2 * D + 1)
      : MU_EL_CLOB, "memory");
}

// Wave 0 of the workgroup is the PRODUCER of the Q slabs and owns no rows.  The vector-memory counter retires in
// order: a wave that issues slab pieces cannot take a window requested after them before the pieces have landed -
// one full memory latency per slab and wave, at the same moment in all waves (they leave the barrier together).
// With one wave doing nothing but `issue the next slab, wait, barrier`, the others never wait for anything but
// their own windows.   MODE (timing ablations, wrong results): 1 no gathers / FMAs, 2 no slab copies.
template <int MODE, typename DT>
__global__ __launch_bounds__(64 * (kEMaxWaves + 1)) void k_spmm_ell16(int64_t n_pos, int64_t n_cols, int n_slabs, int cw,
                                                                     const int32_t* __restrict__ hdr,
                                                                     const int64_t* __restrict__ wave_base,
                                                                     const unsigned char* __restrict__ ent,
                                                                     const int32_t* __restrict__ perm,
                                                                     const DT* __restrict__ Q, DT* __restrict__ Y,
                                                                     int accumulate, int s_per_part, int64_t y_stride) {
  // r06 - COLUMN PARTS (blockIdx.y): workgroup (x, y) sweeps slabs [y s_per_part, (y + 1) s_per_part) only and writes its
  // partial product to Y + y y_stride.  A shard of a few thousand rows is a few dozen workgroups, and every one of them
  // pulls ALL of Q through its LDS at the producer wave's ~25 GB/s: the 12 500-cell shard of configs[4] took 0.25 ms
  // for a product whose share of the full-size launch is 0.07.  (One part: the launch of before.)
  constexpr int kRowBytes = 16 * (int)sizeof(DT);
  const int s_begin = (int)blockIdx.y * s_per_part;
  const int s_end = (s_begin + s_per_part) < n_slabs ? (s_begin + s_per_part) : n_slabs;
  Y += (int64_t)blockIdx.y * y_stride;
  __shared__ __attribute__((aligned(1024))) unsigned char slab[2 * kESlabBytes];
  const int lane = threadIdx.x & 63;
  const int wave = uniform32(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&slab[0]);

  if (wave == 0) {  // 64 pieces of 1 KiB per slab
    const unsigned q_last = (unsigned)(n_cols * kRowBytes - 16);  // (the last slab is clamped to the end of Q)
    auto whole = [&](int s, int b) {
#pragma unroll 8
      for (int piece = 0; piece < ((MODE & 2) ? 1 : kESlabBytes / 1024); ++piece) {
        unsigned o = (unsigned)s * (unsigned)kESlabBytes + (unsigned)(piece * 1024 + lane * 16);
        o = o < q_last ? o : q_last;  // past the end: never consumed
        e_dma_piece(Q, o, lds0 + (unsigned)b * (unsigned)kESlabBytes + (unsigned)piece * 1024u);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    whole(s_begin, 0);
    __syncthreads();
    for (int s = s_begin; s + 1 < s_end; ++s) {
      whole(s + 1, (s + 1 - s_begin) & 1);  // (its buffer held slab s - 1: everyone is past the barrier that ended it)
      __syncthreads();                      // the end of slab s for the others
    }
    return;
  }

  // this wave's group of 16 positions.  The groups are in order of row length: dealt round the workgroups, so that
  // the workgroups - whole rounds of them, one per CU - carry alike shares
  const int64_t gwave = (int64_t)blockIdx.x + (int64_t)gridDim.x * (wave - 1);
  const int64_t n_waves = (n_pos + 15) / 16;
  (void)cw;
  const bool active = gwave < n_waves;
  const unsigned lane4 = (unsigned)lane * 4u, lane2 = (unsigned)lane * 2u;

  DT acc[4] = {(DT)0, (DT)0, (DT)0, (DT)0};
  asm volatile("" ::: MU_EL_CLOB);

  typedef __attribute__((address_space(4))) const int32_t* chdr_p;
  const chdr_p myhdr = (chdr_p)(hdr + (active ? gwave : 0) * (int64_t)n_slabs);
  auto counts_of = [&](int s) -> int { return active ? uniform32(myhdr[s]) : 0; };
  // (a later part starts behind the windows of the slabs before it: a wave-wide sum of its counts)
  int64_t skip = 0;
  if (s_begin > 0 && active) {
    int part = 0;
    for (int i = lane; i < s_begin; i += 64) part += hdr[gwave * (int64_t)n_slabs + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    skip = (int64_t)uniform32(part);
  }
  const unsigned char* wp = ent + (uniform64(active ? wave_base[gwave] : 0) + skip) * kEWin + kEWin * kEDepth;  // next REQUEST

  const unsigned lane_c = (unsigned)(lane & 3) * (unsigned)(4 * sizeof(DT));
  unsigned base = lds0 + lane_c;  // this lane's four columns of the current slab buffer
  // one window: 4 steps of 16 entries
  auto window = [&](auto dc) {
    constexpr int D = decltype(dc)::value;
    unsigned adr[4];
    float val[4];
    e_take<D>(wp, lane4, lane2, base, adr, val);
    wp += kEWin;
    if constexpr (MODE & 1) {
      acc[0] += (DT)(val[0] + val[1] + val[2] + val[3] + (float)(adr[0] ^ adr[1] ^ adr[2] ^ adr[3]));
    } else if constexpr (sizeof(DT) == 4) {
      typedef __attribute__((address_space(3))) const f4* lds_p;
      const f4 q0 = *(lds_p)adr[0];
      const f4 q1 = *(lds_p)adr[1];
      const f4 q2 = *(lds_p)adr[2];
      const f4 q3 = *(lds_p)adr[3];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = fmaf(val[0], q0[c], acc[c]);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = fmaf(val[1], q1[c], acc[c]);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = fmaf(val[2], q2[c], acc[c]);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = fmaf(val[3], q3[c], acc[c]);
    } else {
      typedef __attribute__((address_space(3))) const d2* lds_p;
      d2 q[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q[j][0] = *(lds_p)adr[j];
        q[j][1] = *(lds_p)(adr[j] + 16u);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double v = (double)val[j];
        acc[0] = fma(v, q[j][0][0], acc[0]);
        acc[1] = fma(v, q[j][0][1], acc[1]);
        acc[2] = fma(v, q[j][1][0], acc[2]);
        acc[3] = fma(v, q[j][1][1], acc[3]);
      }
    }
  };

  // prologue: the first kEDepth windows are requested while slab 0 lands
  {
    const unsigned char* w0 = wp - kEWin * kEDepth;
    e_request0<0>(w0, lane4, lane2);
    e_request0<1>(w0, lane4, lane2);
    e_request0<2>(w0, lane4, lane2);
    e_request0<3>(w0, lane4, lane2);
    e_request0<4>(w0, lane4, lane2);
    e_request0<5>(w0, lane4, lane2);
    e_request0<6>(w0, lane4, lane2);
    e_request0<7>(w0, lane4, lane2);
  }
  int s = s_begin;
  int left = s_begin < s_end ? counts_of(s_begin) : 0;  // windows of this wave in slab s still to take
  int next_cnt = s_begin + 1 < s_end ? counts_of(s_begin + 1) : 0;
  __syncthreads();
  // to the next slab that holds windows of this wave (a barrier per slab boundary); false: no slab is left
  auto advance = [&]() -> bool {
    do {
      if (s + 1 >= s_end) return false;
      __syncthreads();  // through with slab s; the producer's slab s + 1 has landed
      ++s;
      base = lds0 + lane_c + (unsigned)((s - s_begin) & 1) * (unsigned)kESlabBytes;
      left = next_cnt;
      next_cnt = s + 1 < s_end ? counts_of(s + 1) : 0;
    } while (left == 0);
    return true;
  };
  bool more = left > 0 || advance();
  // the ring slot of a window is a register name: eight copies of the body, left by a counted branch
#define MU_STEP(D)                                   \
  window(std::integral_constant<int, D>{});          \
  if (--left == 0) {                                 \
    if (!advance()) break;                           \
  }
  if (more) {
    for (;;) {
      MU_STEP(0)
      MU_STEP(1)
      MU_STEP(2)
      MU_STEP(3)
      MU_STEP(4)
      MU_STEP(5)
      MU_STEP(6)
      MU_STEP(7)
    }
  }
#undef MU_STEP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requests still in flight target the ring registers

  const int r = lane >> 2, c = lane & 3;
  const int64_t p = gwave * 16 + r;
  if (active && p < n_pos) {
    const int64_t row = perm ? (int64_t)perm[p] : p;
    if (row >= 0) {
      DT* y = Y + row * 16 + 4 * c;
#pragma unroll
      for (int u = 0; u < 4; ++u) y[u] = accumulate ? y[u] + acc[u] : acc[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// the layout itself (r05): CSR + its slab pointers -> windows.  A team of 16 lanes = one (group, slab): lane l is the
// group's l-th position; per window it reads the next four entries of its row inside the slab (or padding) and writes
// its 16 bytes of values and 8 bytes of offsets - the team's stores are the window's 256 + 128 contiguous bytes, and
// every slot of every window is written (no memset of the padded operand).  r04 laid the operand out with a dozen
// tensor passes over every entry (int64 ranks, destinations, two scatters): 30 ms per operand at 3.1e8 entries, the
// larger part of a fit's set-up.
// ---------------------------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) U4 { int32_t v[4]; };
struct __attribute__((packed, aligned(4))) F4 { float v[4]; };

__global__ __launch_bounds__(256) void k_ell16_fill(int64_t n_groups, int64_t S, int64_t nnz, int32_t col_mask, int row_shift,
                                                    const int32_t* __restrict__ indices,
                                                    const float* __restrict__ values,
                                                    const int64_t* __restrict__ sp, const int32_t* __restrict__ perm,
                                                    const int32_t* __restrict__ hdr,
                                                    const int64_t* __restrict__ win_base,
                                                    unsigned char* __restrict__ ent) {
  const int l = threadIdx.x & 15;
  const int64_t teams = n_groups * S;
  for (int64_t team = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; team < teams;
       team += ((int64_t)gridDim.x * blockDim.x) >> 4) {
    const int64_t g = team / S, sl = team - g * S;
    const int32_t row = perm[16 * g + l];
    int64_t lo = 0, hi = 0;
    if (row >= 0) {
      lo = sp[(int64_t)row * (S + 1) + sl];
      hi = sp[(int64_t)row * (S + 1) + sl + 1];
    }
    const int32_t nw = hdr[team];
    unsigned char* out = ent + win_base[team] * 384;
    for (int32_t w = 0; w < nw; ++w, lo += 4, out += 384) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      unsigned short o[4] = {0, 0, 0, 0};
      if (lo < hi) {
        int32_t ci[4];
        float cv[4];
        if (lo + 4 <= nnz) {  // four entries in two loads (4-byte aligned: fine for global memory); what lies behind
          const U4 a = *reinterpret_cast<const U4*>(indices + lo);  // the row's end in them is masked below
          const F4 b = *reinterpret_cast<const F4*>(values + lo);
#pragma unroll
          for (int j = 0; j < 4; ++j) ci[j] = a.v[j], cv[j] = b.v[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool in = lo + j < hi;
            ci[j] = in ? indices[lo + j] : 0;
            cv[j] = in ? values[lo + j] : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (lo + j < hi) {
            v[j] = cv[j];
            o[j] = (unsigned short)((ci[j] & col_mask) << row_shift);
          }
      }
      *reinterpret_cast<float4*>(out + 16 * l) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<uint2*>(out + 256 + 8 * l) =
          make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
    }
  }
}

}  // namespace

extern "C" {

int mu_spmm_ell16_waves(int64_t n_rows) {
  // row-owning waves (= groups of 16 rows) per workgroup.  The chip runs ONE workgroup per CU (128 KiB of Q
  // slabs) and every workgroup sweeps all of Q.  One round of workgroups while the groups fit; beyond that 14
  // waves (measured at 100k rows on 256 CUs, DESIGN.md 6: 13 / 14 / 15 waves = 0.68 / 0.64 / 0.65 ms - the
  // groups are dealt round the workgroups in order of length, so a ragged last round costs little)
  const int64_t cus = mu_num_cus();
  const int64_t nw = (n_rows + 15) / 16;
  if (nw <= cus) return 1;
  if (nw <= cus * kEMaxWaves) return (int)((nw + cus - 1) / cus);
  return 14;
}

static int ell16_launch(bool wide, int waves, int64_t n_pos, int64_t n_cols, const int32_t* d_hdr,
                        const int64_t* d_wave_base, const void* d_ent, const int32_t* d_perm, const void* d_Q, void* d_Y,
                        int accumulate, void* stream, int parts = 1, int64_t y_stride = 0) {
  MU_REQUIRE(waves >= 1 && waves <= kEMaxWaves, "row-owning waves per workgroup: 1 .. 15");
  const int64_t row_bytes = wide ? 128 : 64;
  MU_REQUIRE(n_pos >= 0 && n_cols > 0 && n_cols * row_bytes < ((int64_t)1 << 32), "shape out of range");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_hdr && d_wave_base && d_ent && d_Q && d_Y, "null pointer");
  const int64_t slab = kESlabBytes / row_bytes;
  const int64_t n_slabs = (n_cols + slab - 1) / slab;
  const int64_t n_waves = (n_pos + 15) / 16;
  const int64_t wgs = (n_waves + waves - 1) / waves;
  const int mode = mu_tune_get("ell_mode") & 3;
  hipStream_t st = (hipStream_t)stream;
  MU_REQUIRE(parts >= 1 && parts <= n_slabs && (parts == 1 || !accumulate), "column parts: 1 .. slabs, partial products");
  const int s_per_part = (int)((n_slabs + parts - 1) / parts);
  const unsigned ny = (unsigned)((n_slabs + s_per_part - 1) / s_per_part);
#define MU_GO(MD, DT)                                                                                                  \
  hipLaunchKernelGGL((k_spmm_ell16<MD, DT>), dim3((unsigned)wgs, ny), dim3(64 * (waves + 1)), 0, st, n_pos, n_cols,     \
                     (int)n_slabs, waves, d_hdr, d_wave_base, (const unsigned char*)d_ent, d_perm, (const DT*)d_Q,       \
                     (DT*)d_Y, accumulate, s_per_part, y_stride)
  if (wide) {
    if (mode == 1) MU_GO(1, double);
    else if (mode == 2) MU_GO(2, double);
    else if (mode == 3) MU_GO(3, double);
    else MU_GO(0, double);
  } else {
    if (mode == 1) MU_GO(1, float);
    else if (mode == 2) MU_GO(2, float);
    else if (mode == 3) MU_GO(3, float);
    else MU_GO(0, float);
  }
#undef MU_GO
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_spmm_ell16_f32(int waves, int64_t n_pos, int64_t n_cols, const int32_t* d_hdr, const int64_t* d_wave_base,
                      const void* d_ent, const int32_t* d_perm, const float* d_Q, float* d_Y, void* stream) {
  return ell16_launch(false, waves, n_pos, n_cols, d_hdr, d_wave_base, d_ent, d_perm, d_Q, d_Y, 0, stream);
}

int mu_spmm_ell16_f64(int waves, int64_t n_pos, int64_t n_cols, const int32_t* d_hdr, const int64_t* d_wave_base,
                      const void* d_ent, const int32_t* d_perm, const double* d_Q, double* d_Y, int accumulate,
                      void* stream) {
  return ell16_launch(true, waves, n_pos, n_cols, d_hdr, d_wave_base, d_ent, d_perm, d_Q, d_Y, accumulate, stream);
}

/* r06 - the same products with the COLUMN SLABS SPLIT into `parts` (blockIdx.y): part y writes its partial product to
 * d_Y + y * y_stride (elements); the caller sums the ceil(slabs / ceil(slabs / parts)) partials in order.  For operands
 * of a few thousand rows (one rank's shard of a sharded fit): see mu_spmm_ell16_parts. */
int mu_spmm_ell16_parts_f32(int waves, int parts, int64_t n_pos, int64_t n_cols, const int32_t* d_hdr,
                            const int64_t* d_wave_base, const void* d_ent, const int32_t* d_perm, const float* d_Q,
                            float* d_Y, int64_t y_stride, void* stream) {
  return ell16_launch(false, waves, n_pos, n_cols, d_hdr, d_wave_base, d_ent, d_perm, d_Q, d_Y, 0, stream, parts, y_stride);
}
int mu_spmm_ell16_parts_f64(int waves, int parts, int64_t n_pos, int64_t n_cols, const int32_t* d_hdr,
                            const int64_t* d_wave_base, const void* d_ent, const int32_t* d_perm, const double* d_Q,
                            double* d_Y, int64_t y_stride, void* stream) {
  return ell16_launch(true, waves, n_pos, n_cols, d_hdr, d_wave_base, d_ent, d_perm, d_Q, d_Y, 0, stream, parts, y_stride);
}
/* (waves, parts) for n_rows x n_cols: one part - the launch of mu_spmm_ell16_waves - unless the row groups fill less
 * than a round of workgroups at full width; then 15 waves per workgroup and enough column parts for two rounds, each
 * part at least four slabs */
int mu_spmm_ell16_parts(int64_t n_rows, int64_t n_cols, int wide, int* waves, int* parts) {
  MU_REQUIRE(waves && parts && n_rows >= 0 && n_cols > 0, "bad arguments");
  const int64_t cus = mu_num_cus();
  const int64_t nw = (n_rows + 15) / 16;
  const int64_t slabs = (n_cols + (wide ? 512 : 1024) - 1) / (wide ? 512 : 1024);
  *waves = mu_spmm_ell16_waves(n_rows);
  *parts = 1;
  const int64_t wgs = (nw + kEMaxWaves - 1) / kEMaxWaves;
  if (nw > 0 && 2 * wgs <= cus && slabs >= 8) {
    int64_t p = (2 * cus + wgs - 1) / wgs;
    if (p > slabs / 4) p = slabs / 4;
    if (p >= 2) {
      *waves = kEMaxWaves;
      *parts = (int)p;
    }
  }
  return MU_OK;
}

int mu_ell16_fill(int64_t n_groups, int64_t n_cols, int64_t nnz, int slab_cols, const int32_t* d_indices, const float* d_values,
                  const int64_t* d_slab_ptr, const int32_t* d_perm, const int32_t* d_hdr, const int64_t* d_win_base,
                  void* d_ent, void* stream) {
  if (slab_cols != 1024 && slab_cols != 512) return MU_ERR_ARG;
  if (n_groups < 0 || n_cols < 0) return MU_ERR_ARG;
  const int64_t S = (n_cols + slab_cols - 1) / slab_cols;
  const int64_t teams = n_groups * S;
  if (teams == 0) return MU_OK;
  if (!d_indices || !d_values || !d_slab_ptr || !d_perm || !d_hdr || !d_win_base || !d_ent) return MU_ERR_ARG;
  int64_t blocks = (teams + 15) / 16;
  const int64_t cap = (int64_t)mu_num_cus() * 64;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_ell16_fill, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_groups, S, nnz,
                     (int32_t)(slab_cols - 1), slab_cols == 1024 ? 6 : 7, d_indices, d_values, d_slab_ptr, d_perm, d_hdr,
                     d_win_base, static_cast<unsigned char*>(d_ent));
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
