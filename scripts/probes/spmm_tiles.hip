// ARCHIVED EXPERIMENT (r03): not compiled into libmuon_amd.so.  Built, wired through the C-ABI (mu_csr_tiles_build,
// mu_spmm_tiles_f32), bit-identical to csrc/spmm_narrow.hip on every test matrix - and NOT faster: 1.245 / 1.385 ms
// against 1.187 / 1.294 ms per product at 100 000 x 100 000 (B = 16), for 3.7 GB of operand instead of 2.5 GB.
// ~40 % fewer instructions per (row, slab) visit did not move the time: DESIGN.md 6 (f).
//
// SpMM for a narrow dense block (f32, B = 16) on a PRE-CUT operand: the "step stream".
//
// The narrow-block kernel (csrc/spmm_narrow.hip) spends its time issuing instructions: ~650 dependent-ish
// instructions per wave and slab at four waves per SIMD, of which the gathers and FMAs - the product itself - are
// a third.  The rest finds out, at every (row, slab) visit, which entries of the row belong to the slab (a
// compare, a ballot, a count, a mask), where the next window starts, and turns columns into LDS offsets.  All
// of that depends on the matrix and the layout only.  MOFA multiplies the same sparse view 200 times per fit
// (/root/reference/muon/_core/tools.py:585 -> ent.run(): two products per iteration), so the operand is cut
// ONCE (mu_csr_tiles_build) into the order and the form the kernel consumes:
//
//   * per (workgroup, wave, sweep): the entries of the sweep's rows in (slab, row, column) order, every
//     (row, slab) run padded to a multiple of 16 entries with (offset 0, value 0) pairs - a "step" is 16 entries,
//     one gather instruction group;
//   * an entry is (LDS byte offset of its Q row inside the slab, swizzle bits included; value): 8 bytes as before;
//   * steps[wave][sweep][slab][row] (one byte) says how many steps a (row, slab) run has: the kernel's cursor is
//     a running sum, known a slab ahead, and the window request needs nothing from the data.
// The kernel per visit: request the row's next run (address from the step table), spread the 64 loaded
// entries over their four lanes (lane swaps, as in spmm_narrow.hip), gather, FMA.  No compare, no ballot, no
// mask, no address arithmetic on columns.  Same Q slabs (1024 columns, XOR-swizzled, LDS-DMA, double
// buffered), same lane layout, same fixed summation order per row (padding adds exact zeros).
//
// Upper bound of the stream: every run grows by at most 15 entries: run (position p0 of the sweep's first
// row) starts at sptr[p0] + 15 * n_slabs * p0 - no count pass, no pointer array; the gaps are never read.
#include <utility>

#include "common.hpp"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"

namespace {

constexpr int kTSlab = 1024;
constexpr int kTSlabBytes = kTSlab * 64;
constexpr int kTW = 16;

typedef float f4 __attribute__((ext_vector_type(4)));

template <int... I, class F>
__device__ __forceinline__ void t_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void t_static_for(F&& f) {
  t_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

__device__ __forceinline__ void t_dma_piece(const void* base, unsigned byte_off, unsigned lds_dst) {
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

struct TRows4 { unsigned x[4]; };
__device__ __forceinline__ TRows4 t_rows4(unsigned v) {  // x[w] in lane 16 c + e = the input's lane 16 w + e
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  const auto a = __builtin_amdgcn_permlane32_swap(r[0], r[0], false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(r[1], r[1], false, false);
  TRows4 o;
  o.x[0] = a[0];
  o.x[1] = b[0];
  o.x[2] = a[1];
  o.x[3] = b[1];
  return o;
}
template <int CTRL>
__device__ __forceinline__ int t_dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float t_dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ int64_t readlane_i64_t(int64_t v, int l) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return ((int64_t)hi << 32) | (int64_t)(uint32_t)lo;
}

// where the run of (position p0, the first row of a wave's sweep) starts in the step stream
__device__ __forceinline__ int64_t t_run_base(const int64_t* sptr, int64_t p0, int64_t n_pos, int64_t n_slabs) {
  const int64_t p = p0 < n_pos ? p0 : n_pos;
  return sptr[p] + 15 * n_slabs * p;
}

// ---- build: row stream -> step stream (once per operand) ---------------------------------------------------
template <int RW>
__global__ __launch_bounds__(1024) void k_tiles_build(int64_t n_pos, int64_t n_cols, int K,
                                                      const int64_t* __restrict__ sptr,
                                                      const unsigned long long* __restrict__ ent,
                                                      unsigned long long* __restrict__ tent,
                                                      unsigned char* __restrict__ tsteps) {
  const int lane = threadIdx.x & 63;
  const int wave = uniform32(threadIdx.x >> 6);
  const int rows_w = 4 * K;
  const int n_sweeps = (rows_w + RW - 1) / RW;
  const int64_t n_slabs = (n_cols + kTSlab - 1) / kTSlab;
  const int64_t rb0 = (int64_t)blockIdx.x * (64 * (int64_t)K);
  const int64_t rb1 = (rb0 + 64 * (int64_t)K) < n_pos ? (rb0 + 64 * (int64_t)K) : n_pos;
  const int64_t pw0 = rb0 + (int64_t)wave * rows_w;
  for (int q = 0; q < n_sweeps; ++q) {
    const int64_t p0 = pw0 + (int64_t)q * RW;
    int64_t out = t_run_base(sptr, p0, n_pos, n_slabs);  // uniform
    unsigned char* st = tsteps + ((((int64_t)blockIdx.x * kTW + wave) * n_sweeps + q) * n_slabs) * 16;
    // lane r: cursor / end of row r of this sweep
    int64_t cur = 0, end = 0;
    if (lane < RW && q * RW + lane < rows_w && p0 + lane < rb1) {
      cur = sptr[p0 + lane];
      end = sptr[p0 + lane + 1];
    }
    for (int64_t s = 0; s < n_slabs; ++s) {
      const int s0 = (int)(s * kTSlab);
      const int s_hi = (s0 + kTSlab) < (int)n_cols ? (s0 + kTSlab) : (int)n_cols;
      int my_steps = 0;  // lane r: steps of row r in this slab
      for (int r = 0; r < RW; ++r) {  // uniform
        int64_t c = readlane_i64_t(cur, r);
        const int64_t e = readlane_i64_t(end, r);
        int steps = 0;
        while (true) {  // chunks of 64 entries of the row inside this slab
          const int64_t left = e - c;
          unsigned long long pair = 0x000000007fffffffull;
          if ((int64_t)lane < left) pair = ent[c + lane];
          const int col = (int)(unsigned)pair;
          const bool in = ((int64_t)lane < left) && col < s_hi;
          const int n = __popcll(__ballot(in));
          if (n == 0) break;
          const int ns = (n + 15) >> 4;
          if (lane < ns * 16) {
            const unsigned key = in ? ((((unsigned)(col - s0)) << 6) | (((unsigned)col & 12u) << 2)) : 0u;
            const unsigned long long o = (unsigned long long)key | (in ? (pair & 0xffffffff00000000ull) : 0ull);
            tent[out + lane] = o;
          }
          out += ns * 16;
          steps += ns;
          c += n;
          if (n < 64) break;
        }
        if (lane == r) {
          cur = c;
          my_steps = steps;
        }
      }
      if (lane < 16) st[s * 16 + lane] = (unsigned char)(lane < RW ? my_steps : 0);
    }
  }
}

// ---- the product ---------------------------------------------------------------------------------------------
// NODMA: timing ablation (wrong results): the Q slabs are not copied
template <int RW, bool NODMA = false>
__global__ __launch_bounds__(1024) void k_spmm_tiles(int64_t n_pos, int64_t n_cols, int K,
                                                     const int64_t* __restrict__ sptr,
                                                     const unsigned long long* __restrict__ tent,
                                                     const unsigned char* __restrict__ tsteps,
                                                     const int32_t* __restrict__ perm, const float* __restrict__ Q,
                                                     float* __restrict__ Y) {
  static_assert(RW >= 4 && RW <= 16, "a sweep covers 4 .. 16 rows per wave");
  typedef __attribute__((address_space(3))) const f4* lds_p;
  __shared__ f4 qs[2][kTSlabBytes / 16];
  const int lane = threadIdx.x & 63;
  const int wave = uniform32(threadIdx.x >> 6);
  const int c = lane >> 4;
  const int rows_w = 4 * K;
  const int n_sweeps = (rows_w + RW - 1) / RW;
  const int n_slabs = (int)((n_cols + kTSlab - 1) / kTSlab);
  const int64_t rb0 = (int64_t)blockIdx.x * (64 * (int64_t)K);
  const int64_t rb1 = (rb0 + 64 * (int64_t)K) < n_pos ? (rb0 + 64 * (int64_t)K) : n_pos;
  const int64_t pw0 = rb0 + (int64_t)wave * rows_w;
  const unsigned qs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&qs[0][0]);
  if ((qs_lds & 0xffffu) != 0u) __builtin_trap();  // the XOR addressing needs the buffers 64 KiB aligned
  const unsigned q_last = (unsigned)(n_cols * 64 - 16);
  const unsigned lane8 = (unsigned)lane * 8u;

  auto dma_one = [&](int s0, int buf, int u) {  // as in spmm_narrow.hip: XOR-swizzled quads
    if constexpr (NODMA) return;
    const int piece = wave + u * kTW;
    const int j = piece * 16 + (lane >> 2);
    const int cq = (lane & 3) ^ ((j >> 2) & 3);
    unsigned off = (unsigned)s0 * 64u + (unsigned)(j * 64 + cq * 16);
    off = off < q_last ? off : q_last;
    t_dma_piece(Q, off, qs_lds + (unsigned)buf * (unsigned)kTSlabBytes + (unsigned)piece * 1024u);
  };
  // steps of the RW rows in one slab: lane r holds row r; returns the exclusive prefix, sets the total
  auto scan = [&](int st, int& tot) -> int {
    int x = st;
    x += t_dpp_i<0x111>(x);  // row_shr:1 (zero fill)
    x += t_dpp_i<0x112>(x);
    x += t_dpp_i<0x114>(x);
    x += t_dpp_i<0x118>(x);
    tot = __builtin_amdgcn_readlane(x, 15);
    return x - st;
  };

  for (int q = 0; q < n_sweeps; ++q) {  // uniform over the workgroup
    const int r0 = q * RW;
    const int64_t p0 = pw0 + r0;
    const char* __restrict__ runb = reinterpret_cast<const char*>(tent + t_run_base(sptr, p0, n_pos, n_slabs));
    const unsigned char* __restrict__ stp = tsteps + ((((int64_t)blockIdx.x * kTW + wave) * n_sweeps + q) * n_slabs) * 16;
    auto request = [&](unsigned first, unsigned& key, float& val) {  // lane l: entry first + l of this sweep's run
      const unsigned long long e = *reinterpret_cast<const unsigned long long*>(runb + (size_t)(first * 8u + lane8));
      key = (unsigned)e;
      val = __builtin_bit_cast(float, (unsigned)(e >> 32));
    };
    f4 acc[RW];
    unsigned wkey[RW];
    float wval[RW];
    int st = (lane < 16) ? (int)stp[lane] : 0, tot = 0;
    int pre = scan(st, tot);
    unsigned off = 0;  // entries of this sweep's run before the current slab
    t_static_for<RW>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      acc[r] = (f4)(0.f);
      request(16u * (unsigned)__builtin_amdgcn_readlane(pre, r), wkey[r], wval[r]);
    });
#pragma unroll
    for (int u = 0; u < 4; ++u) dma_one(0, 0, u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int buf = 0;
    for (int s = 0; s < n_slabs; ++s, buf ^= 1) {
      const int s0 = s * kTSlab;
      const unsigned qx = qs_lds + (unsigned)buf * (unsigned)kTSlabBytes + (unsigned)c * 16u;
      // the next slab's step counts (the last slab reads its own again: never used)
      const int sn = s + 1 < n_slabs ? s + 1 : s;
      const int stn = (lane < 16) ? (int)stp[sn * 16 + lane] : 0;
      int totn = 0;
      const int pren = scan(stn, totn);
      const unsigned offn = off + 16u * (unsigned)tot;
      t_static_for<RW>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r < 4) dma_one(s0 + kTSlab, buf ^ 1, r);
        const int steps = __builtin_amdgcn_readlane(st, r);
        const unsigned key = wkey[r];
        const float val = wval[r];
        // the row's run of the next slab: its address needs nothing from the data
        request(offn + 16u * (unsigned)__builtin_amdgcn_readlane(pren, r), wkey[r], wval[r]);
        if (steps > 0) {  // uniform
          const TRows4 A = t_rows4(key);
          const TRows4 V = t_rows4(__builtin_bit_cast(unsigned, val));
          // exactly `steps` gathers (the entries behind a run belong to the next one), two at a time
          if (steps >= 2) {
            const f4 q0 = *(lds_p)(A.x[0] ^ qx);
            const f4 q1 = *(lds_p)(A.x[1] ^ qx);
            acc[r] += __builtin_bit_cast(float, V.x[0]) * q0;
            acc[r] += __builtin_bit_cast(float, V.x[1]) * q1;
            asm volatile("" : "+v"(acc[r]));
          } else {
            const f4 q0 = *(lds_p)(A.x[0] ^ qx);
            acc[r] += __builtin_bit_cast(float, V.x[0]) * q0;
            asm volatile("" : "+v"(acc[r]));
          }
          if (steps >= 4) {
            const f4 q2 = *(lds_p)(A.x[2] ^ qx);
            const f4 q3 = *(lds_p)(A.x[3] ^ qx);
            acc[r] += __builtin_bit_cast(float, V.x[2]) * q2;
            acc[r] += __builtin_bit_cast(float, V.x[3]) * q3;
            asm volatile("" : "+v"(acc[r]));
          } else if (steps == 3) {
            const f4 q2 = *(lds_p)(A.x[2] ^ qx);
            acc[r] += __builtin_bit_cast(float, V.x[2]) * q2;
            asm volatile("" : "+v"(acc[r]));
          }
          if (steps > 4) {  // a run of more than 64 entries (rare): the chunks behind the window, one by one
            const unsigned first = off + 16u * (unsigned)__builtin_amdgcn_readlane(pre, r);
            for (int done = 4; done < steps; done += 4) {
              unsigned k2;
              float v2;
              request(first + 16u * (unsigned)done, k2, v2);
              const TRows4 A2 = t_rows4(k2);
              const TRows4 V2 = t_rows4(__builtin_bit_cast(unsigned, v2));
#pragma unroll
              for (int w = 0; w < 4; ++w)
                if (done + w < steps) {
                  const f4 qq = *(lds_p)(A2.x[w] ^ qx);
                  acc[r] += __builtin_bit_cast(float, V2.x[w]) * qq;
                }
              asm volatile("" : "+v"(acc[r]));
            }
          }
        }
      });
      off = offn;
      st = stn;
      pre = pren;
      tot = totn;
      // the DMA pieces are older than the run requests of rows 3 .. RW-1 (and of every extra chunk)
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(RW - 3) : "memory");
      __syncthreads();
    }
    // slot partial sums -> row sums (rotation tree inside every 16-lane row), lanes e == 0 store
    t_static_for<RW>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      f4 v = acc[r];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float x = v[k];
        x += t_dpp_f<0x128>(x);
        x += t_dpp_f<0x124>(x);
        x += t_dpp_f<0x122>(x);
        x += t_dpp_f<0x121>(x);
        v[k] = x;
      }
      const int64_t p = p0 + r;
      if (r0 + r < rows_w && p < rb1 && (lane & 15) == 0) {
        const int64_t out = perm ? (int64_t)perm[p] : p;
        if (out >= 0) *reinterpret_cast<f4*>(Y + out * 16 + 4 * c) = v;
      }
    });
  }
}

int tiles_rw(int K) {  // rows a wave sweeps at a time (4 K rows per wave; 12 and more rows spill at 128 registers)
  switch (K) {
    case 1: return 4;
    case 2: return 8;
    case 3: return 6;    // 2 sweeps
    case 4: return 8;    // 2
    case 5: return 10;   // 2
    case 6: return 8;    // 3
    case 7: return 10;   // 3: 10 + 10 + 8
    default: return 8;   // K = 8: 4
  }
}

}  // namespace

extern "C" {

int mu_csr_tiles_rw(int k_layout) { return tiles_rw(k_layout); }

// bytes of the step stream / of the step table for a row stream of n_pos positions and nnz entries
int64_t mu_csr_tiles_entries(int64_t n_pos, int64_t n_cols, int64_t nnz) {
  const int64_t n_slabs = (n_cols + kTSlab - 1) / kTSlab;
  return nnz + 15 * n_slabs * n_pos + 64 * 5;  // (+ slack: a window may read up to 64 entries past a run)
}
int64_t mu_csr_tiles_steps_bytes(int64_t n_pos, int64_t n_cols, int k_layout) {
  const int64_t n_slabs = (n_cols + kTSlab - 1) / kTSlab;
  const int64_t wgs = (n_pos + 64 * (int64_t)k_layout - 1) / (64 * (int64_t)k_layout);
  const int rw = tiles_rw(k_layout);
  const int64_t n_sweeps = (4 * k_layout + rw - 1) / rw;
  return wgs * kTW * n_sweeps * n_slabs * 16;
}

#define MU_TILES_DISPATCH(KERNEL, ...)                                                                        \
  switch (tiles_rw(K)) {                                                                                        \
    case 4: hipLaunchKernelGGL((KERNEL<4>), dim3((unsigned)wgs), dim3(1024), 0, st, __VA_ARGS__); break;        \
    case 6: hipLaunchKernelGGL((KERNEL<6>), dim3((unsigned)wgs), dim3(1024), 0, st, __VA_ARGS__); break;        \
    case 8: hipLaunchKernelGGL((KERNEL<8>), dim3((unsigned)wgs), dim3(1024), 0, st, __VA_ARGS__); break;        \
    default: hipLaunchKernelGGL((KERNEL<10>), dim3((unsigned)wgs), dim3(1024), 0, st, __VA_ARGS__); break;      \
  }

int mu_csr_tiles_build(int64_t n_pos, int64_t n_cols, const int64_t* d_sptr, const void* d_ent, int k_layout,
                       void* d_tent, void* d_tsteps, void* stream) {
  MU_REQUIRE(n_pos >= 0 && n_cols > 0 && n_cols < ((int64_t)1 << 26), "shape out of range");
  MU_REQUIRE(k_layout >= 1 && k_layout <= 8, "layout K must be 1..8");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_sptr && d_ent && d_tent && d_tsteps, "null pointer");
  const int K = k_layout;
  const int64_t wgs = (n_pos + 64 * (int64_t)K - 1) / (64 * (int64_t)K);
  hipStream_t st = (hipStream_t)stream;
  MU_TILES_DISPATCH(k_tiles_build, n_pos, n_cols, K, d_sptr, (const unsigned long long*)d_ent,
                    (unsigned long long*)d_tent, (unsigned char*)d_tsteps)
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_spmm_tiles_f32(int64_t n_pos, int64_t n_cols, const int64_t* d_sptr, const void* d_tent,
                      const void* d_tsteps, const int32_t* d_perm, int k_layout, const float* d_Q, float* d_Y,
                      void* stream) {
  MU_REQUIRE(n_pos >= 0 && n_cols > 0 && n_cols < ((int64_t)1 << 26), "shape out of range");
  MU_REQUIRE(k_layout >= 1 && k_layout <= 8, "layout K must be 1..8");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_sptr && d_tent && d_tsteps && d_Q && d_Y, "null pointer");
  const int K = k_layout;
  const int64_t wgs = (n_pos + 64 * (int64_t)K - 1) / (64 * (int64_t)K);
  hipStream_t st = (hipStream_t)stream;
  if (mu_tune_get("spmm_mode") == 7 && tiles_rw(K) == 10) {  // ablation: no Q-slab copies
    hipLaunchKernelGGL((k_spmm_tiles<10, true>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols, K, d_sptr,
                       (const unsigned long long*)d_tent, (const unsigned char*)d_tsteps, d_perm, d_Q, d_Y);
    MU_CHECK_LAUNCH();
    return MU_OK;
  }
  MU_TILES_DISPATCH(k_spmm_tiles, n_pos, n_cols, K, d_sptr, (const unsigned long long*)d_tent,
                    (const unsigned char*)d_tsteps, d_perm, d_Q, d_Y)
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
