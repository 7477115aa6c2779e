// SpMM on the matrix cores for the LSI iteration (r04):  Y[n x 64] = X[n x d] * Q[d x 64]  with the
// rows of Q GATHERED from LDS by the hardware transpose read and the per-row sums formed by MFMA.
//
// It stands where ARPACK's reverse-communication loop calls csr_matvec / csr_matvecs through
// scipy.sparse.linalg.svds (/root/reference/muon/_atac/tools.py:53, scipy _svds.py:441-466,516), like
// csrc/spmm_win.hip - which gathers a 256-byte f32 row of Q from LDS per stored entry and multiplies
// on the vector ALU (one `ds_read_b128` + four VALU per four entries, four rows in lock step).  r03
// measured that formulation against its own floor: 36 ms per product at 1e6 x 2e5.
//
// This kernel is a different cut.  An MFMA D = A B sums over its K dimension; if column k of A holds
// the value of stored entry k in the row of D that entry belongs to (and zeros elsewhere) and row k of B
// is the row of Q the entry's column names, D accumulates Y for a whole tile of rows and ANY entry of the
// tile can sit in ANY k-slot: no lock step between rows, no padded e-steps.  gfx950 makes the B operand
// cheap: `ds_read_b64_tr_b16` lets every lane name its own 8-byte piece of LDS and hands the 16-lane
// group back the 4 x 16 block transposed - i.e. the B fragment of `v_mfma_f32_16x16x32_f16` built from
// four arbitrary rows of the slab, two such reads per 8 k-slots.  The operands are 16-bit:
//   * the dense block is rounded to f16 with one power-of-two scale per column, and the ROUNDED block is
//     what the caller keeps as its Krylov basis (muon_amd/_atac/tools.py): X Q~ is then exact, not a
//     perturbed product (f16 x f16 products are exact in f32, sums in f32 like the vector kernel);
//   * a stored value is split v = hi + lo (two f16, 22 bits) and the two halves occupy rows m and m + 8 of
//     A for tile row m - the M dimension of the MFMA is otherwise idle, so exact values cost nothing:
//     a tile is 8 rows, D rows 0-7 collect hi x Q~, rows 8-15 lo x Q~, added once at the end;
//   * NSET = 2 (the transposed product, whose dense operand Y is not a basis and must not be rounded):
//     the operand rows hold hi(64) | lo(64) and every A meets both halves (two MFMA sets per step).
//
// Operand layout ("cells", built once per lsi() call by mu_cells_cut): rows in tiles of 8, four tiles = a
// band (one wave), 16 bands = a workgroup's 512 rows; the columns in slabs of kSlabRows operand rows.  The
// entries of (tile, slab) are a cell, stored as steps of 32 k-slots (224 bytes: hi[32] f16, lo[32] f16,
// off[32] u16 in gather order, mask[4][8] u8: bit i of mask[kb][r] = slot 8 kb + i belongs to tile row r);
// a band's steps lie in the order the wave consumes them (slab, tile, step), so the wave reads ONE
// sequential stream, two steps ahead in registers, and the table hdr[band][slab] (steps per slab) is all
// the bookkeeping: no cursors, no compares, no re-requests.  Every step names its tile (bits 14-15 of
// its first offset).
//
// Q slab in LDS: rows padded to 160 bytes (NSET 2: 288), double buffered by LDS-DMA - the padded operand
// is contiguous in HBM, a slab is 80 (72) pieces of 1 KiB.  The pad makes the eight rows one half-wave
// gathers in one `ds_read_b64_tr_b16` fall into eight different 32-byte bank octets whenever their
// indices differ mod 8 (row r, column block nb -> octet (5 r + nb) mod 8; unpadded 128-byte rows would put
// all rows of one parity on the same four banks).
#include <type_traits>
#include <utility>
#include "common.hpp"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
#pragma clang diagnostic ignored "-Winline-asm"  // 32-bit LDS addresses made from integers

namespace {

typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));

constexpr int kStepBytes = 224;   // hi 64 | lo 64 | off 64 | mask 32
constexpr int kOffPlane = 128, kMaskPlane = 192;
constexpr int kRing = 2;          // ring slots per wave (asm-owned registers; 2 .. 5 measured alike)
[[maybe_unused]] constexpr int kWavesPerWg = 16;   // ... and its waves: 4 per SIMD, 128 registers = 64 accumulators + 12 ring (AGPRs, asm-owned) + 52
constexpr int kBandRows = 32;     // 4 tiles of 8 rows

template <int NSET> struct Geo {
  static constexpr int kStride = NSET == 1 ? 160 : 288;       // bytes of an operand row (data + 32 pad)
  static constexpr int kSlabRows = NSET == 1 ? 512 : 256;
  static constexpr int kSlabBytes = kStride * kSlabRows;      // 81920 / 73728
  static constexpr int kPieces = kSlabBytes / 1024;           // 80 / 72
  static_assert(kSlabBytes % 1024 == 0, "a slab is a whole number of LDS-DMA pieces");
};

// one LDS-DMA piece: 64 lanes x 16 B land contiguously at the wave-uniform LDS byte address
__device__ __forceinline__ void dma_piece(const void* base, unsigned byte_off, unsigned lds_dst) {
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

__device__ __forceinline__ u2_t tr16(unsigned lds_addr) {
  typedef __attribute__((address_space(3))) h4_t* lds_p;
  const h4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(lds_addr));
  return __builtin_bit_cast(u2_t, x);
}

struct Step {
  unsigned v[4];  // this lane's 8 values (hi for A rows 0-7, lo for rows 8-15) of k-slots 8 kb .. 8 kb + 7
                  // (four scalars, not a vector: hipcc 7.2 folds element p of a 4-vector that feeds a packed
                  //  16-bit multiply to element 0 - seen in the ISA, r04)
  unsigned m8;    // the dword that holds this lane's mask byte: bit i of it = slot 8 kb + i belongs to the lane's row
  unsigned op;    // gather offsets (8-byte granules) of slots 8 kb + j (low half) and 8 kb + 4 + j (high half)
};

// The 4 x 4 accumulator tiles (tile, column block) of a wave live in a[0:63], named literally in the asm below
// and never visible to hipcc as values: as C++ objects every case of the tile switch got fresh result
// registers and copies (twice the accumulator file, then spills).  The clobber lists make the kernel
// descriptor allocate them.
#define MU_ACC_CLOB                                                                                             \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", \
      "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
      "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", \
      "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", \
      "a62", "a63"

// tile T's four column blocks += A x B[nb].  (SrcC == vDst back to back is interlocked by the hardware; the
// operands come from LDS reads and VALU results, which the compiler waits for before an asm statement that
// names them.)
template <int T>
__device__ __forceinline__ void mfma_tile(const h8_t& A, const h8_t& b0, const h8_t& b1, const h8_t& b2,
                                          const h8_t& b3) {
  asm volatile(
      "v_mfma_f32_16x16x32_f16 a[%c0:%c1], %8, %9, a[%c0:%c1]\n\t"
      "v_mfma_f32_16x16x32_f16 a[%c2:%c3], %8, %10, a[%c2:%c3]\n\t"
      "v_mfma_f32_16x16x32_f16 a[%c4:%c5], %8, %11, a[%c4:%c5]\n\t"
      "v_mfma_f32_16x16x32_f16 a[%c6:%c7], %8, %12, a[%c6:%c7]"
      :
      : "i"(16 * T), "i"(16 * T + 3), "i"(16 * T + 4), "i"(16 * T + 7), "i"(16 * T + 8), "i"(16 * T + 11),
        "i"(16 * T + 12), "i"(16 * T + 15), "v"(A), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
      : MU_ACC_CLOB);
}
template <int T>
__device__ __forceinline__ void acc_poke(_Float16 x) {  // (timing ablation without MFMA: keeps the operands alive)
  asm volatile("v_accvgpr_write_b32 a%c0, %1" ::"i"(16 * T), "v"((float)x) : MU_ACC_CLOB);
}
template <int I>
__device__ __forceinline__ float acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(I) : MU_ACC_CLOB);
  return x;
}
template <int I>
__device__ __forceinline__ void acc_zero() {
  asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"i"(I) : MU_ACC_CLOB);
  if constexpr (I + 1 < 64) acc_zero<I + 1>();
}

// The step ring: two slots per wave, slot D = a[64 + 6 D .. 64 + 6 D + 5] = {8 values (4 dwords), mask dword, offset
// pair}.  The loads that fill it are issued from asm and tracked with exact counted waits (hipcc answers a register
// ring that is reloaded inside a loop with `s_waitcnt vmcnt(0)` - the whole prefetch distance gone - and post-
// processes a loaded byte right behind its load).  A slot is copied out with v_accvgpr_read once its loads landed.
// (Measured, profiles/r04_mfma_probe.txt: 2, 4 and 5 slots run alike; a record spread over the lanes - one dword
// per lane and step, 14 DPP row broadcasts to decode - is 11 % slower than these 16-lane replicated loads.)
#define MU_RING_CLOB "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75"
template <int D, int BYTE_OFF>
__device__ __forceinline__ void ring_request(const unsigned char* sp, unsigned off_val, unsigned off_msk,
                                             unsigned off_op) {
  asm volatile(
      "global_load_dwordx4 a[%c0:%c1], %4, %7 offset:%c8\n\t"
      "global_load_dword a%c2, %5, %7 offset:%c8\n\t"
      "global_load_dword a%c3, %6, %7 offset:%c8"
      :
      : "i"(64 + 6 * D), "i"(64 + 6 * D + 3), "i"(64 + 6 * D + 4), "i"(64 + 6 * D + 5), "v"(off_val), "v"(off_msk),
        "v"(off_op), "s"(sp), "i"(BYTE_OFF)
      : MU_RING_CLOB, "memory");
}
// wait until at most N vector-memory operations are outstanding, then copy slot D out
template <int D, int N>
__device__ __forceinline__ void ring_take(Step& st) {
  asm volatile(
      "s_waitcnt vmcnt(%c6)\n\t"
      "v_accvgpr_read_b32 %0, a%c7\n\t"
      "v_accvgpr_read_b32 %1, a%c8\n\t"
      "v_accvgpr_read_b32 %2, a%c9\n\t"
      "v_accvgpr_read_b32 %3, a%c10\n\t"
      "v_accvgpr_read_b32 %4, a%c11\n\t"
      "v_accvgpr_read_b32 %5, a%c12"
      : "=v"(st.v[0]), "=v"(st.v[1]), "=v"(st.v[2]), "=v"(st.v[3]), "=v"(st.m8), "=v"(st.op)
      : "i"(N), "i"(64 + 6 * D), "i"(64 + 6 * D + 1), "i"(64 + 6 * D + 2), "i"(64 + 6 * D + 3),
        "i"(64 + 6 * D + 4), "i"(64 + 6 * D + 5)
      : MU_RING_CLOB, "memory");
}

template <int... I, class F>
__device__ __forceinline__ void static_for_n_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for_n(F&& f) {
  static_for_n_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
template <int... I, class F>
__device__ __forceinline__ bool all_slots_impl(std::integer_sequence<int, I...>, F& f) {
  return (f(std::integral_constant<int, I>{}) && ...);
}
template <int N, class F>
__device__ __forceinline__ bool all_slots(F& f) {  // f(slot 0) && f(slot 1) && ... (stops at the first false)
  return all_slots_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int I, class F>
__device__ __forceinline__ void static_for_acc(F&& f) {
  f(std::integral_constant<int, I>{});
  if constexpr (I + 1 < 64) static_for_acc<I + 1>(f);
}

// MODE (timing ablations, results then wrong): 1 no MFMA, 4 no A masks, 8 per-wave cycle accounting,
// 16 no slab copies after the first, 32 no slab barrier
template <int NSET, int MODE, int W>
__device__ __forceinline__ void spmm_cells_body(int64_t n_rows, int64_t n_bands, int n_slabs,
                                                const int32_t* __restrict__ hdr,
                                                const int64_t* __restrict__ band_base,
                                                const unsigned char* __restrict__ cells,
                                                const unsigned char* __restrict__ Bop,
                                                const float* __restrict__ outscale, float* __restrict__ Y) {
  using G = Geo<NSET>;
  __shared__ __attribute__((aligned(1024))) unsigned char slab[2 * G::kSlabBytes];
  const int lane = threadIdx.x & 63;
  const int wave = uniform32(threadIdx.x >> 6);
  const int64_t band = (int64_t)blockIdx.x * W + wave;
  const bool active = band < n_bands;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&slab[0]);

  // lane roles.  A / D side: m = lane & 15 (row of A), kb = lane >> 4 (k-slots 8 kb .. 8 kb + 7).
  // Gather side (ds_read_b64_tr_b16): inside a 16-lane group, lane 4 j + c names the 8-byte piece c of the
  // row of k-slot 8 kb + 4 t + j (probed on hardware: tests/test_gpu_mfma.py).
  const int m = lane & 15, kb = lane >> 4;
  const int gj = (lane >> 2) & 3, gc = lane & 3;
  const unsigned off_val = (unsigned)((m < 8 ? 0 : 64) + kb * 16);
  const unsigned off_msk = (unsigned)(kMaskPlane + kb * 8 + (m & 4));  // the dword that holds this lane's mask byte
  const unsigned msk_sh = (unsigned)(8 * (m & 3));
  const unsigned off_op = (unsigned)(kOffPlane + (kb * 4 + gj) * 4);
  // (this lane's gather base inside slab buffer b: recomputed at the slab transitions, not kept in a register)
  auto gather_base = [&](int b) -> unsigned { return lds0 + (unsigned)gc * 8u + (unsigned)b * (unsigned)G::kSlabBytes; };

  acc_zero<0>();
  asm volatile("" ::: MU_RING_CLOB);  // (the ring registers belong to the asm too)

  constexpr int kMyPieces = (G::kPieces + W - 1) / W;  // LDS-DMA pieces per wave and slab: every wave the same
  auto issue_dma = [&](int slab_idx, int b) {          // number (a piece past the slab wraps onto its beginning)
#pragma unroll
    for (int u = 0; u < kMyPieces; ++u) {
      int piece = wave + W * u;
      if (G::kPieces % W != 0 && piece >= G::kPieces) piece -= G::kPieces;
      dma_piece(Bop, (unsigned)slab_idx * (unsigned)G::kSlabBytes + (unsigned)(piece * 1024 + lane * 16),
                lds0 + (unsigned)b * (unsigned)G::kSlabBytes + (unsigned)piece * 1024u);
    }
  };

  // (the table is read through the constant address space: a scalar load, so that the step counts - and with
  //  them the whole control flow and the stream pointer - stay in scalar registers)
  typedef __attribute__((address_space(4))) const int32_t* chdr_p;
  const chdr_p myhdr = (chdr_p)(hdr + (active ? band : 0) * (int64_t)n_slabs);
  auto steps_of = [&](int slab_idx) -> int { return active ? uniform32(myhdr[slab_idx]) : 0; };
  const unsigned char* sp = cells + uniform64(active ? band_base[band] : 0) * (int64_t)kStepBytes;

  // A step in two stages, one step apart (software pipeline): FRONT = take the record from its ring slot, request
  // the slot's next record, decode (A fragment, gather addresses, tile), issue the 8 NSET transpose reads; BACK =
  // the MFMAs, one step later, when the reads have long landed - under them run the front of the next step and the
  // other waves.  Two register sets (ring slot D) hold a step between its stages.
  typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
  struct Pend {
    unsigned a[4];
    u4_t b[NSET][4];
    int tile;
  };
  Pend pend[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {  // (what the first BACK multiplies: zeros into tile 0)
#pragma unroll
    for (int p = 0; p < 4; ++p) pend[d].a[p] = 0u;
#pragma unroll
    for (int h = 0; h < NSET; ++h)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) pend[d].b[h][nb] = u4_t{0u, 0u, 0u, 0u};
    pend[d].tile = 0;
  }
  unsigned cur_base = gather_base(0);  // this lane's gather base inside the slab buffer in use

  auto front = [&](const Step& st, Pend& pd) {
    // mask byte -> per pair p the 16-bit words (bit 2p, bit 2p + 1) in one register: T = m8 + (m8 << 15) has bit
    // 2p at 2p and bit 2p + 1 at 16 + 2p; the f16 bit patterns are then MULTIPLIED by 0 / 1 as packed u16
    const unsigned m8 = (st.m8 >> msk_sh) & 0xffu;
    const unsigned T = m8 * 0x8001u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if constexpr (MODE & 4) {
        pd.a[p] = st.v[p];
      } else {
        const unsigned bits = (T >> (2 * p)) & 0x00010001u;
        pd.a[p] = __builtin_bit_cast(unsigned, __builtin_bit_cast(us2_t, st.v[p]) * __builtin_bit_cast(us2_t, bits));
      }
    }
    unsigned a0 = ((st.op & 0x3fffu) << 3) + cur_base;  // (bits 14-15 of the low half: the tile)
    unsigned a1 = ((st.op >> 16) << 3) + cur_base;
    if constexpr (MODE & 64) {  // (timing: every gather reads row 0 of the slab - no bank conflicts)
      a0 = cur_base + (st.op & 1u) * 0u;
      a1 = cur_base;
    }
    pd.tile = (__builtin_amdgcn_readfirstlane((int)st.op) >> 14) & 3;
#pragma unroll
    for (int h = 0; h < NSET; ++h)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const u2_t x = tr16(a0 + (unsigned)(h * 128 + nb * 32));
        const u2_t y = tr16(a1 + (unsigned)(h * 128 + nb * 32));
        pd.b[h][nb] = u4_t{x[0], x[1], y[0], y[1]};
      }
  };
  auto back = [&](const Pend& pd) {
    const h8_t A = __builtin_bit_cast(h8_t, u4_t{pd.a[0], pd.a[1], pd.a[2], pd.a[3]});
#pragma unroll
    for (int h = 0; h < NSET; ++h) {
      const h8_t b0 = __builtin_bit_cast(h8_t, pd.b[h][0]), b1 = __builtin_bit_cast(h8_t, pd.b[h][1]);
      const h8_t b2 = __builtin_bit_cast(h8_t, pd.b[h][2]), b3 = __builtin_bit_cast(h8_t, pd.b[h][3]);
      if constexpr (MODE & 1) {
        acc_poke<0>(A[0] + b0[0] + b1[1] + b2[2] + b3[3]);
      } else {
        if (pd.tile < 2) {
          if (pd.tile == 0) mfma_tile<0>(A, b0, b1, b2, b3);
          else mfma_tile<1>(A, b0, b1, b2, b3);
        } else {
          if (pd.tile == 2) mfma_tile<2>(A, b0, b1, b2, b3);
          else mfma_tile<3>(A, b0, b1, b2, b3);
        }
      }
    }
  };
  // VMEM order of a wave: ... Q0 Q1 | burst of kMyPieces pieces | Q0 Q1 ... (Q = the three loads of a ring request;
  // a step takes slot D and requests it again).  Behind the loads of the slot being taken there is always the other
  // slot's request (3 loads) - plus the burst for the first two steps after it (BURST).
  auto step = [&](auto dcn, auto burstc) {
    constexpr int D = decltype(dcn)::value;
    constexpr int N = 3 + (decltype(burstc)::value ? kMyPieces : 0);
    Step st;
    ring_take<D, N>(st);
    ring_request<D, 2 * kStepBytes>(sp, off_val, off_msk, off_op);
    if constexpr (!(MODE & 128)) sp += kStepBytes;  // (128, timing: the same two records again and again - cache hits)
    front(st, pend[D]);
    back(pend[D ^ 1]);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // MODE & 8: per-wave cycle accounting (s_memtime): the steps, the slab transitions (DMA wait, barrier, burst)
  unsigned t_steps = 0, t_slab = 0, n_steps = 0;
  auto now = [&]() -> unsigned { return (unsigned)__builtin_amdgcn_s_memtime(); };
  const unsigned t_begin = (MODE & 8) ? now() : 0u;

  // prologue: slab 0 lands, slab 1 is requested, the first two steps are requested
  issue_dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (n_slabs > 1) issue_dma(1, 1);
  ring_request<0, 0>(sp, off_val, off_msk, off_op);
  ring_request<1, kStepBytes>(sp, off_val, off_msk, off_op);
  int left = steps_of(0);                          // steps of this wave in the slab: even (mu_cells_cut pads)
  int next_left = n_slabs > 1 ? steps_of(1) : 0;   // (read a slab ahead: the scalar load is never waited for)
  bool burst = false;                              // a burst went out behind the requests in flight
  for (int s = 0;;) {
    unsigned ts = 0;
    if constexpr (MODE & 8) ts = now();
    int pairs = left >> 1;
    if (pairs > 0) {
      if (burst) {
        step(S0{}, std::true_type{});
        step(S1{}, std::true_type{});
        --pairs;
      }
      for (; pairs > 0; --pairs) {
        step(S0{}, std::false_type{});
        step(S1{}, std::false_type{});
      }
    }
    if constexpr (MODE & 8) {
      t_steps += now() - ts;
      n_steps += (unsigned)left;
      ts = now();
    }
    if (s + 1 >= n_slabs) break;
    // this wave is through with slab s (the gathers of its last step are in registers or on their way: LDS reads of
    // one wave are in order, the barrier's lgkmcnt(0) covers them).  Its pieces of slab s + 1 went out before the
    // two requests in flight - if the slab had steps: leaving those six loads outstanding proves the pieces landed.
    if (left > 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!(MODE & 32)) __syncthreads();  // slab s + 1 visible to everyone; everyone finished reading slab s
    ++s;
    cur_base = gather_base(s & 1);
    burst = false;
    if (s + 1 < n_slabs) {
      if constexpr (!(MODE & 16)) {
        issue_dma(s + 1, (s + 1) & 1);
        burst = true;
      }
    }
    left = next_left;
    next_left = s + 1 < n_slabs ? steps_of(s + 1) : 0;
    if constexpr (MODE & 8) t_slab += now() - ts;
  }
  back(pend[1]);  // (the last step's MFMAs; zeros if the wave had no step)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requests still in flight target the ring registers
  if constexpr (MODE & 8) {
    const unsigned t_all = now() - t_begin;
    if (active && lane < 5) {
      const unsigned t = lane == 0 ? t_all : lane == 1 ? 0u : lane == 2 ? t_steps : lane == 3 ? t_slab : n_steps;
      Y[band * kBandRows * 64 + lane] = (float)t;
    }
    return;
  }

  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMA results are read by the vector ALU below
  // D rows 0-7 (lanes 0-31) hold hi x Q~, rows 8-15 (lanes 32-63) lo x Q~ of the same tile rows
  const int n = lane & 15, q = lane >> 4;
  static_for_acc<0>([&](auto ic) {
    constexpr int I = decltype(ic)::value;  // 16 t + 4 nb + r
    constexpr int t = I >> 4, nb = (I >> 2) & 3, r = I & 3;
    const float x = acc_read<I>();
    const float y = __shfl_xor(x, 32, 64);
    const int64_t row = band * kBandRows + t * 8 + 4 * q + r;
    if (active && q < 2 && row < n_rows) Y[row * 64 + 16 * nb + n] = (x + y) * outscale[16 * nb + n];
  });
}


// 16 waves (4 per SIMD, 128 registers each): 76 accumulator + ring AGPRs leave 52 VGPRs - MU_VGPR_CAP tells hipcc
// (it counts a unified register file in halves: amdgpu_num_vgpr(N) caps its allocation at v[0 .. 2 N - 1])
#define MU_VGPR_CAP 26
template <int NSET, int MODE>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(MU_VGPR_CAP))) void k_spmm_cells16(
    int64_t n_rows, int64_t n_bands, int n_slabs, const int32_t* __restrict__ hdr,
    const int64_t* __restrict__ band_base, const unsigned char* __restrict__ cells,
    const unsigned char* __restrict__ Bop, const float* __restrict__ outscale, float* __restrict__ Y) {
  spmm_cells_body<NSET, MODE, 16>(n_rows, n_bands, n_slabs, hdr, band_base, cells, Bop, outscale, Y);
}
// 12 waves (3 per SIMD, 168 registers each): the two-term operand's stage registers (2 x 36) need them
template <int NSET, int MODE>
__global__ __launch_bounds__(768) void k_spmm_cells12(int64_t n_rows, int64_t n_bands, int n_slabs,
                                                      const int32_t* __restrict__ hdr,
                                                      const int64_t* __restrict__ band_base,
                                                      const unsigned char* __restrict__ cells,
                                                      const unsigned char* __restrict__ Bop,
                                                      const float* __restrict__ outscale, float* __restrict__ Y) {
  spmm_cells_body<NSET, MODE, 12>(n_rows, n_bands, n_slabs, hdr, band_base, cells, Bop, outscale, Y);
}

// ---- the dense operand: f32 block -> padded f16 rows (+ the rounded block back in place) ------------------
__global__ __launch_bounds__(256) void k_colabsmax(int64_t rows, const float* __restrict__ Q,
                                                   float* __restrict__ part) {
  __shared__ float red[256];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  float mx = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + g; r < rows; r += (int64_t)gridDim.x * 4)
    mx = fmaxf(mx, fabsf(Q[r * 64 + c]));
  red[threadIdx.x] = mx;
  __syncthreads();
  if (g == 0) part[(int64_t)blockIdx.x * 64 + c] = fmaxf(fmaxf(red[c], red[64 + c]), fmaxf(red[128 + c], red[192 + c]));
}

// scale[c] = 2^e with the column's largest entry in [2^13, 2^14) after division (f16 keeps 11 bits down to
// 2^-14: 27 binades below the column maximum are rounded relatively, the rest absolutely to 2^-25 of it)
__global__ void k_colscale(int n_part, const float* __restrict__ part, float* __restrict__ scale,
                           float* __restrict__ inv) {
  const int c = threadIdx.x;
  if (c >= 64) return;
  float mx = 0.f;
  for (int i = 0; i < n_part; ++i) mx = fmaxf(mx, part[(int64_t)i * 64 + c]);
  int e = 0;
  if (mx > 0.f && mx < INFINITY) {
    frexpf(mx, &e);  // mx = f * 2^e, f in [0.5, 1)
    e -= 14;         // mx / 2^e in [2^13, 2^14)
  }
  scale[c] = ldexpf(1.f, e);
  inv[c] = ldexpf(1.f, -e);
}

// 8 threads per row, 8 columns each
__global__ __launch_bounds__(256) void k_dense_to_f16(int64_t rows, int64_t rows_padded, float* __restrict__ Q,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ inv,
                                                      unsigned char* __restrict__ out, int stride, int nset,
                                                      int rewrite) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = t >> 3;
  const int c0 = (int)(t & 7) * 8;
  if (r >= rows_padded) return;
  h8_t hi, lo;
  if (r < rows) {
    float x[8];
    const f4_t q0 = *reinterpret_cast<const f4_t*>(Q + r * 64 + c0);
    const f4_t q1 = *reinterpret_cast<const f4_t*>(Q + r * 64 + c0 + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[i] = q0[i];
      x[4 + i] = q1[i];
    }
    float back[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xs = x[i] * inv[c0 + i];
      hi[i] = (_Float16)xs;
      lo[i] = (_Float16)(xs - (float)hi[i]);
      back[i] = (float)hi[i] * scale[c0 + i];
    }
    if (rewrite) {
      *reinterpret_cast<f4_t*>(Q + r * 64 + c0) = f4_t{back[0], back[1], back[2], back[3]};
      *reinterpret_cast<f4_t*>(Q + r * 64 + c0 + 4) = f4_t{back[4], back[5], back[6], back[7]};
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = lo[i] = (_Float16)0.f;
  }
  unsigned char* o = out + r * (int64_t)stride + c0 * 2;
  *reinterpret_cast<u4_t*>(o) = __builtin_bit_cast(u4_t, hi);
  if (nset == 2) *reinterpret_cast<u4_t*>(o + 128) = __builtin_bit_cast(u4_t, lo);
}

// ---- cutting a CSR into cells ----------------------------------------------------------------------------
// A wave per band (32 rows = 4 tiles of 8); 8 lanes per row read the next 8 entries behind the row's cursor,
// the entries of the current slab are a prefix of that window (sorted rows).  The slots of a cell are filled
// in arrival order (any order is a valid cell); a step is staged in LDS plane by plane and leaves as 56 dwords.
constexpr int kCutWaves = 4;
constexpr int kStageSteps = 4;  // 31 slots left over + 64 new ones < 4 x 32

__global__ __launch_bounds__(64 * kCutWaves) void k_cells_cut(
    int64_t n_rows, int64_t n_cols, int n_slabs, int slab_rows, int gran, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const float* __restrict__ values, const float* __restrict__ vinv_p,
    const int64_t* __restrict__ band_base, int64_t n_bands, unsigned char* __restrict__ cells,
    int32_t* __restrict__ hdr, int* __restrict__ d_err) {
  __shared__ unsigned stage_all[kCutWaves][kStageSteps][kStepBytes / 4];
  const int lane = threadIdx.x & 63;
  const int wave = uniform32(threadIdx.x >> 6);
  const int64_t band = (int64_t)blockIdx.x * kCutWaves + wave;
  if (band >= n_bands) return;
  unsigned(*stage)[kStepBytes / 4] = stage_all[wave];
  for (int i = lane; i < kStageSteps * (kStepBytes / 4); i += 64) (&stage[0][0])[i] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

  const float vinv = *vinv_p;
  const int r = lane >> 3, i8 = lane & 7;
  int64_t cur[4], end[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int64_t row = band * kBandRows + t * 8 + r;
    cur[t] = row < n_rows ? indptr[row] : 0;
    end[t] = row < n_rows ? indptr[row + 1] : 0;
  }
  unsigned char* out = cells + uniform64(band_base[band]) * (int64_t)kStepBytes;
  unsigned char* const out_end = cells + uniform64(band_base[band + 1]) * (int64_t)kStepBytes;
  int32_t* myhdr = hdr + band * (int64_t)n_slabs;

  for (int s = 0; s < n_slabs; ++s) {
    const int col0 = s * slab_rows;
    const int64_t ce = (int64_t)col0 + slab_rows;
    const int col_end = (int)(ce < n_cols ? ce : n_cols);
    int steps_total = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int slots = 0, flushed = 0;  // of this cell (uniform)
      bool more;
      do {
        const int64_t p = cur[t] + i8;
        const bool valid = p < end[t];
        const int col = valid ? indices[p] : 0x7fffffff;
        const float val = valid ? values[p] : 0.f;
        const bool in = col < col_end;
        const unsigned long long mask = __ballot(in);
        const int prefix = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (in) {
          const int slot = slots + prefix;
          unsigned* st = stage[(slot >> 5) & (kStageSteps - 1)];
          const int k = slot & 31;
          const float vs = val * vinv;
          const _Float16 h = (_Float16)vs;
          const _Float16 l = (_Float16)(vs - (float)h);
          reinterpret_cast<_Float16*>(st)[k] = h;
          reinterpret_cast<_Float16*>(st)[32 + k] = l;
          const int kbq = k >> 3, ii = k & 7;
          // gather order: (slot 8 kb + j, slot 8 kb + 4 + j) pairs; bits 14-15 of a low half = the tile
          const int pi = kbq * 8 + 2 * (ii & 3) + (ii >> 2);
          reinterpret_cast<uint16_t*>(st)[kOffPlane / 2 + pi] =
              (uint16_t)(((col - col0) * gran) | ((ii >> 2) ? 0 : (t << 14)));
          atomicOr(&st[kMaskPlane / 4 + kbq * 2 + (r >> 2)], 1u << (8 * (r & 3) + ii));
        }
        const unsigned grp = (unsigned)(mask >> (lane & 56)) & 0xffu;
        const int cnt = __popc(grp);
        cur[t] += cnt;
        slots += __popcll(mask);
        more = __ballot(cnt == 8) != 0ull;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        while (slots - flushed * 32 >= 32 || (!more && slots > flushed * 32)) {
          unsigned* st = stage[flushed & (kStageSteps - 1)];
          if (out + kStepBytes > out_end) {
            if (lane == 0) atomicExch(d_err, 1);  // the caller's bound on the band's steps was too small
          } else if (lane < kStepBytes / 4) {
            reinterpret_cast<unsigned*>(out)[lane] = st[lane];
          }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          if (lane < kStepBytes / 4) st[lane] = 0u;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          if (out + kStepBytes <= out_end) out += kStepBytes;
          ++flushed;
        }
      } while (more);
      steps_total += flushed;
    }
    if (steps_total & 1) {
      // the product consumes steps in pairs (its two ring slots are named in the code): an all-zero step - zero
      // values, no row bits, row 0 of the slab, tile 0 - completes an odd slab
      if (out + kStepBytes > out_end) {
        if (lane == 0) atomicExch(d_err, 1);
      } else {
        if (lane < kStepBytes / 4) reinterpret_cast<unsigned*>(out)[lane] = 0u;
        out += kStepBytes;
      }
      ++steps_total;
    }
    if (lane == 0) myhdr[s] = steps_total;
  }
}

// ---- probes (hardware semantics the kernel rests on; tests/test_gpu_mfma.py) -------------------------------
__global__ void k_probe_tr16(const unsigned* __restrict__ image, int n_dwords, const unsigned* __restrict__ addr,
                             unsigned* __restrict__ out) {
  __shared__ unsigned img[16384];
  for (int i = threadIdx.x; i < n_dwords && i < 16384; i += blockDim.x) img[i] = image[i];
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&img[0]);
  const u2_t x = tr16(lds0 + addr[threadIdx.x]);
  out[2 * threadIdx.x] = x[0];
  out[2 * threadIdx.x + 1] = x[1];
}

__global__ void k_probe_mfma16(const unsigned* __restrict__ a, const unsigned* __restrict__ b, float* __restrict__ d) {
  const int l = threadIdx.x;
  const u4_t av = u4_t{a[4 * l], a[4 * l + 1], a[4 * l + 2], a[4 * l + 3]};
  const u4_t bv = u4_t{b[4 * l], b[4 * l + 1], b[4 * l + 2], b[4 * l + 3]};
  f4_t c = f4_t{0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, av), __builtin_bit_cast(h8_t, bv), c, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) d[4 * l + i] = c[i];
}

}  // namespace

extern "C" {

int mu_cells_geometry(int nset, int* slab_rows, int* stride, int* step_bytes, int* band_rows, int* ring) {
  MU_REQUIRE(nset == 1 || nset == 2, "nset must be 1 or 2");
  if (slab_rows) *slab_rows = nset == 1 ? Geo<1>::kSlabRows : Geo<2>::kSlabRows;
  if (stride) *stride = nset == 1 ? Geo<1>::kStride : Geo<2>::kStride;
  if (step_bytes) *step_bytes = kStepBytes;
  if (band_rows) *band_rows = kBandRows;
  if (ring) *ring = kRing;
  return MU_OK;
}

int mu_cells_cut(int nset, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices,
                 const float* d_values, const float* d_value_inv_scale, const int64_t* d_band_base, void* d_cells,
                 int32_t* d_hdr, int* d_err, void* stream) {
  MU_REQUIRE(nset == 1 || nset == 2, "nset must be 1 or 2");
  MU_REQUIRE(n_rows >= 0 && n_cols > 0 && n_cols < ((int64_t)1 << 31), "shape out of range");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_indices && d_values && d_value_inv_scale && d_band_base && d_cells && d_hdr && d_err,
             "null pointer");
  const int slab_rows = nset == 1 ? Geo<1>::kSlabRows : Geo<2>::kSlabRows;
  const int stride = nset == 1 ? Geo<1>::kStride : Geo<2>::kStride;
  const int64_t n_bands = (n_rows + kBandRows - 1) / kBandRows;
  const int64_t n_slabs = (n_cols + slab_rows - 1) / slab_rows;
  MU_REQUIRE(n_slabs < 65536, "too many column slabs");
  hipLaunchKernelGGL(k_cells_cut, dim3((unsigned)((n_bands + kCutWaves - 1) / kCutWaves)), dim3(64 * kCutWaves), 0,
                     (hipStream_t)stream, n_rows, n_cols, (int)n_slabs, slab_rows, stride / 8, d_indptr, d_indices,
                     d_values, d_value_inv_scale, d_band_base, n_bands, (unsigned char*)d_cells, d_hdr, d_err);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

size_t mu_dense_f16_worksize(int64_t rows) {
  int64_t blocks = (rows + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  return (size_t)blocks * 64 * sizeof(float);
}

int mu_dense_to_f16(int nset, int64_t rows, int64_t rows_padded, float* d_Q, int rewrite, void* d_out,
                    float* d_scale, float* d_inv, void* d_work, size_t work_bytes, void* stream) {
  MU_REQUIRE(nset == 1 || nset == 2, "nset must be 1 or 2");
  MU_REQUIRE(rows >= 0 && rows_padded >= rows, "shape out of range");
  if (rows_padded == 0) return MU_OK;
  MU_REQUIRE(d_Q && d_out && d_scale && d_inv && d_work, "null pointer");
  MU_REQUIRE(work_bytes >= mu_dense_f16_worksize(rows), "work buffer too small");
  const int stride = nset == 1 ? Geo<1>::kStride : Geo<2>::kStride;
  MU_REQUIRE(rows_padded * (int64_t)stride < ((int64_t)1 << 32), "dense operand of 4 GiB or more");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)(mu_dense_f16_worksize(rows) / (64 * sizeof(float)));
  hipLaunchKernelGGL(k_colabsmax, dim3((unsigned)blocks), dim3(256), 0, st, rows, d_Q, (float*)d_work);
  hipLaunchKernelGGL(k_colscale, dim3(1), dim3(64), 0, st, blocks, (const float*)d_work, d_scale, d_inv);
  const int64_t threads = rows_padded * 8;
  hipLaunchKernelGGL(k_dense_to_f16, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, rows, rows_padded,
                     d_Q, d_scale, d_inv, (unsigned char*)d_out, stride, nset, rewrite);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_spmm_cells_f32(int nset, int64_t n_rows, int64_t n_operand_rows, const int32_t* d_hdr,
                      const int64_t* d_band_base, const void* d_cells, const void* d_B16,
                      const float* d_outscale, float* d_Y, void* stream) {
  MU_REQUIRE(nset == 1 || nset == 2, "nset must be 1 or 2");
  MU_REQUIRE(n_rows >= 0 && n_operand_rows > 0, "shape out of range");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_hdr && d_band_base && d_cells && d_B16 && d_outscale && d_Y, "null pointer");
  const int slab_rows = nset == 1 ? Geo<1>::kSlabRows : Geo<2>::kSlabRows;
  const int64_t n_bands = (n_rows + kBandRows - 1) / kBandRows;
  const int64_t n_slabs = (n_operand_rows + slab_rows - 1) / slab_rows;
  const int mode = mu_tune_get("mfma_mode");
  hipStream_t st = (hipStream_t)stream;
#define MU_GO(NS, MD, WW)                                                                                      \
  hipLaunchKernelGGL((k_spmm_cells##WW<NS, MD>), dim3((unsigned)((n_bands + WW - 1) / WW)), dim3(64 * WW), 0,  \
                     st, n_rows, n_bands, (int)n_slabs, d_hdr, d_band_base, (const unsigned char*)d_cells,     \
                     (const unsigned char*)d_B16, d_outscale, d_Y)
  if (nset == 1) {
    if (mode == 0) MU_GO(1, 0, 16);
    else if (mode == 1) MU_GO(1, 1, 16);
    else if (mode == 4) MU_GO(1, 4, 16);
    else if (mode == 8) MU_GO(1, 8, 16);
    else if (mode == 16) MU_GO(1, 16, 16);
    else if (mode == 32) MU_GO(1, 32, 16);
    else if (mode == 53) MU_GO(1, 53, 16);
    else if (mode == 64) MU_GO(1, 64, 16);
    else if (mode == 117) MU_GO(1, 117, 16);
    else if (mode == 128) MU_GO(1, 128, 16);
    else if (mode == 181) MU_GO(1, 181, 16);
    else if (mode == 245) MU_GO(1, 245, 16);
    else { mu_set_error("mfma_mode %d has no compiled instance", mode); return MU_ERR_ARG; }
  } else {
    if (mode == 0) MU_GO(2, 0, 12);  // (two stage register sets of 36: 12 waves of 168 registers)
    else if (mode == 8) MU_GO(2, 8, 12);
    else { mu_set_error("mfma_mode %d has no compiled instance", mode); return MU_ERR_ARG; }
  }
#undef MU_GO
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_probe_tr16(const void* d_image, int n_dwords, const void* d_addr, void* d_out, void* stream) {
  MU_REQUIRE(d_image && d_addr && d_out && n_dwords > 0, "null pointer");
  hipLaunchKernelGGL(k_probe_tr16, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)d_image, n_dwords,
                     (const unsigned*)d_addr, (unsigned*)d_out);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_probe_mfma16(const void* d_a, const void* d_b, float* d_d, void* stream) {
  MU_REQUIRE(d_a && d_b && d_d, "null pointer");
  hipLaunchKernelGGL(k_probe_mfma16, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)d_a,
                     (const unsigned*)d_b, d_d);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
