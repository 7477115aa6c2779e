// EXPERIMENT (r04, VERDICT r03 item 1 b): the LSI product Y[n x 64] = X Q on a PRE-TILED f32 operand.
//
// `k_spmm_win` (csrc/spmm_win.hip) finds the entries of every (row-set, slab) visit again in every product: an
// unaligned 128-byte window request behind a cursor, a compare, a ballot, counts, the cursor update, the tail of the
// window requested once more one slab later (stage A: 35 of its 103 instructions per visit, 2.0-2.9x the algorithmic
// bytes through the fabric), and every wave issues its share of the Q-slab copies.  VERDICT's probe: hand stage B an
// operand in which the entries of a (row-set, 256-column slab) visit are contiguous with a count - no cursor, no
// ballot, no re-request - built offline by any means, and see what the product then costs (gate: 3.7 ms at 125 000 x
// 200 000 against 4.36, traffic <= 1.2x).  This file is that probe, written with what csrc/spmm_ell.hip learnt:
//
//   * lane layout of `k_spmm_win`: a wave = four 16-lane groups, group g walks row g of a row-set, lane c of the group
//     owns dense columns 4c .. 4c+3; K row-sets per wave (accumulators: K x float4);
//   * operand: ONE window of 384 bytes per (row-set, slab) at a FIXED place - value[64] f32, then offset[64] u16 (byte
//     offset of the Q row inside the slab = column % 256 * 256); slot 16 g + t = row g's step t; steps past a row's own
//     entries are (0.0, 0).  A byte per (row-set, slab) holds the steps (the longest of the four rows); more than 16
//     (0.4 % of the rows per slab on the bench matrix) continue in a per-wave overflow stream of the same windows.
//     75 GB per product at 1e6 x 200k - against 50.5 algorithmic and 104-134 GB fetched by the window kernel;
//   * a visit: take the row-set's window (requested one slab sweep ago into the row-set's own pair of asm-owned AGPRs:
//     between request and take there are exactly K - 1 younger requests, `s_waitcnt vmcnt(2 (K - 1))`), request the
//     next slab's window at an immediate offset from one pointer, then `steps` e-steps of add_dpp / mov_dpp /
//     ds_read_b128 / 2 v_pk_fma_f32, left in pairs by a counted branch;
//   * wave 0 of the workgroup only copies Q slabs (csrc/spmm_ell.hip: why).
// Sums: a row's products in stored (column) order into one accumulator - the same arithmetic as `k_spmm_win`.
#include <type_traits>
#include <utility>

#include "common.hpp"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"  // 32-bit LDS addresses made from integers

namespace {

constexpr int kTSlabRows = 256;              // Q rows per slab: 256 x 256 B = 64 KiB, double buffered
constexpr int kTSlabBytes = kTSlabRows * 256;
constexpr int kTMaxWaves = 15;               // row-owning waves (+ the producer)
constexpr int kTWin = 384;                   // bytes of a window

typedef float f4 __attribute__((ext_vector_type(4)));

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

// The window of row-set k in flight lives in v[106 + 2 k] (offsets) and v[107 + 2 k] (values): registers hipcc never
// sees as values (amdgpu_num_vgpr(53) caps its own allocation at v[0 .. 105] on the unified register file when no AGPR
// is in use - csrc/spmm_win.hip has the same arrangement and tests/test_layout.py audits the ISA), written by loads
// issued from asm, waited for with exact counts, read by the DPP instructions of the steps by NAME.
#define MU_T_CLOB                                                                                         \
  "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", \
      "v119", "v120", "v121", "v122", "v123", "v124", "v125"
constexpr int kTRing = 106;  // (K <= 10 row-sets)
// request the window of row-set KI of the slab `wp` points at (offsets are immediates: KI * 384 (+ 256))
template <int KI>
__device__ __forceinline__ void t_request(const unsigned char* wp, unsigned lane4, unsigned lane2) {
  asm volatile(
      "global_load_ushort v%c0, %3, %4 offset:%c6\n\t"
      "global_load_dword v%c1, %2, %4 offset:%c5" ::"i"(kTRing + 2 * KI),
      "i"(kTRing + 2 * KI + 1), "v"(lane4), "v"(lane2), "s"(wp), "i"(kTWin * KI), "i"(kTWin * KI + 256)
      : MU_T_CLOB, "memory");
}
template <int N>
__device__ __forceinline__ void t_wait() {
  asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(N) : MU_T_CLOB, "memory");
}
// (address, value) of step T of this lane's row: lane 16 g + T of row-set KI's window, broadcast inside the 16-lane group
template <int KI, int T>
__device__ __forceinline__ void t_step_operands(unsigned base, unsigned& adr, float& v) {
  asm volatile(
      "v_add_u32_dpp %0, v%c3, %2 row_newbcast:%c5 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %1, v%c4 row_newbcast:%c5 row_mask:0xf bank_mask:0xf"
      : "=&v"(adr), "=&v"(v)
      : "v"(base), "i"(kTRing + 2 * KI), "i"(kTRing + 2 * KI + 1), "i"(T)
      : MU_T_CLOB);
}
// the same from a window held in compiler registers (the overflow path)
template <int T>
__device__ __forceinline__ void t_step_operands_r(unsigned off, float val, unsigned base, unsigned& adr, float& v) {
  asm("v_add_u32_dpp %0, %2, %4 row_newbcast:%c5 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %1, %3 row_newbcast:%c5 row_mask:0xf bank_mask:0xf"
      : "=&v"(adr), "=&v"(v)
      : "v"(off), "v"(val), "v"(base), "i"(T));
}

// MODE (timing ablations, wrong results): 1 no gathers / FMAs, 2 no slab copies, 4 overflow windows ignored
template <int K, int MODE>
__global__ __launch_bounds__(64 * (kTMaxWaves + 1)) __attribute__((amdgpu_num_vgpr(53))) void k_spmm_ell64(
    int64_t n_pos, int64_t n_cols, int n_slabs, const unsigned long long* __restrict__ hdr,
    const unsigned char* __restrict__ ent, const int64_t* __restrict__ ovf_base, const unsigned char* __restrict__ ovf,
    const int32_t* __restrict__ perm, const float* __restrict__ Q, float* __restrict__ Y) {
  __shared__ __attribute__((aligned(1024))) unsigned char slab[2 * kTSlabBytes];
  const int lane = threadIdx.x & 63;
  const int wave = uniform32(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&slab[0]);

  if (wave == 0) {
    const unsigned q_last = (unsigned)(n_cols * 256 - 16);
    auto whole = [&](int s, int b) {
#pragma unroll 8
      for (int piece = 0; piece < ((MODE & 2) ? 1 : kTSlabBytes / 1024); ++piece) {
        unsigned o = (unsigned)s * (unsigned)kTSlabBytes + (unsigned)(piece * 1024 + lane * 16);
        o = o < q_last ? o : q_last;
        t_dma_piece(Q, o, lds0 + (unsigned)b * (unsigned)kTSlabBytes + (unsigned)piece * 1024u);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    whole(0, 0);
    __syncthreads();
    for (int s = 0; s + 1 < n_slabs; ++s) {
      whole(s + 1, (s + 1) & 1);
      __syncthreads();
    }
    return;
  }

  const int64_t n_waves = (n_pos + 4 * K - 1) / (4 * K);
  const int64_t gwave = (int64_t)blockIdx.x + (int64_t)gridDim.x * (wave - 1);
  const bool active = gwave < n_waves;
  const unsigned lane4 = (unsigned)lane * 4u, lane2 = (unsigned)lane * 2u;
  const unsigned lane_c = (unsigned)(lane & 15) * 16u;

  f4 acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
  asm volatile("" ::: MU_T_CLOB);  // (the kernel descriptor must allocate v110 .. v125)

  typedef __attribute__((address_space(4))) const unsigned long long* chdr_p;
  const chdr_p myhdr = (chdr_p)(hdr + (active ? gwave : 0) * (int64_t)n_slabs * 2);  // 16 count bytes per (wave, slab)
  auto counts_of = [&](int s, int half) -> unsigned long long { return active ? uniform64(myhdr[2 * s + half]) : 0ull; };
  const unsigned char* wp = ent + (active ? gwave : 0) * (int64_t)n_slabs * K * kTWin;  // windows of slab 0
  const unsigned char* op = ovf + uniform64(active ? ovf_base[gwave] : 0) * kTWin;     // next overflow window
  unsigned base = lds0 + lane_c;

  // the e-steps of one window of row-set KI: `ring` = read it from the row-set's asm-owned registers, else from (off, val)
  auto steps_of = [&](auto kc, auto ring, unsigned off, float val, int c) {
    constexpr int KI = decltype(kc)::value;
    constexpr bool RING = decltype(ring)::value;
    typedef __attribute__((address_space(3))) const f4* lds_p;
    // N steps from T0 on as one batch: operands, then all LDS reads, then the FMAs (one LDS round trip per batch)
    auto batch = [&](auto t0c, auto nc) {
      constexpr int T0 = decltype(t0c)::value;
      constexpr int N = decltype(nc)::value;
      unsigned adr[N];
      float v[N];
      f4 q[N];
      [&]<int... I>(std::integer_sequence<int, I...>) {
        if constexpr (RING) (t_step_operands<KI, T0 + I>(base, adr[I], v[I]), ...);
        else (t_step_operands_r<T0 + I>(off, val, base, adr[I], v[I]), ...);
      }(std::make_integer_sequence<int, N>{});
      if constexpr (MODE & 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc[KI][0] += v[i] + (float)adr[i];
      } else {
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = *(lds_p)adr[i];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[KI][u] = fmaf(v[i], q[i][u], acc[KI][u]);
      }
    };
    using I2 = std::integral_constant<int, 2>;
    using I8 = std::integral_constant<int, 8>;
#define MU_T(T) std::integral_constant<int, T>{}
    // (batches: a count that is not a multiple runs padding steps - the slots are (0.0, offset 0))
    if (c > 4) {
      batch(MU_T(0), I8{});  // 8 entries per row and slab on the bench matrices: the common case in one round trip
      if (c > 8) {
        batch(MU_T(8), I2{});
        if (c > 10) {
          batch(MU_T(10), I2{});
          if (c > 12) {
            batch(MU_T(12), I2{});
            if (c > 14) batch(MU_T(14), I2{});
          }
        }
      }
    } else if (c > 0) {
      batch(MU_T(0), I2{});
      if (c > 2) batch(MU_T(2), I2{});
    }
#undef MU_T
  };

  unsigned long long cnt = 0ull, cnt_hi = 0ull;
  // one visit: row-set KI in the current slab
  auto visit = [&](auto kc) {
    constexpr int KI = decltype(kc)::value;
    if constexpr (KI < K) {
      const int c = KI < 8 ? (int)((cnt >> (8 * (KI & 7))) & 0xffull) : (int)((cnt_hi >> (8 * (KI & 7))) & 0xffull);
      t_wait<2 * (K - 1)>();  // this row-set's window: everything but the K - 1 younger requests has returned
      steps_of(kc, std::true_type{}, 0u, 0.f, c < 16 ? c : 16);
      t_request<KI>(wp, lane4, lane2);  // (after the steps: they read the registers the request overwrites)
      if (!(MODE & 4) && c > 16) {  // rare: the visit continues in the overflow stream (full waits)
        for (int rest = c - 16; rest > 0; rest -= 16) {
          // (from asm, with its own full wait: a load hipcc knows about makes its wait-count pass guard the registers
          //  involved with `s_waitcnt vmcnt(0)` all over the loop - which drains the windows in flight)
          float oval;
          unsigned ooff;
          asm volatile(
              "global_load_dword %0, %2, %4\n\t"
              "global_load_ushort %1, %3, %4 offset:256\n\t"
              "s_waitcnt vmcnt(0)"
              : "=&v"(oval), "=&v"(ooff)
              : "v"(lane4), "v"(lane2), "s"(op)
              : "memory");
          op += kTWin;
          steps_of(kc, std::false_type{}, ooff, oval, rest < 16 ? rest : 16);
        }
      }
    }
  };
  auto request0 = [&](auto kc) {
    constexpr int KI = decltype(kc)::value;
    if constexpr (KI < K) t_request<KI>(wp, lane4, lane2);
  };
#define MU_ALL_K(f)                          \
  f(std::integral_constant<int, 0>{});       \
  f(std::integral_constant<int, 1>{});       \
  f(std::integral_constant<int, 2>{});       \
  f(std::integral_constant<int, 3>{});       \
  f(std::integral_constant<int, 4>{});       \
  f(std::integral_constant<int, 5>{});       \
  f(std::integral_constant<int, 6>{});       \
  f(std::integral_constant<int, 7>{});       \
  f(std::integral_constant<int, 8>{});       \
  f(std::integral_constant<int, 9>{});

  // prologue: the windows of slab 0, one request per row-set
  cnt = counts_of(0, 0);
  cnt_hi = K > 8 ? counts_of(0, 1) : 0ull;
  unsigned long long next_cnt = n_slabs > 1 ? counts_of(1, 0) : 0ull;
  unsigned long long next_hi = (K > 8 && n_slabs > 1) ? counts_of(1, 1) : 0ull;
  MU_ALL_K(request0)
  __syncthreads();
  for (int s = 0;; ++s) {
    wp += K * kTWin;  // the windows of slab s + 1 (the last slab requests the slack behind the stream)
    MU_ALL_K(visit)
    if (s + 1 >= n_slabs) break;
    __syncthreads();
    base = lds0 + lane_c + (unsigned)((s + 1) & 1) * (unsigned)kTSlabBytes;
    cnt = next_cnt;
    cnt_hi = next_hi;
    next_cnt = s + 2 < n_slabs ? counts_of(s + 2, 0) : 0ull;
    next_hi = (K > 8 && s + 2 < n_slabs) ? counts_of(s + 2, 1) : 0ull;
  }
#undef MU_ALL_K
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int g = lane >> 4, c16 = lane & 15;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int64_t p = (gwave * K + k) * 4 + g;
    if (active && p < n_pos) {
      const int64_t row = perm ? (int64_t)perm[p] : p;
      if (row >= 0) *reinterpret_cast<f4*>(Y + row * 64 + 4 * c16) = acc[k];
    }
  }
}

}  // namespace

extern "C" {

int mu_spmm_ell64_f32(int k_sets, int waves, int64_t n_pos, int64_t n_cols, const void* d_hdr, const void* d_ent,
                      const int64_t* d_ovf_base, const void* d_ovf, const int32_t* d_perm, const float* d_Q, float* d_Y,
                      void* stream) {
  MU_REQUIRE(k_sets == 6 || k_sets == 8 || k_sets == 9 || k_sets == 10, "row-sets per wave: 6, 8, 9 or 10");
  MU_REQUIRE(waves >= 1 && waves <= kTMaxWaves, "row-owning waves per workgroup: 1 .. 15");
  MU_REQUIRE(n_pos >= 0 && n_cols > 0 && n_cols * 256 < ((int64_t)1 << 32), "shape out of range");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_hdr && d_ent && d_ovf_base && d_ovf && d_Q && d_Y, "null pointer");
  const int64_t n_slabs = (n_cols + kTSlabRows - 1) / kTSlabRows;
  const int64_t n_waves = (n_pos + 4 * k_sets - 1) / (4 * k_sets);
  const int64_t wgs = (n_waves + waves - 1) / waves;
  const int mode = mu_tune_get("ell_mode") & 7;
  hipStream_t st = (hipStream_t)stream;
#define MU_GO(KK, MD)                                                                                             \
  hipLaunchKernelGGL((k_spmm_ell64<KK, MD>), dim3((unsigned)wgs), dim3(64 * (waves + 1)), 0, st, n_pos, n_cols,    \
                     (int)n_slabs, (const unsigned long long*)d_hdr, (const unsigned char*)d_ent, d_ovf_base,       \
                     (const unsigned char*)d_ovf, d_perm, d_Q, d_Y)
#define MU_GO_K(KK)            \
  do {                         \
    if (mode == 1) MU_GO(KK, 1); \
    else if (mode == 2) MU_GO(KK, 2); \
    else if (mode == 3) MU_GO(KK, 3); \
    else if (mode == 4) MU_GO(KK, 4); \
    else if (mode == 7) MU_GO(KK, 7); \
    else MU_GO(KK, 0);         \
  } while (0)
  if (k_sets == 6) MU_GO_K(6);
  else if (k_sets == 8) MU_GO_K(8);
  else if (k_sets == 9) MU_GO_K(9);
  else MU_GO_K(10);
#undef MU_GO_K
#undef MU_GO
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
