// CSR SpMM  Y[n x B] = X[n x d] * Q[d x B]  (f32 values / f32 dense / f32 accumulate).
// Replaces the csr_matvec / csr_matvecs calls that ARPACK's reverse-communication
// loop issues through scipy.sparse.linalg.svds (/root/reference/muon/_atac/tools.py:53,
// scipy _svds.py:441-466,516).  The transposed product runs through the same kernel on
// the device CSC copy (transpose.hip).
//
// v1 layout: one wave per row; the wave loads 64 (column, value) pairs with one
// coalesced access each, then walks them, every lane owning one of the B dense
// columns (B=64) so that each gathered Q row is a single coalesced 256-byte read.
#include <cstdlib>
#include "common.hpp"

template <typename T>
__device__ __forceinline__ T readlane_t(T v, int src);
template <>
__device__ __forceinline__ float readlane_t<float>(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
template <>
__device__ __forceinline__ double readlane_t<double>(double v, int src) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

template <typename T, int B>
__global__ __launch_bounds__(256) void k_spmm_rowwave(int64_t n_rows,
                                                      const int64_t* __restrict__ indptr,
                                                      const int32_t* __restrict__ indices,
                                                      const T* __restrict__ values,
                                                      const T* __restrict__ Q,
                                                      T* __restrict__ Y, int accumulate) {
  constexpr int NPS = 64 / B;  // stored entries consumed per step
  const int lane = threadIdx.x & 63;
  const int sub = lane % B;
  const int grp = lane / B;
  const int64_t wave0 = uniform64(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t row = wave0; row < n_rows; row += n_waves) {
    const int64_t lo = uniform64(indptr[row]), hi = uniform64(indptr[row + 1]);
    T acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int64_t p0 = lo; p0 < hi; p0 += 64) {
      const int64_t p = p0 + lane;
      const bool in = p < hi;
      const int32_t c = in ? indices[p] : 0;
      const T v = in ? values[p] : (T)0;
      const int cnt = (hi - p0) < 64 ? (int)(hi - p0) : 64;
      for (int j = 0; j < cnt; j += 4 * NPS) {
        // four independent gathers in flight; entries past cnt carry v == 0 and c == 0
        int32_t cj[4];
        T vj[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if constexpr (NPS == 1) {
            const int src = (j + u) & 63;
            cj[u] = __builtin_amdgcn_readlane(c, src);
            vj[u] = readlane_t<T>(v, src);
          } else {
            const int src = (j + u * NPS + grp) & 63;
            cj[u] = __shfl(c, src, 64);
            vj[u] = __shfl(v, src, 64);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = Q[(int64_t)cj[u] * B + sub];
        acc0 = fma(vj[0], q[0], acc0);
        acc1 = fma(vj[1], q[1], acc1);
        acc2 = fma(vj[2], q[2], acc2);
        acc3 = fma(vj[3], q[3], acc3);
      }
    }
    T acc = (acc0 + acc1) + (acc2 + acc3);
#pragma unroll
    for (int off = B; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (grp == 0) {
      T* y = Y + row * B + sub;
      *y = accumulate ? (*y + acc) : acc;
    }
  }
}


// ---------------------------------------------------------------------------------
// v2 (B = 64): Q column slabs staged in LDS.
//
// The v1 kernel gathers a 256-byte Q row per stored entry through L1/L2; at d = 200k the
// 51 MB of Q only fits the Infinity Cache, and the kernel runs at the fabric's gather
// rate (~7 TB/s of re-read traffic for 6 GB of matrix).  Here the columns of X are cut
// into slabs of kQSlab columns whose Q rows (kQSlab x 256 B = 128 KiB) are copied into LDS
// once per workgroup; a workgroup owns 64*K consecutive rows and keeps their K x float4
// accumulators in registers while it sweeps the slabs, so the gather runs at the LDS rate
// (ds_read_b128, 256 B/clk/CU = one entry per clock per CU) and HBM only streams
// (column, value) pairs.
//
// Lane layout: a wave is four 16-lane groups; group g walks row 4k+g, lane `sub` of the
// group owns dense columns 4 sub..4 sub+3.  A group's 16 lanes load the next 16 entries of
// their row with one coalesced access; entry e is then broadcast inside the group with DPP
// row_newbcast (no LDS traffic, no readlane), so each ds_read_b128 serves four rows.
// Rows are sorted by column, so the entries of a row that fall into the current slab are a
// prefix of the not-yet-consumed entries; a per-row cursor replaces any search.
// ---------------------------------------------------------------------------------
constexpr int kQSlab = 256;  // columns per slab: 256 x 256 B = 64 KiB of Q, double buffered
constexpr int kLdsThreads = 1024;
constexpr int kKMax = 8;
constexpr int kStage = (kQSlab * 16) / kLdsThreads;  // float4 per thread per slab copy (= 4)

template <int E>
__device__ __forceinline__ int dpp_bcast_i(int x) {
#ifdef MU_SPMM_NO_DPP
  return __shfl(x, (threadIdx.x & 48) | E, 64);
#else
  return __builtin_amdgcn_update_dpp(0, x, 0x150 + E, 0xf, 0xf, false);  // row_newbcast:E
#endif
}
template <int E>
__device__ __forceinline__ float dpp_bcast_f(float x) {
  return __builtin_bit_cast(float, dpp_bcast_i<E>(__builtin_bit_cast(int, x)));
}

// Entries E..E+3 of every group's window: four independent ds_read_b128 in flight per
// uniform branch.  Lanes whose group has fewer entries carry a = 0, v = 0 (a harmless
// broadcast read of Q row 0 of the slab and an FMA with zero).
template <int E>
__device__ __forceinline__ void lds_quad(const char* qs, int a, float v, int sub16, float4& acc) {
  const int a0 = dpp_bcast_i<E>(a), a1 = dpp_bcast_i<E + 1>(a);
  const int a2 = dpp_bcast_i<E + 2>(a), a3 = dpp_bcast_i<E + 3>(a);
  const float4 q0 = *reinterpret_cast<const float4*>(qs + a0 + sub16);
  const float4 q1 = *reinterpret_cast<const float4*>(qs + a1 + sub16);
  const float4 q2 = *reinterpret_cast<const float4*>(qs + a2 + sub16);
  const float4 q3 = *reinterpret_cast<const float4*>(qs + a3 + sub16);
  const float v0 = dpp_bcast_f<E>(v), v1 = dpp_bcast_f<E + 1>(v);
  const float v2 = dpp_bcast_f<E + 2>(v), v3 = dpp_bcast_f<E + 3>(v);
  acc.x = fmaf(v0, q0.x, acc.x); acc.y = fmaf(v0, q0.y, acc.y);
  acc.z = fmaf(v0, q0.z, acc.z); acc.w = fmaf(v0, q0.w, acc.w);
  acc.x = fmaf(v1, q1.x, acc.x); acc.y = fmaf(v1, q1.y, acc.y);
  acc.z = fmaf(v1, q1.z, acc.z); acc.w = fmaf(v1, q1.w, acc.w);
  acc.x = fmaf(v2, q2.x, acc.x); acc.y = fmaf(v2, q2.y, acc.y);
  acc.z = fmaf(v2, q2.z, acc.z); acc.w = fmaf(v2, q2.w, acc.w);
  acc.x = fmaf(v3, q3.x, acc.x); acc.y = fmaf(v3, q3.y, acc.y);
  acc.z = fmaf(v3, q3.z, acc.z); acc.w = fmaf(v3, q3.w, acc.w);
}

__global__ __launch_bounds__(kLdsThreads) void k_spmm_lds64(
    int64_t n_rows, int64_t n_cols, int K, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const float* __restrict__ values,
    const float* __restrict__ Q, float* __restrict__ Y, int mode) {
  __shared__ float4 qs[2][kQSlab * 16];  // 2 x 64 KiB; Q row c of a slab at [16 c .. 16 c + 15]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 15, g = lane >> 4;
  const int sub16 = sub * 16;
  const int64_t rb0 = (int64_t)blockIdx.x * (64 * K);
  const int64_t rb1 = (rb0 + 64 * K) < n_rows ? (rb0 + 64 * K) : n_rows;
  const int64_t base = indptr[rb0];  // 32-bit cursors are relative to the workgroup's first entry
  const int32_t* __restrict__ ind_b = indices + base;
  const float* __restrict__ val_b = values + base;
  const float4* __restrict__ Q4 = reinterpret_cast<const float4*>(Q);
  const int64_t q4_total = n_cols * 16;

  float4 acc[kKMax];
  int c[kKMax];
  float v[kKMax];
  // cursor / row end of (row-set k, group g) live in lane 16 g + k of ONE register each and are
  // broadcast inside the group with DPP when needed (saves 14 VGPRs over per-k registers)
  int curv = 0, rendv = 0;
  {
    const int64_t row = rb0 + ((int64_t)wave * K + sub) * 4 + g;
    const bool ok = (sub < K) && (row < rb1);
    curv = ok ? (int)(indptr[row] - base) : 0;
    rendv = ok ? (int)(indptr[row + 1] - base) : 0;
  }
#pragma unroll
  for (int k = 0; k < kKMax; ++k) acc[k] = float4{0.f, 0.f, 0.f, 0.f};
#define MU_FETCH(k)                                              \
  {                                                              \
    const int idx_ = dpp_bcast_i<k>(curv) + sub;                 \
    const bool in_ = idx_ < dpp_bcast_i<k>(rendv);               \
    c[k] = in_ ? ind_b[idx_] : 0x7fffffff;                       \
    v[k] = in_ ? val_b[idx_] : 0.f;                              \
  }
  MU_FETCH(0) MU_FETCH(1) MU_FETCH(2) MU_FETCH(3) MU_FETCH(4) MU_FETCH(5) MU_FETCH(6) MU_FETCH(7)

  // Q slab copy: LDS-DMA (global_load_lds, 16 B per lane => 1 KiB per wave instruction lands
  // contiguously at the wave-uniform LDS base); no VGPR staging, no ds_write.
  auto slab_dma = [&](int64_t s0, int buf) {
#pragma unroll
    for (int u = 0; u < kStage; ++u) {
      const int piece = wave * kStage + u;                    // 1 KiB piece of the 64 KiB slab
      int64_t i = s0 * 16 + piece * 64 + lane;                // float4 index into Q
      if (i >= q4_total) i = q4_total - 1;                    // tail slab: clamp (never consumed)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(Q4 + i),
          (__attribute__((address_space(3))) void*)(&qs[buf][piece * 64]), 16, 0, 0);
    }
  };
  slab_dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int buf = 0;
  for (int64_t s0 = 0; s0 < n_cols; s0 += kQSlab, buf ^= 1) {
    const bool has_next = (s0 + kQSlab) < n_cols;
    if (has_next && !(mode & 2)) slab_dma(s0 + kQSlab, buf ^ 1);  // in flight while this slab is consumed
    const int s_lo = (int)s0;
    const int s_hi = (int)((s0 + kQSlab) < n_cols ? (s0 + kQSlab) : n_cols);
    const char* qbytes = reinterpret_cast<const char*>(qs[buf]);
#define MU_WINDOW(k)                                                                         \
  {                                                                                          \
    const bool valid = c[k] < s_hi; /* sorted rows: valid entries are a prefix */            \
    const unsigned long long m = __ballot(valid);                                            \
    const int cnt = __popc((unsigned)(m >> (16 * g)) & 0xffffu);                             \
    const unsigned mm = (unsigned)m | (unsigned)(m >> 32);                                   \
    const unsigned any16 = (mm | (mm >> 16)) & 0xffffu; /* bit e: some group has entry e */  \
    const int a = valid ? ((c[k] - s_lo) << 8) : 0;                                          \
    const float vv = valid ? v[k] : 0.f;                                                     \
    curv += (sub == k) ? cnt : 0;                                                            \
    /* next window right away: the prefetch for the next slab (or for the overflow pass) */  \
    MU_FETCH(k)                                                                              \
    /* entries 0..7 unconditionally: eight ds_read_b128 in flight in one basic block */      \
    if (!(mode & 1)) {                                                                       \
    lds_quad<0>(qbytes, a, vv, sub16, acc[k]);                                               \
    lds_quad<4>(qbytes, a, vv, sub16, acc[k]);                                               \
    if (any16 & 0x0f00u) lds_quad<8>(qbytes, a, vv, sub16, acc[k]);                          \
    if (any16 & 0xf000u) lds_quad<12>(qbytes, a, vv, sub16, acc[k]);                         \
    } else { acc[k].x += vv + (float)a; }                                                    \
    /* a group that consumed all 16 may have more entries of this slab: revisit this row  */ \
    /* set after the others so that the window just requested has time to arrive          */ \
    const bool more = (cnt == 16) && (dpp_bcast_i<k>(curv) < dpp_bcast_i<k>(rendv));         \
    if (__ballot(more)) again |= 1u << k;                                                    \
  }
#define MU_PASS(k) if (pend & (1u << k)) MU_WINDOW(k)
    unsigned pend = (1u << K) - 1u;  // row-sets with (possibly) unconsumed entries in this slab
    while (pend) {
      unsigned again = 0;
      MU_PASS(0) MU_PASS(1) MU_PASS(2) MU_PASS(3) MU_PASS(4) MU_PASS(5) MU_PASS(6) MU_PASS(7)
      pend = again;
    }
    // The four DMA pieces of this wave are older than the >= 2*K window loads issued above
    // (vmcnt retires loads in order), so allowing 2*K outstanding loads proves the DMA landed
    // without draining the prefetched windows.
    switch (K) {
      case 8: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    }
    if (!(mode & 4)) __syncthreads();  // next slab visible; everyone finished reading this one
  }
#pragma unroll
  for (int k = 0; k < kKMax; ++k) {
    const int64_t row = rb0 + ((int64_t)wave * K + k) * 4 + g;
    if ((k < K) && (row < rb1)) *reinterpret_cast<float4*>(Y + row * 64 + sub * 4) = acc[k];
  }
}

extern "C" int mu_spmm_f32(int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                           const int32_t* d_indices, const float* d_values, const float* d_Q, int B,
                           float* d_Y, int accumulate, void* stream) {
  MU_REQUIRE(B == 16 || B == 32 || B == 64, "B must be 16, 32 or 64");
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative shape");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_Q && d_Y, "null pointer");
  int64_t blocks = (n_rows + 3) / 4;
  const int64_t cap = (int64_t)mu_num_cus() * 16;
  if (blocks > cap) blocks = cap;
  hipStream_t st = (hipStream_t)stream;
  if (B == 64 && !accumulate && n_cols <= ((int64_t)1 << 21)) {
    // LDS-slab kernel: 64*K rows per workgroup, K chosen so that the grid fills the CUs
    // smallest number of full-chip rounds R whose row blocks fit the register budget
    const int64_t cus = mu_num_cus();
    int K = kKMax;
    for (int64_t R = 1; R <= 64; ++R) {
      const int64_t k = (n_rows + 64 * cus * R - 1) / (64 * cus * R);
      if (k <= kKMax) { K = (int)(k < 1 ? 1 : k); break; }
    }
    const int64_t wgs = (n_rows + 64 * K - 1) / (64 * K);
    static const int mode = getenv("MU_SPMM_MODE") ? atoi(getenv("MU_SPMM_MODE")) : 0;  // ablation only
    hipLaunchKernelGGL(k_spmm_lds64, dim3((unsigned)wgs), dim3(kLdsThreads), 0, st, n_rows, n_cols,
                       K, d_indptr, d_indices, d_values, d_Q, d_Y, mode);
    MU_CHECK_LAUNCH();
    return MU_OK;
  }
  switch (B) {
    case 64:
      hipLaunchKernelGGL((k_spmm_rowwave<float, 64>), dim3((unsigned)blocks), dim3(256), 0, st, n_rows,
                         d_indptr, d_indices, d_values, d_Q, d_Y, accumulate);
      break;
    case 32:
      hipLaunchKernelGGL((k_spmm_rowwave<float, 32>), dim3((unsigned)blocks), dim3(256), 0, st, n_rows,
                         d_indptr, d_indices, d_values, d_Q, d_Y, accumulate);
      break;
    default:
      hipLaunchKernelGGL((k_spmm_rowwave<float, 16>), dim3((unsigned)blocks), dim3(256), 0, st, n_rows,
                         d_indptr, d_indices, d_values, d_Q, d_Y, accumulate);
      break;
  }
  MU_CHECK_LAUNCH();
  return MU_OK;
}

extern "C" int mu_spmm_f64(int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                           const int32_t* d_indices, const double* d_values, const double* d_Q, int B,
                           double* d_Y, int accumulate, void* stream) {
  MU_REQUIRE(B == 16 || B == 32 || B == 64, "B must be 16, 32 or 64");
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative shape");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_Q && d_Y, "null pointer");
  int64_t blocks = (n_rows + 3) / 4;
  const int64_t cap = (int64_t)mu_num_cus() * 16;
  if (blocks > cap) blocks = cap;
  hipStream_t st = (hipStream_t)stream;
  switch (B) {
    case 64:
      hipLaunchKernelGGL((k_spmm_rowwave<double, 64>), dim3((unsigned)blocks), dim3(256), 0, st, n_rows,
                         d_indptr, d_indices, d_values, d_Q, d_Y, accumulate);
      break;
    case 32:
      hipLaunchKernelGGL((k_spmm_rowwave<double, 32>), dim3((unsigned)blocks), dim3(256), 0, st, n_rows,
                         d_indptr, d_indices, d_values, d_Q, d_Y, accumulate);
      break;
    default:
      hipLaunchKernelGGL((k_spmm_rowwave<double, 16>), dim3((unsigned)blocks), dim3(256), 0, st, n_rows,
                         d_indptr, d_indices, d_values, d_Q, d_Y, accumulate);
      break;
  }
  MU_CHECK_LAUNCH();
  return MU_OK;
}
