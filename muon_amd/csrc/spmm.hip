// CSR SpMM  Y[n x B] = X[n x d] * Q[d x B]  (f32 values / f32 dense / f32 accumulate).
// Replaces the csr_matvec / csr_matvecs calls that ARPACK's reverse-communication
// loop issues through scipy.sparse.linalg.svds (/root/reference/muon/_atac/tools.py:53,
// scipy _svds.py:441-466,516).  The transposed product runs through the same kernel on
// the device CSC copy (transpose.hip).
//
// (The LSI iteration runs on the row-stream kernel of spmm_win.hip; this one is the general
// fallback: f64 values - MOFA in f64 -, accumulate, any width.)
// Layout: one wave per row; the wave loads 64 (column, value) pairs with one
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
