// Stable CSR transpose: builds the CSR of X^T (the device "CSC copy") so that
// Z = X^T * Y runs through the same gather SpMM as Y = X * Q.  Replaces the
// implicit transpose inside scipy's rmatvec (A.T.conj() @ y, _svds.py:441-466).
//
// Algorithm (deterministic, no global atomics):
//   1. slab pointers (sweep.hpp)
//   2. count:  workgroup g sweeps its contiguous row range slab by slab and counts the
//              entries of every column in LDS -> cnt[g][col]
//   3. base:   per column, exclusive prefix of cnt over g (in place) + column total
//   4. scan:   column totals -> t_indptr
//   5. fill:   same sweep; rows are taken in batches of 16 (one per wave).  Inside a batch
//              the order of the <=16 entries that hit one column is fixed by a 16-bit wave
//              mask in LDS (rank = popcount of lower waves), so every output row of X^T
//              lists X's row ids in ascending order.
#include "sweep.hpp"

__global__ __launch_bounds__(kSweepThreads) void k_col_count(
    int64_t n_rows, int64_t n_cols, int64_t S, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const int64_t* __restrict__ sp,
    uint32_t* __restrict__ cnt) {
  __shared__ uint32_t bins[kSlab];
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) sweep_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  __syncthreads();
  const int64_t r0 = s_r[0], r1 = s_r[1];
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  for (int64_t s = 0; s < S; ++s) {
    for (int t = threadIdx.x; t < kSlab; t += kSweepThreads) bins[t] = 0u;
    __syncthreads();
    const int32_t cbase = (int32_t)(s * kSlab);
    for (int64_t row = r0 + wave; row < r1; row += kSweepWaves) {
      const int64_t lo = sp[row * (S + 1) + s], hi = sp[row * (S + 1) + s + 1];
      for (int64_t p = lo + lane; p < hi; p += 256) {
        int32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t q = p + 64 * u;
          c[u] = (q < hi) ? indices[q] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c[u] >= 0) atomicAdd(&bins[c[u] - cbase], 1u);
      }
    }
    __syncthreads();
    const int64_t ncol_here = (n_cols - (int64_t)cbase) < kSlab ? (n_cols - (int64_t)cbase) : kSlab;
    uint32_t* dst = cnt + (int64_t)g * n_cols + cbase;
    for (int t = threadIdx.x; t < ncol_here; t += kSweepThreads) dst[t] = bins[t];
    __syncthreads();
  }
}

// cnt[g][c] <- sum_{g' < g} cnt[g'][c];  coltot[c] = sum_g cnt[g][c]
__global__ __launch_bounds__(256) void k_col_base(int64_t n_cols, int G, uint32_t* __restrict__ cnt,
                                                  int64_t* __restrict__ coltot) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  uint32_t run = 0;
  for (int g = 0; g < G; ++g) {
    const uint32_t t = cnt[(int64_t)g * n_cols + c];
    cnt[(int64_t)g * n_cols + c] = run;
    run += t;
  }
  coltot[c] = (int64_t)run;
}

template <typename T>
__global__ __launch_bounds__(kSweepThreads) void k_transpose_fill(
    int64_t n_rows, int64_t n_cols, int64_t S, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const T* __restrict__ values,
    const int64_t* __restrict__ sp, const int64_t* __restrict__ t_indptr,
    const uint32_t* __restrict__ base, int32_t* __restrict__ t_indices, T* __restrict__ t_values) {
  __shared__ uint32_t cur[kSlab];       // entries of the column already emitted by this workgroup
  __shared__ uint32_t mask[kSlab / 2];  // per column 16-bit wave mask, two columns per word
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) sweep_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  __syncthreads();
  const int64_t r0 = s_r[0], r1 = s_r[1];
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t wbit = 1u << wave;
  const uint32_t* mybase = base + (int64_t)g * n_cols;

  for (int64_t s = 0; s < S; ++s) {
    for (int t = threadIdx.x; t < kSlab; t += kSweepThreads) cur[t] = 0u;
    for (int t = threadIdx.x; t < kSlab / 2; t += kSweepThreads) mask[t] = 0u;
    __syncthreads();
    const int32_t cbase = (int32_t)(s * kSlab);
    for (int64_t rb = r0; rb < r1; rb += kSweepWaves) {  // uniform across the workgroup
      const int64_t row = rb + wave;
      int64_t lo = 0, hi = 0;
      if (row < r1) {
        lo = sp[row * (S + 1) + s];
        hi = sp[row * (S + 1) + s + 1];
      }
      // phase 1: announce (columns inside one row are distinct)
      for (int64_t p = lo + lane; p < hi; p += 64) {
        const int c = indices[p] - cbase;
        atomicOr(&mask[c >> 1], wbit << (16 * (c & 1)));
      }
      __syncthreads();
      // phase 2: emit at  t_indptr[col] + base[g][col] + cur[col] + rank
      for (int64_t p = lo + lane; p < hi; p += 64) {
        const int32_t cg = indices[p];
        const int c = cg - cbase;
        const uint32_t m = (mask[c >> 1] >> (16 * (c & 1))) & 0xffffu;
        const int rank = __popc(m & (wbit - 1u));
        const int64_t pos = t_indptr[cg] + (int64_t)mybase[cg] + (int64_t)cur[c] + rank;
        t_indices[pos] = (int32_t)row;
        t_values[pos] = values[p];
      }
      __syncthreads();
      // phase 3: the highest wave of each column advances the cursor and clears its mask
      for (int64_t p = lo + lane; p < hi; p += 64) {
        const int c = indices[p] - cbase;
        const uint32_t m = (mask[c >> 1] >> (16 * (c & 1))) & 0xffffu;
        if ((m >> wave) == 1u) {  // no higher wave set
          cur[c] += __popc(m);
          atomicAnd(&mask[c >> 1], ~(0xffffu << (16 * (c & 1))));
        }
      }
      __syncthreads();
    }
  }
}

extern "C" {

size_t mu_csr_transpose_worksize(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  (void)nnz;
  const int64_t S = num_slabs(n_cols);
  const int G = sweep_grid();
  size_t b = 0;
  b += ((size_t)(n_rows * (S + 1)) * sizeof(int64_t) + 255) & ~(size_t)255;  // slab pointers
  b += ((size_t)G * (size_t)n_cols * sizeof(uint32_t) + 255) & ~(size_t)255;  // cnt / base
  b += ((size_t)n_cols * sizeof(int64_t) + 255) & ~(size_t)255;               // column totals
  return b + 256;
}

int mu_csr_transpose(int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz,
                     const int64_t* d_indptr, const int32_t* d_indices, const void* d_values,
                     int64_t* d_t_indptr, int32_t* d_t_indices, void* d_t_values, void* d_work,
                     size_t work_bytes, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "negative size");
  MU_REQUIRE(d_indptr && d_t_indptr, "null pointer");
  MU_REQUIRE(n_rows < (int64_t)1 << 31, "row ids must fit int32");
  MU_REQUIRE(d_work && work_bytes >= mu_csr_transpose_worksize(n_rows, n_cols, nnz),
             "work buffer too small");
  hipStream_t st = (hipStream_t)stream;
  if (n_cols == 0) return MU_OK;
  if (n_rows == 0 || nnz == 0) {
    MU_CHECK_HIP(hipMemsetAsync(d_t_indptr, 0, sizeof(int64_t) * (n_cols + 1), st));
    return MU_OK;
  }
  const int64_t S = num_slabs(n_cols);
  const int G = sweep_grid();
  char* w = (char*)d_work;
  int64_t* sp = (int64_t*)w;
  w += ((size_t)(n_rows * (S + 1)) * sizeof(int64_t) + 255) & ~(size_t)255;
  uint32_t* cnt = (uint32_t*)w;
  w += ((size_t)G * (size_t)n_cols * sizeof(uint32_t) + 255) & ~(size_t)255;
  int64_t* coltot = (int64_t*)w;

  int rc = launch_slab_ptr(n_rows, n_cols, d_indptr, d_indices, sp, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_col_count, dim3(G), dim3(kSweepThreads), 0, st, n_rows, n_cols, S, d_indptr,
                     d_indices, sp, cnt);
  MU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_col_base, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, n_cols, G,
                     cnt, coltot);
  MU_CHECK_LAUNCH();
  rc = mu_exclusive_scan_i64(n_cols, coltot, d_t_indptr, stream);
  if (rc) return rc;
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_transpose_fill<float>, dim3(G), dim3(kSweepThreads), 0, st, n_rows, n_cols,
                       S, d_indptr, d_indices, (const float*)d_values, sp, d_t_indptr, cnt,
                       d_t_indices, (float*)d_t_values);
  else
    hipLaunchKernelGGL(k_transpose_fill<double>, dim3(G), dim3(kSweepThreads), 0, st, n_rows, n_cols,
                       S, d_indptr, d_indices, (const double*)d_values, sp, d_t_indptr, cnt,
                       d_t_indices, (double*)d_t_values);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
