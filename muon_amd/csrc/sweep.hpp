// Slab sweep machinery shared by the TF-IDF reductions and the CSR transpose.
//
// A CSR with sorted rows is cut into column slabs of kSlab columns.  slab_ptr
// (sp[row*(S+1)+s] = first position of `row` whose column >= s*kSlab) lets any
// workgroup find the contiguous piece of a row that falls into a slab without
// searching, so per-column state for one slab can live in LDS while the
// workgroup walks its rows: per-column sums / counts / cursors never touch
// global atomics (MI355X device-scope atomics are performed at the memory side
// and would be ~10x slower than the HBM stream they accompany).
#pragma once
#include "common.hpp"

constexpr int kSlab = 8192;        // columns per slab
constexpr int kSweepThreads = 1024;
constexpr int kSweepWaves = kSweepThreads / 64;

inline int64_t num_slabs(int64_t n_cols) { return (n_cols + kSlab - 1) / kSlab; }
// workgroups of the sweep kernels: two 1024-thread groups per CU
inline int sweep_grid() { return 2 * mu_num_cus(); }

// rows [r0, r1) owned by workgroup g of G: contiguous, balanced by nnz
__device__ __forceinline__ void sweep_row_range(const int64_t* indptr, int64_t n_rows, int g, int G,
                                                int64_t& r0, int64_t& r1) {
  const int64_t nnz = indptr[n_rows];
  auto cut = [&](int k) -> int64_t {
    if (k <= 0) return 0;
    if (k >= G) return n_rows;
    int64_t key = (nnz / G) * k + ((nnz % G) * k) / G;
    int64_t r = lower_bound_i64(indptr, 0, n_rows, key);
    return r > n_rows ? n_rows : r;
  };
  r0 = cut(g);
  r1 = cut(g + 1);
}

int launch_slab_ptr(int64_t n_rows, int64_t n_cols, const int64_t* indptr, const int32_t* indices,
                    int64_t* sp, hipStream_t stream);
