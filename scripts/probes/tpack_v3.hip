// X^T built straight from the CSR of X: as the row stream the SpMM reads (csrc/spmm_win.hip) or as a
// plain CSR ("transpose-pack"; the name is r01's, when the target was a chunked packed copy).
//
// Z = X^T * Y of the block Lanczos iteration (the rmatvec side of scipy svds, _svds.py:441-466,
// reached from /root/reference/muon/_atac/tools.py:53) runs through the same SpMM as Y = X * Q;
// this file builds its operand without materialising a CSR of X^T first.
// Stable (every output row lists the cells in ascending order => canonical rows, and the f32 sums
// of the SpMM are bit-reproducible), no global atomics.
//
//   1. slab pointers: sp[row][s] = first entry of `row` with column >= s * kTSlab
//   2. count: workgroup g owns a contiguous, nnz-balanced row range and counts the entries of every
//      column in LDS, slab by slab                                           -> cnt[g][col]
//   3. base:  per column, exclusive prefix of cnt over g (in place) and the column total
//                                                      -> caller lays the rows out, scans -> sptr
//   4. fill:  a workgroup stages a (row block x column tile) in LDS sorted by (column, cell) and
//      writes every column's run with consecutive lanes (k_t_fill3 / k_t_fill2 below; the first
//      generation, a bitmap-rank fill with one 8-byte store per pair, ran at the fabric's
//      partial-write rate and is gone)
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int kTSlab = 8192;  // columns per slab of the count pass: 32 KiB of LDS bins (= sweep.hpp kSlab: shared pointers)
constexpr int kTThreads = 1024;
constexpr int kTWaves = kTThreads / 64;

inline int64_t t_slabs(int64_t n_cols) { return (n_cols + kTSlab - 1) / kTSlab; }
// Row blocks (= workgroups of the count and fill sweeps): one per CU at a time (the staged fill needs
// 156 KiB of LDS), in as many rounds as it takes for the expected tile of a block - its rows x the
// widest tile, kF3Cols columns - to fill ~93 % of the staging buffer (~4.2e6 stored entries per block
// at 200k columns; measured at 1e6 x 200k: 1024-column tiles with 1.9e6-entry blocks 110 ms, 640 columns
// with 3.2e6-entry blocks 89 ms).
inline int t_grid(int64_t nnz, int64_t n_cols) {
  const int64_t cus = mu_num_cus();
  // (tune tpack_rows: stored entries per row block in units of 1e5 instead)
  int64_t per_block = (int64_t)(0.93 * 14336 / 640 * (double)(n_cols > 0 ? n_cols : 1));  // kF3CapNarrow / kF3Cols
  if (per_block < 100000) per_block = 100000;
  if (mu_tune_get("tpack_rows") > 0) per_block = 100000ll * mu_tune_get("tpack_rows");
  int64_t rounds = (nnz + per_block * cus - 1) / (per_block * cus);
  if (rounds < 1) rounds = 1;
  if (rounds > 64) rounds = 64;
  return (int)(cus * rounds);
}

// rows [r0, r1) owned by workgroup g of G: contiguous, balanced by nnz
__device__ __forceinline__ void t_row_range(const int64_t* indptr, int64_t n_rows, int g, int G,
                                            int64_t& r0, int64_t& r1) {
  const int64_t nnz = indptr[n_rows];
  auto cut = [&](int k) -> int64_t {
    if (k <= 0) return 0;
    if (k >= G) return n_rows;
    const int64_t key = (nnz / G) * k + ((nnz % G) * k) / G;
    const int64_t r = lower_bound_i64(indptr, 0, n_rows, key);
    return r > n_rows ? n_rows : r;
  };
  r0 = cut(g);
  r1 = cut(g + 1);
}

__global__ __launch_bounds__(256) void k_t_slab_ptr(int64_t n_rows, int64_t S,
                                                    const int64_t* __restrict__ indptr,
                                                    const int32_t* __restrict__ indices,
                                                    int64_t* __restrict__ sp) {
  const int64_t total = n_rows * (S + 1);
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = id / (S + 1);
    const int64_t s = id - row * (S + 1);
    int64_t lo = indptr[row], hi = indptr[row + 1];
    if (s == S) {
      sp[id] = hi;
      continue;
    }
    const int64_t key = s * (int64_t)kTSlab;
    while (lo < hi) {
      const int64_t mid = lo + ((hi - lo) >> 1);
      if ((int64_t)indices[mid] < key) lo = mid + 1; else hi = mid;
    }
    sp[id] = lo;
  }
}

__global__ __launch_bounds__(kTThreads) void k_t_count(int64_t n_rows, int64_t n_cols, int64_t S,
                                                       const int64_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ indices,
                                                       const int64_t* __restrict__ sp,
                                                       uint32_t* __restrict__ cnt) {
  __shared__ uint32_t bins[kTSlab];
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) t_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  __syncthreads();
  const int64_t r0 = s_r[0], r1 = s_r[1];
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  for (int64_t s = 0; s < S; ++s) {
    for (int t = threadIdx.x; t < kTSlab; t += kTThreads) bins[t] = 0u;
    __syncthreads();
    const int32_t cbase = (int32_t)(s * kTSlab);
    for (int64_t row = r0 + wave; row < r1; row += kTWaves) {
      const int64_t lo = sp[row * (S + 1) + s], hi = sp[row * (S + 1) + s + 1];
      for (int64_t p = lo + lane; p < hi; p += 256) {  // four independent chunks in flight per wave
        int32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = (p + 64 * u < hi) ? indices[p + 64 * u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c[u] >= 0) atomicAdd(&bins[c[u] - cbase], 1u);
      }
    }
    __syncthreads();
    const int64_t here = (n_cols - (int64_t)cbase) < kTSlab ? (n_cols - (int64_t)cbase) : kTSlab;
    uint32_t* dst = cnt + (int64_t)g * n_cols + cbase;
    for (int t = threadIdx.x; t < here; t += kTThreads) dst[t] = bins[t];
    __syncthreads();
  }
}

// cnt[g][c] <- sum_{g' < g} cnt[g'][c];  coltot[c] = sum_g cnt[g][c] (also handed to the caller,
// who sorts the output rows by it and lays them out: muon_amd/_backend.py launch_layout)
// The count sweep, software pipelined (r04; csrc/tfidf.hip's k_row_col_sums_pipe has the story): the pieces of a wave's
// strip of 64 rows as one flat sequence of iterations, unconditional clamped loads from asm, two iterations in
// flight, bins of M x 8192 columns (M = 4: 128 KiB of u32, one workgroup per CU).  Same counts.
struct TFlatWalk {
  int l, pb, hi, nrow, step;
  __device__ __forceinline__ bool advance(int lo_l, int hi_l) {
    pb += step;
    while (pb >= hi) {
      if (++l >= nrow) return false;
      pb = __builtin_amdgcn_readlane(lo_l, l);
      hi = __builtin_amdgcn_readlane(hi_l, l);
    }
    return true;
  }
};
template <int N>
__device__ __forceinline__ void tc_wait(int32_t (&c)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : "n"(N) : "memory");
}
template <int M>
__global__ __launch_bounds__(kTThreads) void k_t_count_pipe(int64_t n_rows, int64_t n_cols, int64_t S,
                                                            const int64_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const int64_t* __restrict__ sp,
                                                            uint32_t* __restrict__ cnt) {
  __shared__ uint32_t bins[kTSlab * M];
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) t_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  __syncthreads();
  const int64_t r0 = uniform64(s_r[0]), r1 = uniform64(s_r[1]);
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t wg_base = uniform64(indptr[r0 < n_rows ? r0 : n_rows]);
  const int32_t* ib = indices + wg_base;  // (32-bit byte offsets from here: the host keeps a block under 2^29 entries)
  for (int64_t s = 0; s < S; s += M) {
    for (int t = threadIdx.x; t < kTSlab * M; t += kTThreads) bins[t] = 0u;
    __syncthreads();
    const int32_t cbase = (int32_t)(s * kTSlab);
    const int64_t s_hi = s + M < S ? s + M : S;
    for (int64_t strip = r0 + wave; strip < r1; strip += (int64_t)kTWaves * 64) {
      const int64_t myrow = strip + (int64_t)kTWaves * lane;
      int lo_l = 0, hi_l = 0;
      if (myrow < r1) {
        lo_l = (int)(sp[myrow * (S + 1) + s] - wg_base);
        hi_l = (int)(sp[myrow * (S + 1) + s_hi] - wg_base);
      }
      asm volatile("" ::"v"(lo_l), "v"(hi_l));  // (the compiler's wait for these loads: here, not inside the walk)
      const int64_t left = (r1 - strip + kTWaves - 1) / kTWaves;
      TFlatWalk w{-1, 0, 0, left < 64 ? (int)left : 64, 256};
      int32_t ca[4], cb[4];
      auto load = [&](int32_t (&c)[4], int pb, int hi) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int q = pb + lane + 64 * u;
          q = q < hi ? q : hi - 1;
          asm volatile("global_load_dword %0, %1, %2" : "=&v"(c[u]) : "v"((unsigned)q * 4u), "s"(ib) : "memory");
        }
      };
      auto work = [&](int32_t (&c)[4], int pb, int hi) {
        tc_wait<4>(c);  // (the next iteration's four loads stay in flight)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = pb + lane + 64 * u < hi;  // a lane past the end adds 0 to a bin of its own
          atomicAdd(&bins[ok ? c[u] - cbase : lane], ok ? 1u : 0u);
        }
      };
      if (w.advance(lo_l, hi_l)) {
        int a_pb = w.pb, a_hi = w.hi, b_pb, b_hi;
        load(ca, a_pb, a_hi);
        for (;;) {
          bool more = w.advance(lo_l, hi_l);
          b_pb = more ? w.pb : a_pb, b_hi = more ? w.hi : a_hi;
          load(cb, b_pb, b_hi);
          work(ca, a_pb, a_hi);
          if (!more) break;
          more = w.advance(lo_l, hi_l);
          a_pb = more ? w.pb : b_pb, a_hi = more ? w.hi : b_hi;
          load(ca, a_pb, a_hi);
          work(cb, b_pb, b_hi);
          if (!more) break;
        }
        tc_wait<0>(ca);
        tc_wait<0>(cb);
      }
    }
    __syncthreads();
    const int64_t here = (n_cols - (int64_t)cbase) < kTSlab * M ? (n_cols - (int64_t)cbase) : kTSlab * M;
    uint32_t* dst = cnt + (int64_t)g * n_cols + cbase;
    for (int t = threadIdx.x; t < here; t += kTThreads) dst[t] = bins[t];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_t_base(int64_t n_cols, int G, uint32_t* __restrict__ cnt,
                                                int64_t* __restrict__ coltot,
                                                int64_t* __restrict__ col_nnz) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  uint32_t run = 0;
  for (int g = 0; g < G; ++g) {
    const uint32_t t = cnt[(int64_t)g * n_cols + c];
    cnt[(int64_t)g * n_cols + c] = run;
    run += t;
  }
  coltot[c] = (int64_t)run;
  col_nnz[c] = (int64_t)run;
}

// ---- fill, version 2: LDS-staged, run-coalesced stores -----------------------------------------
// The version above stores every pair on its own: 7.8e8 scattered 8-byte stores, each a 32-byte
// partial write at the memory side (WRITE_SIZE 7.7x the useful bytes on the old transpose), and the
// kernel ran at the partial-write rate of the fabric.  Here a workgroup stages the (row block x
// column slab) tile in LDS, sorted by (column, row), and writes every column's run with
// consecutive lanes, so the stores of a run are one or two full lines.
//   * the tile is sorted with a counting sort whose buckets are (column, wave): a wave owns a
//     contiguous range of the workgroup's rows and walks them IN ORDER, so a plain
//     read-increment-write of its own bucket cursor gives every pair its stable slot - no
//     atomics, no per-batch barriers (4 barriers per tile);
//   * per-row cursors (first entry not yet consumed) live in global memory: no slab pointers;
//   * a tile that does not fit the staging buffer falls back to direct stores (same slots).
// Where a (cell, value) pair of an output row goes: slot `pos` of the row stream (`ent`, 8 bytes per
// pair) or of the CSR arrays of X^T (idx != nullptr: t_indices / t_values).
struct TOut {
  unsigned long long* ent;
  int32_t* idx;
  float* val;
};
__device__ __forceinline__ void t_store(const TOut& o, int64_t pos, unsigned long long e) {
  if (o.idx) {
    o.idx[pos] = (int32_t)(unsigned)e;
    o.val[pos] = __builtin_bit_cast(float, (unsigned)(e >> 32));
  } else {
    o.ent[pos] = e;
  }
}

constexpr int kF2Cols = 768;     // max columns per slab (the host picks C <= kF2Cols from nnz / d)
constexpr int kF2Cap = 10240;    // staged pairs: 80 KiB
constexpr int kF2Rows = 8;       // rows a wave has in flight

__device__ __forceinline__ int64_t readlane_i64(int64_t v, int l) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return ((int64_t)hi << 32) | (int64_t)(uint32_t)lo;
}

// PHASE 0: count pairs per (wave, column).  PHASE 1: place them (LDS when `staged`, else global).
// A wave takes its rows 64 at a time (lane l keeps the cursor of row l), eight rows per batch;
// the entries of the next batch are requested before the current batch is processed, so the LDS
// bucket chains of one batch hide the memory latency of the next.
struct F2Batch {
  int32_t ci[kF2Rows];
  float cv[kF2Rows];
};

// (cursors are 32-bit offsets from the row block's first entry: one v_readlane instead of two and
//  32-bit scalar arithmetic in the per-row code, which is instruction bound - an earlier variant
//  with ~30 more scalar instructions per row was 45 % slower)
template <int PHASE>
__device__ __forceinline__ void f2_load(F2Batch& b, int cur, int end, int first,
                                        const int32_t* __restrict__ indices_b,
                                        const float* __restrict__ values_b) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < kF2Rows; ++j) {
    const int src = (first + j) & 63;
    const int p = __builtin_amdgcn_readlane(cur, src) + lane;
    const bool in = (first + j < 64) && (p < __builtin_amdgcn_readlane(end, src));  // rows past the range: cur = end = 0
    b.ci[j] = in ? indices_b[p] : 0x7fffffff;
    if (PHASE == 1) b.cv[j] = in ? values_b[p] : 0.f;
  }
}

template <int PHASE>
__device__ __forceinline__ void f2_process(const F2Batch& b, int& cur, int end, int first,
                                           int64_t row0, int32_t cbase, int32_t cend,
                                           const int32_t* __restrict__ indices_b,
                                           const float* __restrict__ values_b, uint32_t* wbucket,
                                           const uint32_t* lpos, const int64_t* gdst,
                                           unsigned long long* stage, bool staged,
                                           const TOut& ent) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < kF2Rows; ++j) {
    if (first + j >= 64) break;  // uniform
    int c0 = __builtin_amdgcn_readlane(cur, first + j);
    const int e0 = __builtin_amdgcn_readlane(end, first + j);
    int32_t c = b.ci[j];
    float v = (PHASE == 1) ? b.cv[j] : 0.f;
    while (true) {
      const bool valid = c < cend;  // sorted rows: a prefix of the 64 loaded entries
      const int n = __popcll(__ballot(valid));
      if (valid) {
        const int cl = c - cbase;
        if (PHASE == 0) {
          wbucket[cl] += 1u;  // columns inside one row are distinct: no two lanes share a counter
        } else {
          const uint32_t k = wbucket[cl];
          wbucket[cl] = k + 1u;
          const unsigned long long e = (unsigned long long)(unsigned)(row0 + first + j) |
                                       ((unsigned long long)__builtin_bit_cast(unsigned, v) << 32);
          if (staged) stage[lpos[cl] + k] = e;
          else t_store(ent, gdst[cl] + k, e);
        }
      }
      c0 += n;
      if (n < 64) break;  // wave-uniform
      const int p = c0 + lane;  // a row with more than 64 entries in this slab
      const bool in = p < e0;
      c = in ? indices_b[p] : 0x7fffffff;
      if (PHASE == 1) v = in ? values_b[p] : 0.f;
    }
    if (PHASE == 1 && lane == first + j) cur = c0;
  }
}

template <int PHASE>
__device__ __forceinline__ void f2_walk(int64_t wrow0, int64_t wrow1, int32_t cbase, int32_t cend,
                                        int64_t wg_base, const int64_t* __restrict__ indptr,
                                        const int32_t* __restrict__ indices,
                                        const float* __restrict__ values, int64_t* __restrict__ curs,
                                        uint32_t* wbucket, const uint32_t* lpos, const int64_t* gdst,
                                        unsigned long long* stage, bool staged,
                                        const TOut& ent) {
  const int lane = threadIdx.x & 63;
  const int32_t* __restrict__ indices_b = indices + wg_base;
  const float* __restrict__ values_b = values + wg_base;
  for (int64_t sb = wrow0; sb < wrow1; sb += 64) {  // wave-uniform
    const int nr = (wrow1 - sb) < 64 ? (int)(wrow1 - sb) : 64;
    int cur = 0, end = 0;
    if (lane < nr) {
      cur = (int)(curs[sb + lane] - wg_base);
      end = (int)(indptr[sb + lane + 1] - wg_base);
    }
    F2Batch ba, bb;
    f2_load<PHASE>(ba, cur, end, 0, indices_b, values_b);
    for (int first = 0; first < nr; first += 2 * kF2Rows) {
      // (the cursors of the rows of a batch are final before its loads are issued: rows are
      //  independent, only `cur` of the rows being processed changes)
      if (first + kF2Rows < nr) f2_load<PHASE>(bb, cur, end, first + kF2Rows, indices_b, values_b);
      f2_process<PHASE>(ba, cur, end, first, sb, cbase, cend, indices_b, values_b, wbucket, lpos, gdst, stage,
                        staged, ent);
      if (first + kF2Rows < nr) {
        if (first + 2 * kF2Rows < nr) f2_load<PHASE>(ba, cur, end, first + 2 * kF2Rows, indices_b, values_b);
        f2_process<PHASE>(bb, cur, end, first + kF2Rows, sb, cbase, cend, indices_b, values_b, wbucket, lpos,
                          gdst, stage, staged, ent);
      }
    }
    if (PHASE == 1 && lane < nr) curs[sb + lane] = wg_base + (int64_t)cur;
  }
}

__global__ __launch_bounds__(kTThreads) void k_t_fill2(int64_t n_rows, int64_t n_cols, int C, int abl,
                                                       const int64_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ indices,
                                                       const float* __restrict__ values,
                                                       int64_t* __restrict__ curs,
                                                       const int64_t* __restrict__ cptr,
                                                       const int32_t* __restrict__ inv,
                                                       const uint32_t* __restrict__ base,
                                                       const int64_t* __restrict__ coltot,
                                                       TOut ent) {
  __shared__ unsigned long long stage[kF2Cap];     // 80 KiB
  __shared__ uint32_t bucket[kTWaves][kF2Cols];    // 48 KiB: per (wave, column) count, then cursor
  __shared__ uint32_t lcount[kF2Cols], lpos[kF2Cols];
  __shared__ int64_t gdst[kF2Cols];                // first pair this row block writes in the column's output row
  __shared__ uint32_t wsum[kTWaves];
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) t_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  __syncthreads();
  const int64_t r0 = s_r[0], r1 = s_r[1];
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t rw = (r1 - r0 + kTWaves - 1) / kTWaves;  // rows per wave
  const int64_t wrow0 = (r0 + wave * rw) < r1 ? (r0 + wave * rw) : r1;
  const int64_t wrow1 = (wrow0 + rw) < r1 ? (wrow0 + rw) : r1;
  const int64_t wg_base = uniform64(indptr[r0 < n_rows ? r0 : n_rows]);  // entries of this row block start here
  const uint32_t* base_g = base + (int64_t)g * n_cols;
  const uint32_t* base_n = (g + 1 < G) ? base + (int64_t)(g + 1) * n_cols : nullptr;

  for (int64_t cb = 0; cb < n_cols; cb += C) {
    const int32_t cbase = (int32_t)cb;
    const int32_t cend = (int32_t)((cb + C) < n_cols ? (cb + C) : n_cols);
    // tile counts per column, their exclusive scan, global run starts; clear the buckets
    for (int t = threadIdx.x; t < kTWaves * kF2Cols; t += kTThreads) (&bucket[0][0])[t] = 0u;
    uint32_t mine = 0;
    if (threadIdx.x < kF2Cols) {
      const int64_t c = (int64_t)cbase + threadIdx.x;
      if (c < cend) {
        const uint32_t b0 = base_g[c];
        const uint32_t b1 = base_n ? base_n[c] : (uint32_t)coltot[c];
        mine = b1 - b0;
        gdst[threadIdx.x] = cptr[inv ? (int64_t)inv[c] : c] + (int64_t)b0;
      }
      lcount[threadIdx.x] = mine;
    }
    // block exclusive scan of lcount (first kF2Cols threads = 12 waves)
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wpre = 0, total = 0;
    for (int w = 0; w < kTWaves; ++w) {
      const uint32_t t = wsum[w];
      if (w < wave) wpre += t;
      total += t;
    }
    if (threadIdx.x < kF2Cols) lpos[threadIdx.x] = wpre + incl - mine;
    const bool staged = total <= (uint32_t)kF2Cap;
    __syncthreads();
    if (total == 0) continue;  // uniform: nothing of this row block falls into the slab

    if (!(abl & 1))
    f2_walk<0>(wrow0, wrow1, cbase, cend, wg_base, indptr, indices, values, curs, bucket[wave], lpos, gdst,
               stage, staged, ent);
    __syncthreads();
    // per column: exclusive prefix of the wave counts = first slot of every wave inside the run
    if (threadIdx.x < kF2Cols) {
      uint32_t run = 0;
      for (int w = 0; w < kTWaves; ++w) {
        const uint32_t t = bucket[w][threadIdx.x];
        bucket[w][threadIdx.x] = run;
        run += t;
      }
    }
    __syncthreads();
    if (!(abl & 2))
    f2_walk<1>(wrow0, wrow1, cbase, cend, wg_base, indptr, indices, values, curs, bucket[wave], lpos, gdst,
               stage, staged, ent);
    __syncthreads();
    if (staged && !(abl & 4)) {
      // write-out: one 16-lane group per column, consecutive lanes = consecutive pairs of the run
      const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
      for (int cl = grp; cl < cend - cbase; cl += kTThreads / 16) {
        const uint32_t L = lcount[cl], src = lpos[cl];
        const int64_t dst = gdst[cl];
        for (uint32_t i = sub; i < L; i += 16) t_store(ent, dst + i, stage[src + i]);
      }
    }
    __syncthreads();
  }
}

// ---- fill, version 3: the count walk rides on the previous tile's place walk ---------------------
// The row loops of version 2 are instruction bound and run twice per tile (count, place).  The 64
// entries a wave loads from a row's cursor reach well past the tile (a row has ~20 entries per
// tile), i.e. the place walk of tile t has already seen tile t+1's entries of that row: it counts
// them into a second, 16-bit bucket array, and tile t+1 starts with its counts done.  Only the
// first tile (and a tile that follows an empty one) needs an explicit count walk.
// cycle accounting of the fill's phases (tune tpack_dbg = 1; read with mu_csr_tpack_phase_cycles):
// header + scan, count walk, per-column prefix over waves, place walk, write-out - per workgroup,
// measured by wave 0 between the barriers, summed over workgroups and tiles
__device__ unsigned long long g_t_phase[8];

// Two builds of the third-generation fill.  NARROW: 16-bit cursors packed in pairs (20 KiB) leave
// room for 112 KiB of staging - more rows per block, fewer rounds of blocks: 86.5 instead of 88.9 ms
// at 1e6 x 200k.  The packing costs a shift and a mask per visit, so an input that is one round of
// blocks either way (125k x 200k: 12.9 against 12.3 ms) takes the 32-bit build with 80 KiB of staging.
constexpr int kF3Cols = 640;
constexpr int kF3CapNarrow = 14336, kF3CapWide = 10240;  // staged pairs
constexpr int64_t kF3MaxRows = 16ll * 65535;  // a wave's count of one column fits 16 bits

// 16-bit count i of a wave's count row += 1, as a 32-bit LDS atomic on the word that holds it
// (neighbouring columns of one row share a word: the atomic serialises them)
__device__ __forceinline__ void cnt_add(uint16_t* wcnt, int i) {
  __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(wcnt) + (i >> 1), 1u << ((i & 1) * 16), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}

// 16-bit cursor i of a wave's cursor row: fetch and add 1, same packing (the neighbour's half of the
// returned word is whatever it was at the time - only this lane's half is looked at)
__device__ __forceinline__ uint32_t cur_fetch_inc(uint16_t* wcur, int i) {
  const int sh = (i & 1) * 16;
  const uint32_t old = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(wcur) + (i >> 1), 1u << sh,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return (old >> sh) & 0xffffu;
}

// ---- the batch loads of the third-generation walk, from asm (r04) -----------------------------------------------
// f2_load's predicated loads (`in ? indices_b[p] : 0x7fffffff`) are conditional blocks to the compiler, and a wait
// behind conditional loads cannot be a counted one: k_t_fill3 carried 94 `s_waitcnt vmcnt(0)` and not one counted wait -
// one in front of every batch's loads and one right behind them, i.e. the batch "requested before the current one is
// processed" was awaited before the current one was processed, and a wave walked its rows one memory latency per
// batch of eight.  Here every load is unconditional (a lane past its row's end reads a clamped position and is
// masked when the batch is taken), issued from asm, and the batch is taken behind a counted wait that leaves the
// next batch's loads in flight.
template <int PHASE>
__device__ __forceinline__ void f3_load_asm(F2Batch& b, int cur, int first, int last,
                                            const int32_t* indices_b, const float* values_b) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < kF2Rows; ++j) {
    const int src = (first + j) & 63;
    int p = __builtin_amdgcn_readlane(cur, src) + lane;
    p = p < last ? p : last;
    const unsigned off = (unsigned)p * 4u;
    if (PHASE == 1)
      asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %4"
                   : "=&v"(b.ci[j]), "=&v"(b.cv[j])
                   : "v"(off), "s"(indices_b), "s"(values_b)
                   : "memory");
    else
      asm volatile("global_load_dword %0, %1, %2" : "=&v"(b.ci[j]) : "v"(off), "s"(indices_b) : "memory");
  }
}
// The reloads inside a take (a row with more than 64 entries in the tile; the look-ahead past a full window) are
// rare, but a compiler-visible load in the row loop puts the compiler's own `s_waitcnt vmcnt(0)` at the loop's head -
// on EVERY row, where it drains the prefetched batch.  They load and wait inside one asm instead.
template <bool WITH_VALUE>
__device__ __forceinline__ void f3_reload_asm(int32_t& c, float& v, int p, int last, const int32_t* indices_b,
                                              const float* values_b) {
  p = p < last ? p : last;
  const unsigned off = (unsigned)p * 4u;
  if (WITH_VALUE)
    asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %4\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(c), "=&v"(v)
                 : "v"(off), "s"(indices_b), "s"(values_b)
                 : "memory");
  else
    asm volatile("global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=&v"(c) : "v"(off), "s"(indices_b) : "memory");
}

template <int PHASE, int N>
__device__ __forceinline__ void f3_wait_asm(F2Batch& b) {
  static_assert(kF2Rows == 8, "the operand list names 8 rows");
  if (PHASE == 1)
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(b.ci[0]), "+v"(b.ci[1]), "+v"(b.ci[2]), "+v"(b.ci[3]), "+v"(b.ci[4]), "+v"(b.ci[5]), "+v"(b.ci[6]),
                   "+v"(b.ci[7]), "+v"(b.cv[0]), "+v"(b.cv[1]), "+v"(b.cv[2]), "+v"(b.cv[3]), "+v"(b.cv[4]), "+v"(b.cv[5]),
                   "+v"(b.cv[6]), "+v"(b.cv[7])
                 : "n"(N)
                 : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(b.ci[0]), "+v"(b.ci[1]), "+v"(b.ci[2]), "+v"(b.ci[3]), "+v"(b.ci[4]), "+v"(b.ci[5]), "+v"(b.ci[6]),
                   "+v"(b.ci[7])
                 : "n"(N)
                 : "memory");
}

// STAGED: 1 / 0 = the tile is / is not staged in LDS, known at compile time (the asm walk is instantiated for both: the
// per-row code loses its branches on flags that do not change during a walk); -1 = the run-time flag `staged_rt`
template <int PHASE, bool NARROW, bool ASM = false, int STAGED = -1>
__device__ __forceinline__ void f3_process(F2Batch& b, int& cur, int end, int first,
                                           int64_t row0, int32_t cbase, int32_t cend, int32_t cend2,
                                           const int32_t* __restrict__ indices_b,
                                           const float* __restrict__ values_b, uint16_t* wcur,
                                           uint16_t* wcnt, uint32_t* wcur32, const int64_t* gdst,
                                           unsigned long long* stage, bool staged_rt,
                                           const TOut& ent, int last = 0) {
  const int lane = threadIdx.x & 63;
  const bool staged = STAGED < 0 ? staged_rt : (STAGED == 1);
  if constexpr (ASM) f3_wait_asm<PHASE, (PHASE == 1 ? 2 : 1) * kF2Rows>(b);  // (the next batch's loads stay in flight)
#pragma unroll
  for (int j = 0; j < kF2Rows; ++j) {
    if (first + j >= 64) break;  // uniform
    int c0 = __builtin_amdgcn_readlane(cur, first + j);
    const int e0 = __builtin_amdgcn_readlane(end, first + j);
    int ws = c0;  // where the loaded window starts
    int32_t c = b.ci[j];
    float v = (PHASE == 1) ? b.cv[j] : 0.f;
    if constexpr (ASM) {  // (the unconditional load read a clamped position for the lanes past the row's end)
      const bool in = c0 + lane < e0;
      c = in ? c : 0x7fffffff;
      if (PHASE == 1) v = in ? v : 0.f;
    }
    while (true) {
      const bool valid = c < cend;  // sorted rows: a prefix of the 64 loaded entries
      const int n = __popcll(__ballot(valid));
      if (PHASE == 0) {
        if (valid) cnt_add(wcnt, c - cbase);
      } else {
        // Both bucket updates of the visit are LDS atomics (columns inside one row are distinct, so
        // they never collide inside a wave - the atomic is there for the single round trip): the
        // cursor of this tile for the lanes inside it (ds_add_rtn_u32, the one result the visit waits
        // for) and the count of the next tile for the lanes just past it (ds_add_u32, not waited for).
        // (r01/r02: read-modify-write pairs through a dummy slot for idle lanes - four LDS
        //  instructions and two results to wait for.)
        const bool nxt = !valid && c < cend2;
        uint32_t k = 0;   // absolute slot in the staging buffer (run-relative when not staged)
        if constexpr (NARROW && ASM && STAGED == 1) {
          // cursors and counts of a wave are one array (k_t_fill3): column c of this tile or the next is entry c - cbase
          if (valid || nxt) k = cur_fetch_inc(wcur, c - cbase);
          if (valid) {
            const unsigned long long e = (unsigned long long)(unsigned)(row0 + first + j) |
                                         ((unsigned long long)__builtin_bit_cast(unsigned, v) << 32);
            stage[k] = e;
          }
        } else {
        if (valid) {
          // NARROW, staged: 16-bit absolute staging slots; a tile that does not fit the staging
          // buffer keeps 32-bit run-relative cursors in the (then unused) staging memory.  Wide
          // build: one 32-bit cursor array for both cases.
          k = (NARROW && staged)
                  ? cur_fetch_inc(wcur, c - cbase)
                  : __hip_atomic_fetch_add(&wcur32[c - cbase], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (nxt) cnt_add(wcnt, c - cend);
        if (valid) {
          const unsigned long long e = (unsigned long long)(unsigned)(row0 + first + j) |
                                       ((unsigned long long)__builtin_bit_cast(unsigned, v) << 32);
          if (staged) stage[k] = e;
          else t_store(ent, gdst[c - cbase] + k, e);
        }
        }
      }
      c0 += n;
      if (n == 64) {  // wave-uniform: a row with more than 64 entries in this tile
        ws = c0;
        const int p = c0 + lane;
        const bool in = p < e0;
        if constexpr (ASM) {
          f3_reload_asm<PHASE == 1>(c, v, p, last, indices_b, values_b);
          c = in ? c : 0x7fffffff;
          if (PHASE == 1) v = in ? v : 0.f;
        } else {
          c = in ? indices_b[p] : 0x7fffffff;
          if (PHASE == 1) v = in ? values_b[p] : 0.f;
        }
        continue;
      }
      if (PHASE == 1) {
        // the window ends inside the next tile: keep counting that tile (rare: > 64 entries in two tiles)
        while (__popcll(__ballot(c < cend2)) == 64) {
          ws += 64;
          const int p = ws + lane;
          if constexpr (ASM) {
            float unused;
            f3_reload_asm<false>(c, unused, p, last, indices_b, values_b);
            c = (p < e0) ? c : 0x7fffffff;
          } else {
            c = (p < e0) ? indices_b[p] : 0x7fffffff;
          }
          if (c < cend2) cnt_add(wcnt, c - cend);
        }
      }
      break;
    }
    if (PHASE == 1 && lane == first + j) cur = c0;
  }
}

template <int PHASE, bool NARROW, bool ASM = false, int STAGED = -1>
__device__ __forceinline__ void f3_walk(int64_t wrow0, int64_t wrow1, int32_t cbase, int32_t cend,
                                        int32_t cend2, int64_t wg_base,
                                        const int64_t* __restrict__ indptr,
                                        const int32_t* __restrict__ indices,
                                        const float* __restrict__ values, int64_t* __restrict__ curs,
                                        uint16_t* wcur, uint16_t* wcnt, uint32_t* wcur32,
                                        const int64_t* gdst, unsigned long long* stage, bool staged,
                                        const TOut& ent, int last = 0) {
  const int lane = threadIdx.x & 63;
  const int32_t* __restrict__ indices_b = indices + wg_base;
  const float* __restrict__ values_b = values + wg_base;
  for (int64_t sb = wrow0; sb < wrow1; sb += 64) {  // wave-uniform
    const int nr = (wrow1 - sb) < 64 ? (int)(wrow1 - sb) : 64;
    int cur = 0, end = 0;
    if (lane < nr) {
      cur = (int)(curs[sb + lane] - wg_base);
      end = (int)(indptr[sb + lane + 1] - wg_base);
    }
    F2Batch ba, bb;
    if constexpr (ASM) {
      asm volatile("" ::"v"(cur), "v"(end));  // (a use: the compiler's wait for these two loads sits here, not inside a take)
      // Every load is issued whether or not its batch exists (a batch past the strip reads the positions its lanes'
      // cursors name - valid addresses, never taken): the number of loads behind a batch is then the same on every
      // path, ONE counted wait serves all takes, and there is no conditional definition of a batch register for the
      // compiler to merge with a copy (a copy of a register whose load is still in flight reads garbage: the first
      // version, with `if (more) load`, had sixteen of them in front of its wait).  What is in flight at the end is
      // drained with the registers as operands.
      f3_load_asm<PHASE>(ba, cur, 0, last, indices_b, values_b);
      for (int first = 0; first < nr; first += 2 * kF2Rows) {
        f3_load_asm<PHASE>(bb, cur, first + kF2Rows, last, indices_b, values_b);
        f3_process<PHASE, NARROW, true, STAGED>(ba, cur, end, first, sb, cbase, cend, cend2, indices_b, values_b, wcur, wcnt,
                                        wcur32, gdst, stage, staged, ent, last);
        if (first + kF2Rows >= nr) break;
        f3_load_asm<PHASE>(ba, cur, first + 2 * kF2Rows, last, indices_b, values_b);
        f3_process<PHASE, NARROW, true, STAGED>(bb, cur, end, first + kF2Rows, sb, cbase, cend, cend2, indices_b, values_b,
                                        wcur, wcnt, wcur32, gdst, stage, staged, ent, last);
      }
      f3_wait_asm<PHASE, 0>(ba);
      f3_wait_asm<PHASE, 0>(bb);
      if (PHASE == 1 && lane < nr) curs[sb + lane] = wg_base + (int64_t)cur;
      continue;
    }
    f2_load<PHASE>(ba, cur, end, 0, indices_b, values_b);
    for (int first = 0; first < nr; first += 2 * kF2Rows) {
      if (first + kF2Rows < nr) f2_load<PHASE>(bb, cur, end, first + kF2Rows, indices_b, values_b);
      f3_process<PHASE, NARROW>(ba, cur, end, first, sb, cbase, cend, cend2, indices_b, values_b, wcur, wcnt, wcur32,
                        gdst, stage, staged, ent);
      if (first + kF2Rows < nr) {
        if (first + 2 * kF2Rows < nr) f2_load<PHASE>(ba, cur, end, first + 2 * kF2Rows, indices_b, values_b);
        f3_process<PHASE, NARROW>(bb, cur, end, first + kF2Rows, sb, cbase, cend, cend2, indices_b, values_b, wcur,
                          wcnt, wcur32, gdst, stage, staged, ent);
      }
    }
    if (PHASE == 1 && lane < nr) curs[sb + lane] = wg_base + (int64_t)cur;
  }
}

template <bool NARROW, bool ASM = false>
__global__ __launch_bounds__(kTThreads) void k_t_fill3(int64_t n_rows, int64_t n_cols, int C, int flags,
                                                       const int64_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ indices,
                                                       const float* __restrict__ values,
                                                       int64_t* __restrict__ curs,
                                                       const int64_t* __restrict__ cptr,
                                                       const int32_t* __restrict__ inv,
                                                       const uint32_t* __restrict__ base,
                                                       const int64_t* __restrict__ coltot,
                                                       TOut ent) {
  constexpr int kF3Cap = NARROW ? kF3CapNarrow : kF3CapWide;
  typedef typename std::conditional<NARROW, uint16_t, uint32_t>::type CurT;
  __shared__ unsigned long long stage[kF3Cap];     // 112 / 80 KiB
  // per (wave, column): the cursor of this tile and the count of the tile in the making.  Narrow build: ONE 16-bit array
  // per wave, cursors at [0, C), counts at [C, 2 C) - column c of either tile is entry c - cbase, so a row visit updates
  // both with one packed atomic (r04; two index computations and two LDS atomics before).  Wide build: 32-bit cursors
  // and 16-bit counts apart.
  __shared__ CurT wcur_all[kTWaves][NARROW ? 2 * kF3Cols : kF3Cols];  // 40 KiB
  __shared__ uint16_t wcnt_wide[NARROW ? 1 : kTWaves][NARROW ? 1 : kF3Cols];  // (wide build: 20 KiB)
  auto wcnt_of = [&](int w) -> uint16_t* {
    if constexpr (NARROW) return reinterpret_cast<uint16_t*>(&wcur_all[w][0]) + C;
    else return &wcnt_wide[w][0];
  };
  __shared__ uint16_t lcount[kF3Cols], lpos[kF3Cols];  // staged tiles only (<= kF3Cap pairs)
  __shared__ int64_t gdst[kF3Cols];
  static_assert(sizeof(uint32_t) * kTWaves * kF3Cols <= sizeof(unsigned long long) * kF3Cap, "direct-mode cursors live in the staging buffer");
  // 32-bit cursors: the wide build's own array; the narrow build's direct-mode cursors (staging memory)
  uint32_t(*wcur32_all)[kF3Cols] = NARROW ? reinterpret_cast<uint32_t(*)[kF3Cols]>(stage)
                                          : reinterpret_cast<uint32_t(*)[kF3Cols]>(&wcur_all[0][0]);
  static_assert(NARROW || sizeof(wcur_all) == sizeof(uint32_t) * kTWaves * kF3Cols, "wide build: 32-bit cursors in wcur_all");
  __shared__ uint32_t wsum[kTWaves];
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) t_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  if constexpr (NARROW) {
    for (int t = threadIdx.x; t < kTWaves * 2 * kF3Cols; t += kTThreads) (&wcur_all[0][0])[t] = (CurT)0;
  } else {
    for (int t = threadIdx.x; t < kTWaves * kF3Cols; t += kTThreads) (&wcnt_wide[0][0])[t] = (uint16_t)0;
  }
  __syncthreads();
  const int64_t r0 = s_r[0], r1 = s_r[1];
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t wg_base = uniform64(indptr[r0 < n_rows ? r0 : n_rows]);
  // A wave owns a contiguous range of the block's rows - any contiguous split keeps the fill stable.  r04: ranges of
  // alike ENTRY counts instead of alike row counts (the waves meet at four barriers per tile; flags bit 1: equal row
  // counts, for comparison - and whenever a range could exceed the 65 535 rows a 16-bit count stands for)
  int64_t wrow0, wrow1;
  if ((flags & 2) || r1 - r0 > 65535) {
    const int64_t rw = (r1 - r0 + kTWaves - 1) / kTWaves;  // rows per wave
    wrow0 = (r0 + wave * rw) < r1 ? (r0 + wave * rw) : r1;
    wrow1 = (wrow0 + rw) < r1 ? (wrow0 + rw) : r1;
  } else {
    const int64_t e0 = wg_base, e1 = uniform64(indptr[r1 < n_rows ? r1 : n_rows]);
    auto cut = [&](int w) -> int64_t {
      if (w <= 0) return r0;
      if (w >= kTWaves) return r1;
      const int64_t key = e0 + ((e1 - e0) / kTWaves) * w + (((e1 - e0) % kTWaves) * w) / kTWaves;
      const int64_t r = lower_bound_i64(indptr, r0, r1, key);
      return r > r1 ? r1 : r;
    };
    wrow0 = uniform64(cut(wave));
    wrow1 = uniform64(cut(wave + 1));
  }
  // (ASM: the last position an unconditional load may read, relative to the block's first entry, as a 30-bit offset)
  int64_t last64 = uniform64(indptr[n_rows]) - 1 - wg_base;
  last64 = last64 < 0 ? 0 : (last64 > ((1ll << 30) - 1) ? ((1ll << 30) - 1) : last64);
  const int last = (int)last64;
  const uint32_t* base_g = base + (int64_t)g * n_cols;
  const uint32_t* base_n = (g + 1 < G) ? base + (int64_t)(g + 1) * n_cols : nullptr;
  const bool dbg = flags & 1;  // (phase accounting)
  bool have = false;  // wcnt_all holds the counts of the tile about to be processed (uniform)
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
  auto mark = [&](int i) {
    if (dbg) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      ph[i] += t - t_prev;
      t_prev = t;
    }
  };

  for (int64_t cb = 0; cb < n_cols; cb += C) {
    const int32_t cbase = (int32_t)cb;
    const int32_t cend = (int32_t)((cb + C) < n_cols ? (cb + C) : n_cols);
    const int32_t cend2 = (int32_t)((cb + 2 * (int64_t)C) < n_cols ? (cb + 2 * (int64_t)C) : n_cols);
    uint32_t mine = 0;
    if (threadIdx.x < kF3Cols) {
      const int64_t c = (int64_t)cbase + threadIdx.x;
      if (c < cend) {
        const uint32_t b0 = base_g[c];
        const uint32_t b1 = base_n ? base_n[c] : (uint32_t)coltot[c];
        mine = b1 - b0;
        gdst[threadIdx.x] = cptr[inv ? (int64_t)inv[c] : c] + (int64_t)b0;
      }
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wpre = 0, total = 0;
    for (int w = 0; w < kTWaves; ++w) {
      const uint32_t t = wsum[w];
      if (w < wave) wpre += t;
      total += t;
    }
    const bool staged = total <= (uint32_t)kF3Cap;
    if (threadIdx.x < kF3Cols && staged) {  // (16 bits are enough for a staged tile)
      lcount[threadIdx.x] = (uint16_t)mine;
      lpos[threadIdx.x] = (uint16_t)(wpre + incl - mine);
    }
    const uint32_t my_lpos = wpre + incl - mine;
    __syncthreads();
    if (total == 0) {  // uniform: nothing here, so nobody counted the next tile either
      have = false;    // (the counts of an empty tile are zeros: wcnt_all stays clear)
      continue;
    }
    mark(0);
    if (!have)
      f3_walk<0, NARROW, ASM>(wrow0, wrow1, cbase, cend, cend2, wg_base, indptr, indices, values, curs,
                              reinterpret_cast<uint16_t*>(wcur_all[wave]), wcnt_of(wave), wcur32_all[wave], gdst, stage,
                              staged, ent, last);  // (the count walk does not look at `staged`)
    __syncthreads();
    mark(1);
    // per column: exclusive prefix of the wave counts = first slot of every wave inside the run;
    // the counts are consumed (zeroed) for the next tile
    if (threadIdx.x < (NARROW ? C : kF3Cols)) {  // (narrow build: entries C .. of the cursor half are the counts)
      // a staged tile's cursors are absolute staging slots (no lpos lookup in the walk), a direct
      // one's are relative to the column's run in the output
      uint32_t run = staged ? my_lpos : 0u;
      for (int w = 0; w < kTWaves; ++w) {
        uint16_t* wc = wcnt_of(w);
        const uint32_t t = wc[threadIdx.x];
        wc[threadIdx.x] = (uint16_t)0;
        if (NARROW && staged) wcur_all[w][threadIdx.x] = (CurT)run;
        else wcur32_all[w][threadIdx.x] = run;
        run += t;
      }
    }
    __syncthreads();
    mark(2);
    if (ASM && staged)
      f3_walk<1, NARROW, ASM, ASM ? 1 : -1>(wrow0, wrow1, cbase, cend, cend2, wg_base, indptr, indices, values, curs,
                                           reinterpret_cast<uint16_t*>(wcur_all[wave]), wcnt_of(wave), wcur32_all[wave],
                                           gdst, stage, staged, ent, last);
    else if (ASM)
      f3_walk<1, NARROW, ASM, ASM ? 0 : -1>(wrow0, wrow1, cbase, cend, cend2, wg_base, indptr, indices, values, curs,
                                           reinterpret_cast<uint16_t*>(wcur_all[wave]), wcnt_of(wave), wcur32_all[wave],
                                           gdst, stage, staged, ent, last);
    else
      f3_walk<1, NARROW, ASM>(wrow0, wrow1, cbase, cend, cend2, wg_base, indptr, indices, values, curs,
                              reinterpret_cast<uint16_t*>(wcur_all[wave]), wcnt_of(wave), wcur32_all[wave], gdst, stage,
                              staged, ent, last);
    have = true;
    mark(3);
    __syncthreads();
    mark(4);
    // (r04: the write-out as flat passes - the run headers of all of a group's columns first, then the first 16 pairs
    //  of every run, then the next 16: a few LDS round trips per tile instead of four per column - measured SLOWER,
    //  73.3 -> 77.7 ms: the two halves of a 176-byte run then reach the memory side ten stores apart instead of back
    //  to back.  The write-out is bound by its partial-line stores, not by the LDS chain.)
    if (staged) {
      const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
      for (int cl = grp; cl < cend - cbase; cl += kTThreads / 16) {
        const uint32_t L = lcount[cl], src = lpos[cl];
        const int64_t dst = gdst[cl];
        for (uint32_t i = sub; i < L; i += 16) t_store(ent, dst + i, stage[src + i]);
      }
    }
    __syncthreads();
    mark(5);
  }
  if (dbg && threadIdx.x == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_t_phase[i], ph[i]);
}

struct TWork {
  int64_t* sp;
  uint32_t* cnt;
  int64_t* coltot;
  int64_t* curs;
};
inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
inline TWork carve(void* work, int64_t n_rows, int64_t n_cols, int64_t nnz) {
  const int64_t S = t_slabs(n_cols);
  char* w = (char*)work;
  TWork t;
  t.sp = (int64_t*)w;
  w += al((size_t)(n_rows * (S + 1)) * sizeof(int64_t));
  t.cnt = (uint32_t*)w;
  w += al((size_t)t_grid(nnz, n_cols) * (size_t)n_cols * sizeof(uint32_t));
  t.coltot = (int64_t*)w;
  w += al((size_t)n_cols * sizeof(int64_t));
  t.curs = (int64_t*)w;
  return t;
}

}  // namespace

extern "C" {

size_t mu_csr_tpack_worksize(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  const int64_t S = t_slabs(n_cols);
  return al((size_t)(n_rows * (S + 1)) * sizeof(int64_t)) +
         al((size_t)t_grid(nnz, n_cols) * (size_t)n_cols * sizeof(uint32_t)) +
         al((size_t)n_cols * sizeof(int64_t)) + al((size_t)(n_rows + 1) * sizeof(int64_t)) + 256;
}

int mu_csr_tpack_count(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr,
                       const int32_t* d_indices, int64_t* d_col_nnz, void* d_work,
                       size_t work_bytes, const int64_t* d_slab_ptr, void* stream) {
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative size");
  MU_REQUIRE(n_rows < ((int64_t)1 << 31), "row ids must fit int32");
  if (n_cols == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_col_nnz && d_work, "null pointer");
  MU_REQUIRE(work_bytes >= mu_csr_tpack_worksize(n_rows, n_cols, nnz), "work buffer too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t S = t_slabs(n_cols);
  const int G = t_grid(nnz, n_cols);
  const TWork w = carve(d_work, n_rows, n_cols, nnz);
  MU_CHECK_HIP(hipMemsetAsync(w.cnt, 0, (size_t)G * (size_t)n_cols * sizeof(uint32_t), st));
  if (n_rows > 0) {
    const int64_t total = n_rows * (S + 1);
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)mu_num_cus() * 32;
    if (blocks > cap) blocks = cap;
    if (!d_slab_ptr) {  // (the TF-IDF sweeps of the same index arrays searched the same slabs: sweep.hpp kSlab)
      hipLaunchKernelGGL(k_t_slab_ptr, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, S, d_indptr,
                         d_indices, w.sp);
      MU_CHECK_LAUNCH();
    }
    // pipelined sweep with 32 768-column bins (r04); tune "tcount_pipe" = 1 or a row block of 2^29 entries and more
    // (32-bit byte offsets): the sweep of before
    if (mu_tune_get("tcount_pipe") != 1 && nnz / G < (1ll << 29) - (1ll << 24))
      hipLaunchKernelGGL((k_t_count_pipe<4>), dim3(G), dim3(kTThreads), 0, st, n_rows, n_cols, S, d_indptr, d_indices,
                         d_slab_ptr ? d_slab_ptr : w.sp, w.cnt);
    else
      hipLaunchKernelGGL(k_t_count, dim3(G), dim3(kTThreads), 0, st, n_rows, n_cols, S, d_indptr,
                       d_indices, d_slab_ptr ? d_slab_ptr : w.sp, w.cnt);
    MU_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_t_base, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, n_cols, G,
                     w.cnt, w.coltot, d_col_nnz);
  MU_CHECK_LAUNCH();
  // the fill's per-row cursors start at the row starts.  Done here, not in the fill: a caller that
  // overlaps the fill with other work (the pack of X on a second stream) would have this small copy
  // wait for a free CU behind that work, and the fill behind the copy.
  if (n_rows > 0)
    MU_CHECK_HIP(hipMemcpyAsync(w.curs, d_indptr, sizeof(int64_t) * (size_t)n_rows, hipMemcpyDeviceToDevice, st));
  return MU_OK;
}

static int tpack_fill_impl(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr,
                           const int32_t* d_indices, const float* d_values, const int64_t* d_cptr,
                           const int32_t* d_inv, TOut out, void* d_work, hipStream_t st) {
  const int G = t_grid(nnz, n_cols);
  const TWork w = carve(d_work, n_rows, n_cols, nnz);
  if (n_rows > 0) {
    // slab width: the expected tile (nnz / G rows x C columns) fills ~93 % of the staging buffer
    // (measured best on the bench matrix: fewer, fuller tiles; a tile that overflows takes the
    //  direct-store path)
    const double per_col = (double)nnz / (double)G / (double)n_cols;  // pairs of a tile per column
    const bool v3 = n_rows <= kF3MaxRows && !mu_tune_get("tpack_v2");
    // more than one round of row blocks (or one round of blocks too big for 512-column tiles in the
    // small staging buffer): the bigger staging buffer pays
    bool narrow = G > mu_num_cus() || per_col > 0.93 * kF3CapWide / 512;
    if (mu_tune_get("tpack_narrow") > 0) narrow = mu_tune_get("tpack_narrow") == 1;  // tests: 1 narrow, 2 wide
    int64_t C = per_col > 0 ? (int64_t)(0.93 * (v3 ? (narrow ? kF3CapNarrow : kF3CapWide) : kF2Cap) / per_col) : kF2Cols;
    C = (C / 32) * 32;
    if (C < 32) C = 32;
    if (C > (v3 ? kF3Cols : kF2Cols)) C = v3 ? kF3Cols : kF2Cols;
    if (mu_tune_get("tpack_c") > 0) C = mu_tune_get("tpack_c");
    if (v3) {
      if (C > kF3Cols) C = kF3Cols;
      const bool asm_loads = mu_tune_get("tpack_asm") != 1;  // (tune tpack_asm = 1: the compiler's loads, for comparison)
#define MU_FILL3(NARROW_, ASM_)                                                                                    \
  hipLaunchKernelGGL((k_t_fill3<NARROW_, ASM_>), dim3(G), dim3(kTThreads), 0, st, n_rows, n_cols, (int)C,          \
                     (mu_tune_get("tpack_dbg") ? 1 : 0) | (mu_tune_get("tpack_split") == 1 ? 2 : 0), d_indptr,      \
                     d_indices, d_values, w.curs, d_cptr, d_inv, w.cnt, w.coltot, out)
      if (narrow && asm_loads) MU_FILL3(true, true);
      else if (narrow) MU_FILL3(true, false);
      else if (asm_loads) MU_FILL3(false, true);
      else MU_FILL3(false, false);
#undef MU_FILL3
    } else {
      hipLaunchKernelGGL(k_t_fill2, dim3(G), dim3(kTThreads), 0, st, n_rows, n_cols, (int)C,
                         mu_tune_get("tpack_abl"), d_indptr, d_indices, d_values, w.curs, d_cptr, d_inv,
                         w.cnt, w.coltot, out);
    }
    MU_CHECK_LAUNCH();
  }
  return MU_OK;
}

int mu_csr_tpack_fill_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr,
                          const int32_t* d_indices, const float* d_values, const int64_t* d_t_indptr,
                          int32_t* d_t_indices, float* d_t_values, void* d_work, size_t work_bytes,
                          void* stream) {
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative size");
  if (n_cols == 0 || nnz == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_t_indptr && d_t_indices && d_t_values && d_work, "null pointer");
  MU_REQUIRE(work_bytes >= mu_csr_tpack_worksize(n_rows, n_cols, nnz), "work buffer too small");
  const TOut out{nullptr, d_t_indices, d_t_values};
  return tpack_fill_impl(n_rows, n_cols, nnz, d_indptr, d_indices, d_values, d_t_indptr, nullptr, out,
                         d_work, (hipStream_t)stream);
}

/* tune tpack_dbg = 1: cycles (s_memtime) of the phases of the third-generation fill, summed over
 * workgroups and tiles: 0 header + scan, 1 count walk, 2 per-column prefix, 3 place walk (this wave),
 * 4 waiting for the other waves' place walks, 5 write-out.  reset != 0 clears the counters first. */
int mu_csr_tpack_phase_cycles(unsigned long long* h_out6, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (reset) {
    MU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_t_phase), z, sizeof(z)));
    return MU_OK;
  }
  MU_CHECK_HIP(hipDeviceSynchronize());
  MU_CHECK_HIP(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_t_phase), sizeof(z)));
  for (int i = 0; i < 6; ++i) h_out6[i] = z[i];
  return MU_OK;
}

/* the same with the row stream of X^T as the target (csrc/spmm_win.hip): output row c goes to
 * position inv[c] of the launch layout, ent[sptr[inv[c]] + i] = (cell, value bits) */
int mu_csr_tpack_fill_stream(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr,
                             const int32_t* d_indices, const float* d_values, const int64_t* d_sptr,
                             const int32_t* d_inv, void* d_ent, void* d_work, size_t work_bytes,
                             void* stream) {
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative size");
  if (n_cols == 0 || nnz == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_sptr && d_ent && d_work, "null pointer");
  MU_REQUIRE(work_bytes >= mu_csr_tpack_worksize(n_rows, n_cols, nnz), "work buffer too small");
  const TOut out{(unsigned long long*)d_ent, nullptr, nullptr};
  return tpack_fill_impl(n_rows, n_cols, nnz, d_indptr, d_indices, d_values, d_sptr, d_inv, out, d_work,
                         (hipStream_t)stream);
}

}  // extern "C"
