// Exhaustive nearest-neighbour search, the filter pass: which candidates of a panel beat a query's
// current threshold?
//
// muon.pp.neighbors (/root/reference/muon/_core/preproc.py:264-640) looks for neighbours three times per
// modality - the per-modality kNN graph it takes from scanpy (:366-373), the n_multineighbors candidates
// of the weighted graph (:525-533) - with UMAP's approximate NN-descent.  This framework searches
// exhaustively; as tensor operations that is a GEMM, an n x n distance matrix written to
// and read from HBM in panels, and a radix top-k over it: 1.1 of the 1.6 s of kernel time of a 100 000-cell
// call (profiles/r03_wnn_kernel_stats.md).  The distance matrix never has to exist: once a query has k
// candidates, only a vanishing fraction of the remaining rows can still enter its list.
//
// muon_amd/_core/preproc.py walks the (randomly permuted) candidates in panels of doubling size; the first
// panel is searched densely and leaves every query a threshold tau_i = its k-th smallest distance so far;
// for every later panel THIS kernel computes the panel's squared distances tile by tile on the f64 matrix
// cores (v_mfma_f64_16x16x4_f64: |q|^2 + |c|^2 - 2 q.c, the form the tensor version used), compares them
// in registers with tau_i and appends the (position, distance) pairs that pass to the query's buffer -
// expected: k per panel and query, whatever the panel size, because the panel doubles what the query has
// seen; the host merges the buffers into the lists (a top-k over ~4 k entries per query instead of n) and
// lowers the thresholds.  A buffer that overflows (ties, adversarial order) is reported through its count
// and that panel is redone densely for those queries.
//
// One workgroup = 64 queries (a wave: 16 queries x 4 candidate sub-tiles of 16), 64 candidates per tile
// staged in LDS with the queries; operands padded to a multiple of 4 columns by the caller.
#include "common.hpp"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kKnnT = 64;        // queries per workgroup, candidates per tile
constexpr int kKnnThreads = 256;

__global__ __launch_bounds__(kKnnThreads) void k_knn_filter(
    int64_t n_q, int64_t c_lo, int64_t c_hi, int p_pad, const double* __restrict__ Xq,
    const double* __restrict__ Xc, const double* __restrict__ sqq, const double* __restrict__ sqc,
    const double* __restrict__ thr, const int32_t* __restrict__ self_pos, int cap,
    int32_t* __restrict__ buf_pos, double* __restrict__ buf_d, int32_t* __restrict__ cnt) {
  extern __shared__ double smem[];
  const int ld = p_pad + 1;  // odd row stride (in doubles): the 16 rows of an operand fall into different banks
  double* Qs = smem;                   // [64][ld]
  double* Cs = Qs + kKnnT * ld;        // [64][ld]
  double* s_sqc = Cs + kKnnT * ld;     // [64]
  int* s_cnt = reinterpret_cast<int*>(s_sqc + kKnnT);  // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane >> 4, lc = lane & 15;
  const int64_t q0 = (int64_t)blockIdx.x * kKnnT;
  const int trow = tid >> 2, tk = (tid & 3) * 2;  // tile loads: four threads per row, 16 bytes each per step

  {  // the queries of this workgroup
    const int64_t q = q0 + trow;
    for (int k = tk; k < p_pad; k += 8) {
      d2 v = d2{0.0, 0.0};
      if (q < n_q) v = *reinterpret_cast<const d2*>(Xq + q * p_pad + k);
      Qs[trow * ld + k] = v[0];
      Qs[trow * ld + k + 1] = v[1];
    }
    if (tid < kKnnT) s_cnt[tid] = 0;
  }
  // this lane's four query rows: 16 wave + lr + 4 r (the D layout of v_mfma_f64_16x16x4_f64: register r of
  // lane (lr, lc) holds row lr + 4 r, column lc)
  double t_i[4];   // tau_i - |q_i|^2: a candidate passes when |c|^2 - 2 q.c < t_i
  double sq_i[4];
  int self_i[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t q = q0 + 16 * wave + lr + 4 * r;
    const bool ok = q < n_q;
    sq_i[r] = ok ? sqq[q] : 0.0;
    t_i[r] = ok ? thr[q] - sq_i[r] : -1e300;  // (rows past the end never pass)
    self_i[r] = ok ? self_pos[q] : -1;
  }

  for (int64_t c0 = c_lo; c0 < c_hi; c0 += kKnnT) {
    __syncthreads();  // the previous tile is consumed (first pass: the query tile and the counters are written)
    {
      const int64_t c = c0 + trow;
      for (int k = tk; k < p_pad; k += 8) {
        d2 v = d2{0.0, 0.0};
        if (c < c_hi) v = *reinterpret_cast<const d2*>(Xc + c * p_pad + k);
        Cs[trow * ld + k] = v[0];
        Cs[trow * ld + k + 1] = v[1];
      }
      if (tid < kKnnT) s_sqc[tid] = (c0 + tid < c_hi) ? sqc[c0 + tid] : 0.0;
    }
    __syncthreads();
    d4 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = d4{0.0, 0.0, 0.0, 0.0};
    const double* qa = Qs + (16 * wave + lc) * ld + lr;  // A operand: row i = lc, column 4 step + lr
    const double* cb = Cs + lc * ld + lr;                // B operand of sub-tile s: row 16 s + lc
    for (int k = 0; k < p_pad; k += 4) {
      const double a = qa[k];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, cb[16 * s * ld + k], acc[s], 0, 0, 0);
    }
    // acc[s][r] = q_{16 wave + lr + 4 r} . c_{c0 + 16 s + lc}
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int64_t c = c0 + 16 * s + lc;
      const double sc = s_sqc[16 * s + lc];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double part = sc - 2.0 * acc[s][r];
        if (part < t_i[r] && c < c_hi && (int)c != self_i[r]) {
          const int qi = 16 * wave + lr + 4 * r;
          const int pos = atomicAdd(&s_cnt[qi], 1);
          if (pos < cap) {
            const int64_t o = (q0 + qi) * (int64_t)cap + pos;
            buf_pos[o] = (int32_t)c;
            buf_d[o] = part + sq_i[r];
          }
        }
      }
    }
  }
  __syncthreads();
  if (tid < kKnnT && q0 + tid < n_q) cnt[q0 + tid] = s_cnt[tid];
}


// ---- the merge after a filter pass -------------------------------------------------------------------------------
// list [kc] ++ buffer [min(cnt, cap)] -> the kc smallest (distance, position) pairs and the new threshold, a wave per
// query: the pairs go to LDS, a bitonic sort over the next power of two (padding = +inf), the first kc come back.
// (r04; was: a mask, two concatenations, torch's top-k over 4 kc columns and a gather per panel - the radix passes were
// a quarter of a search's kernel time.)  Ties are broken by position, so the result does not depend on the order in
// which the filter's workgroups appended to the buffer.
constexpr int kMergeWaves = 4;
constexpr int kMergeMax = 1024;  // kc + cap must fit

__device__ __forceinline__ void merge_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64 * kMergeWaves) void k_knn_merge(int64_t n_q, int kc, int cap,
                                                                const double* __restrict__ cur_d,
                                                                const int64_t* __restrict__ cur_p,
                                                                const double* __restrict__ buf_d,
                                                                const int32_t* __restrict__ buf_pos,
                                                                const int32_t* __restrict__ cnt,
                                                                double* __restrict__ out_d, int64_t* __restrict__ out_p,
                                                                double* __restrict__ thr) {
  __shared__ double keys[kMergeWaves][kMergeMax];
  __shared__ int vals[kMergeWaves][kMergeMax];
  const int lane = threadIdx.x & 63, wave = uniform32(threadIdx.x >> 6);
  const int64_t q = (int64_t)blockIdx.x * kMergeWaves + wave;
  if (q >= n_q) return;
  double* K = keys[wave];
  int* V = vals[wave];
  const int c = cnt[q] < cap ? cnt[q] : cap;  // (an overflowed row is redone by the caller)
  const int m = kc + c;
  int P = 64;
  while (P < m) P <<= 1;
  for (int t = lane; t < P; t += 64) {
    double d = __builtin_inf();
    int v = 0x7fffffff;
    if (t < kc) {
      d = cur_d[q * kc + t];
      v = (int)cur_p[q * kc + t];
    } else if (t < m) {
      d = buf_d[q * cap + (t - kc)];
      v = buf_pos[q * cap + (t - kc)];
    }
    K[t] = d;
    V[t] = v;
  }
  merge_sync();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < P; t += 64) {
        const int u = t ^ j;
        if (u > t) {
          const double a = K[t], b = K[u];
          const int va = V[t], vb = V[u];
          const bool gt = a > b || (a == b && va > vb);  // (no NaN: distances are finite or +inf)
          const bool up = (t & k) == 0;
          if (gt == up) {
            K[t] = b;
            K[u] = a;
            V[t] = vb;
            V[u] = va;
          }
        }
      }
      merge_sync();
    }
  }
  for (int t = lane; t < kc; t += 64) {
    out_d[q * kc + t] = K[t];
    out_p[q * kc + t] = (int64_t)V[t];
  }
  if (lane == 0) thr[q] = K[kc - 1];
}

}  // namespace

extern "C" {

int mu_knn_filter_f64(int64_t n_q, int64_t c_lo, int64_t c_hi, int p_pad, const double* d_Xq, const double* d_Xc,
                      const double* d_sqq, const double* d_sqc, const double* d_thr, const int32_t* d_self_pos,
                      int cap, int32_t* d_buf_pos, double* d_buf_d, int32_t* d_cnt, void* stream) {
  MU_REQUIRE(n_q >= 0 && c_lo >= 0 && c_hi >= c_lo && c_hi < ((int64_t)1 << 31), "bad range");
  MU_REQUIRE(p_pad >= 4 && p_pad % 4 == 0 && p_pad <= 1024, "columns must be padded to a multiple of 4 (<= 1024)");
  MU_REQUIRE(cap >= 1, "buffer capacity");
  if (n_q == 0) return MU_OK;
  MU_REQUIRE(d_Xq && d_Xc && d_sqq && d_sqc && d_thr && d_self_pos && d_buf_pos && d_buf_d && d_cnt, "null pointer");
  const size_t lds = (size_t)(2 * kKnnT * (p_pad + 1) + kKnnT) * sizeof(double) + kKnnT * sizeof(int);
  MU_REQUIRE(lds <= 160 * 1024, "operand rows too wide for the LDS tiles");
  if (lds > 64 * 1024)
    MU_CHECK_HIP(hipFuncSetAttribute((const void*)k_knn_filter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t wgs = (n_q + kKnnT - 1) / kKnnT;
  hipLaunchKernelGGL(k_knn_filter, dim3((unsigned)wgs), dim3(kKnnThreads), lds, (hipStream_t)stream, n_q, c_lo, c_hi,
                     p_pad, d_Xq, d_Xc, d_sqq, d_sqc, d_thr, d_self_pos, cap, d_buf_pos, d_buf_d, d_cnt);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_knn_merge_f64(int64_t n_q, int kc, int cap, const double* d_cur_d, const int64_t* d_cur_p, const double* d_buf_d,
                     const int32_t* d_buf_pos, const int32_t* d_cnt, double* d_out_d, int64_t* d_out_p, double* d_thr,
                     void* stream) {
  MU_REQUIRE(n_q >= 0 && kc >= 1 && cap >= 1 && kc + cap <= kMergeMax, "kc + cap must be <= 1024");
  if (n_q == 0) return MU_OK;
  MU_REQUIRE(d_cur_d && d_cur_p && d_buf_d && d_buf_pos && d_cnt && d_out_d && d_out_p && d_thr, "null pointer");
  MU_REQUIRE(d_out_d != d_cur_d && d_out_p != d_cur_p, "the merge is not in place");
  const int64_t wgs = (n_q + kMergeWaves - 1) / kMergeWaves;
  hipLaunchKernelGGL(k_knn_merge, dim3((unsigned)wgs), dim3(64 * kMergeWaves), 0, (hipStream_t)stream, n_q, kc, cap,
                     d_cur_d, d_cur_p, d_buf_d, d_buf_pos, d_cnt, d_out_d, d_out_p, d_thr);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
