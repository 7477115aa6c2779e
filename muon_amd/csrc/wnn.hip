// Kernel bandwidths of the weighted-nearest-neighbour graph, one wave per cell.
//
// muon.pp.neighbors gives every cell i of a modality a bandwidth csigma_i = the mean Euclidean distance to
// the n_bandwidth_neighbors cells whose kNN sets overlap its own LEAST but do overlap, ties towards the
// larger distance (/root/reference/muon/_core/preproc.py:400-472: a nearest-neighbour search under the
// metric N (1 - jaccard distance) + (bbox - euclid) / bbox, :53-77, run through UMAP's NN-descent).  The
// candidates of a cell are the cells that share one of its neighbours; as tensor operations that was the
// sparse product A A^T of the binary kNN graph (3e7 pairs at 100 000 cells x 20 neighbours), a gather of
// both embeddings of every pair and three stable sorts over the pairs - 0.75 of the 1.1 s of a call.
//
// Here a wave takes a cell: it walks the cell's neighbours u and appends the cells that list u (the
// reverse graph) to an LDS buffer, sorts the buffer (bitonic, in place), turns runs into (candidate,
// |N(i) & N(j)|) pairs, computes the Jaccard distance, the Euclidean distance (x_i sits in LDS, x_j comes
// through the L2) and the reference's key for every candidate, and keeps the n_bw smallest (key, candidate)
// pairs in its lanes (slot l in lane l, the largest slot is the threshold a candidate has to beat: after the
// first few dozen candidates almost none does).  The sum is taken over the slots in lane order, so the mean
// can differ from the tensor formulation's in the last bits.  A cell whose candidate list (repetitions
// included) does not fit the buffer is flagged; the caller then runs the tensor formulation for the call.
#include "common.hpp"

namespace {

constexpr int kBwCap = 8192;     // candidate entries per wave, repetitions included (power of two; measured on 100 000
                                 // clustered cells x 19 neighbours: mean 1359, max 3946, 1035 / 2284 of them distinct -
                                 // a cell's neighbours are hubs: scripts/probes/wnn_candidate_stats.py)
constexpr int kBwWaves = 4;
constexpr int kBwPMax = 256;     // embedding dimensions held in LDS

__device__ __forceinline__ void bw_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

struct BwLds {
  int ids[kBwCap];  // candidates with repetitions, then sorted
  double xi[kBwPMax];
};

// (key, cell) pairs compare lexicographically
__device__ __forceinline__ bool bw_less(double ka, int ja, double kb, int jb) { return ka < kb || (ka == kb && ja < jb); }

__global__ __launch_bounds__(64 * kBwWaves) void k_wnn_bandwidth(
    int64_t n, int p, const double* __restrict__ X, const int64_t* __restrict__ g_ptr,
    const int32_t* __restrict__ g_idx, const int64_t* __restrict__ r_ptr, const int32_t* __restrict__ r_idx,
    int n_bw, double bbox, double* __restrict__ csigma, int32_t* __restrict__ overflow) {
  __shared__ BwLds lds[kBwWaves];
  const int lane = threadIdx.x & 63, wave = uniform32(threadIdx.x >> 6);
  BwLds& L = lds[wave];
  const int64_t n_waves = (int64_t)gridDim.x * kBwWaves;
  for (int64_t cell = (int64_t)blockIdx.x * kBwWaves + wave; cell < n; cell += n_waves) {
    const int64_t g0 = uniform64(g_ptr[cell]), g1 = uniform64(g_ptr[cell + 1]);
    const double deg_i = (double)(g1 - g0);
    bw_sync();  // (the previous cell is done with xi)
    for (int d = lane; d < p; d += 64) L.xi[d] = X[cell * p + d];
    // 1. the cells that list one of this cell's neighbours
    int m = 0;
    bool over = false;
    for (int64_t a = g0; a < g1 && !over; ++a) {
      const int u = uniform32(g_idx[a]);
      const int64_t r0 = uniform64(r_ptr[u]), r1 = uniform64(r_ptr[u + 1]);
      for (int64_t b0 = r0; b0 < r1; b0 += 64) {
        const int64_t b = b0 + lane;
        const int j = b < r1 ? r_idx[b] : -1;
        const bool take = j >= 0 && j != (int)cell;
        const unsigned long long mask = __ballot(take);
        const int cnt = __popcll(mask);
        if (m + cnt > kBwCap) {
          over = true;
          break;
        }
        if (take) L.ids[m + __popcll(mask & ((1ull << lane) - 1ull))] = j;
        m += cnt;
      }
    }
    if (over || m == 0) {  // uniform
      if (lane == 0) {
        csigma[cell] = __builtin_nan("");
        if (over) *overflow = 1;
      }
      continue;
    }
    // 2. sort (bitonic over the next power of two, padding = INT_MAX)
    int P = 64;
    while (P < m) P <<= 1;
    for (int t = m + lane; t < P; t += 64) L.ids[t] = 0x7fffffff;
    bw_sync();
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = lane; t < P; t += 64) {
          const int q = t ^ j;
          if (q > t) {
            const int a = L.ids[t], b = L.ids[q];
            const bool up = (t & k) == 0;
            if ((a > b) == up) {
              L.ids[t] = b;
              L.ids[q] = a;
            }
          }
        }
        bw_sync();
      }
    }
    // 3. runs -> (candidate, overlap) -> key; 4. the n_bw smallest (key, candidate) pairs live in the lanes:
    //    lane l < n_bw holds slot l (empty: key = +inf), (tk, tj, tl) = the largest slot = the threshold
    double sk = __builtin_inf(), se = 0.0;
    int sj = 0x7fffffff;
    double tk = __builtin_inf();
    int tj = 0x7fffffff, tl = 0;
    for (int t0 = 0; t0 < m; t0 += 64) {
      const int t = t0 + lane;
      const int id = t < m ? L.ids[t] : 0x7fffffff;
      const bool start = t < m && (t == 0 || L.ids[t - 1] != id);
      double key = __builtin_inf(), e = 0.0;
      if (start) {
        int inter = 1;
        while (t + inter < m && L.ids[t + inter] == id) ++inter;
        const double deg_j = (double)(g_ptr[id + 1] - g_ptr[id]);
        const double jac = 1.0 - (double)inter / (deg_i + deg_j - (double)inter);
        double s = 0.0;
        const double* xj = X + (int64_t)id * p;
        for (int d = 0; d < p; ++d) {
          const double df = L.xi[d] - xj[d];
          s += df * df;
        }
        e = sqrt(s);
        if (jac < 1.0) key = ((double)n - jac * (double)n) + (bbox - e) / bbox;
      }
      unsigned long long pend = __ballot(start && bw_less(key, id, tk, tj));
      while (pend) {  // uniform
        const int l = __builtin_ctzll(pend);
        pend &= pend - 1;
        const double ck = readlane_f64(key, l), ce = readlane_f64(e, l);
        const int cj = __builtin_amdgcn_readlane(id, l);
        if (!bw_less(ck, cj, tk, tj)) continue;  // the threshold moved in the meantime
        if (lane == tl) sk = ck, sj = cj, se = ce;
        // the new threshold: the largest of the n_bw slots
        double mk = lane < n_bw ? sk : -__builtin_inf();
        int mj = lane < n_bw ? sj : -1, ml = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const double ok = __shfl_xor(mk, off, 64);
          const int oj = __shfl_xor(mj, off, 64), ol = __shfl_xor(ml, off, 64);
          if (bw_less(mk, mj, ok, oj) || (mk == ok && mj == oj && ol < ml)) mk = ok, mj = oj, ml = ol;
        }
        tk = mk, tj = mj, tl = ml;
      }
    }
    // mean distance of the slots in use
    double sum = (lane < n_bw && sk < __builtin_inf()) ? se : 0.0;
    int got = (lane < n_bw && sk < __builtin_inf()) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sum += __shfl_xor(sum, off, 64);
      got += __shfl_xor(got, off, 64);
    }
    if (lane == 0) csigma[cell] = got ? sum / (double)got : __builtin_nan("");
  }
}

}  // namespace

extern "C" {

int mu_wnn_bandwidth_f64(int64_t n, int p, const double* d_X, const int64_t* d_g_indptr, const int32_t* d_g_indices,
                         const int64_t* d_r_indptr, const int32_t* d_r_indices, int n_bw, double bbox,
                         double* d_csigma, int32_t* d_overflow, void* stream) {
  MU_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && p >= 1 && p <= kBwPMax, "shape out of range (p <= 256)");
  MU_REQUIRE(n_bw >= 1 && n_bw <= 64, "1 <= n_bandwidth_neighbors <= 64");
  if (n == 0) return MU_OK;
  MU_REQUIRE(d_X && d_g_indptr && d_g_indices && d_r_indptr && d_r_indices && d_csigma && d_overflow, "null pointer");
  int64_t blocks = (n + kBwWaves - 1) / kBwWaves;
  const int64_t cap = (int64_t)mu_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_wnn_bandwidth, dim3((unsigned)blocks), dim3(64 * kBwWaves), 0, (hipStream_t)stream, n, p, d_X,
                     d_g_indptr, d_g_indices, d_r_indptr, d_r_indices, n_bw, bbox, d_csigma, d_overflow);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"

// ---- UMAP's smooth_knn_dist + membership strengths for a fixed-degree neighbour table, a thread per row ------------
// scanpy's `umap` connectivities behind sc.pp.neighbors / mu.pp.neighbors (/root/reference/muon/_core/preproc.py:
// 615-622 -> _compute_connectivities_umap; umap/umap_.py smooth_knn_dist, compute_membership_strengths with
// local_connectivity = 1, bandwidth = 1): rho = the first positive distance of the row, sigma by 64 bisection
// steps so that sum_{j >= 1} exp(-max(d_j - rho, 0) / sigma) = log2(n_neighbors), floors at 1e-3 of the row's /
// the table's mean distance, then the membership strength of every entry.  As tensor operations the bisection
// was 64 x ~10 launches over [n, k] tensors with a host check per step.  Arithmetic in f64 on f32-rounded
// distances, as umap (float32 distances) and the tensor version did.
namespace {

__global__ __launch_bounds__(256) void k_umap_strengths(int64_t n, int k, const double* __restrict__ dist,
                                                        const int64_t* __restrict__ idx, double target,
                                                        double mean_all, double* __restrict__ val) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const double* d = dist + r * k;
  double rho = 0.0, sum = 0.0;
  bool has = false;
  for (int j = 0; j < k; ++j) {
    const double x = (double)(float)d[j];
    sum += x;
    if (!has && x > 0.0) {
      rho = x;
      has = true;
    }
  }
  double lo = 0.0, hi = __builtin_inf(), mid = 1.0;
  for (int it = 0; it < 64; ++it) {
    double ps = 0.0;
    for (int j = 1; j < k; ++j) {
      const double x = (double)(float)d[j] - rho;
      ps += x > 0.0 ? exp(-x / mid) : 1.0;
    }
    if (fabs(ps - target) < 1e-5) break;
    if (ps > target) {
      hi = mid;
      mid = (lo + mid) * 0.5;  // (lo is the old one: it did not move)
    } else {
      lo = mid;
      mid = (hi == __builtin_inf()) ? mid * 2.0 : (mid + hi) * 0.5;
    }
  }
  double sigma = mid;
  const double floor_ = 1e-3 * (rho > 0.0 ? sum / (double)k : mean_all);
  sigma = sigma > floor_ ? sigma : floor_;
  for (int j = 0; j < k; ++j) {
    const double x = (double)(float)d[j];
    double v;
    if (idx[r * k + j] == r) v = 0.0;
    else if (x - rho <= 0.0 || sigma == 0.0) v = 1.0;
    else v = exp(-(x - rho) / sigma);
    val[r * k + j] = v;
  }
}

}  // namespace

extern "C" int mu_umap_strengths_f64(int64_t n, int k, const double* d_dist, const int64_t* d_idx, double target,
                                     double mean_all, double* d_val, void* stream) {
  MU_REQUIRE(n >= 0 && k >= 1, "shape");
  if (n == 0) return MU_OK;
  MU_REQUIRE(d_dist && d_idx && d_val, "null pointer");
  hipLaunchKernelGGL(k_umap_strengths, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, k,
                     d_dist, d_idx, target, mean_all, d_val);
  MU_CHECK_LAUNCH();
  return MU_OK;
}
