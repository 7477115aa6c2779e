// Device-side synthetic planted-topic count matrix (bench / tests only).
// Not part of the reference's path: it exists because the full-scale configuration
// (1M x 200k, 6e9 stored entries) cannot be generated on, or shipped from, the host.
//
// Model: cell i has topic t(i) and depth_i ~ max(50, LogNormal(ln(1.15*density*d), 0.3));
// peak j has background weight bg_j ~ Gamma(2,1); topic t up-weights a 5 % subset of the
// peaks by w_tj ~ Gamma(2,1); count_ij ~ Poisson(depth_i * (0.5*bg_j/(2d) + 0.5*w_tj/(0.1d))).
// Every random number is a hash of (seed, global row, column), so a row shard generated on
// any rank equals the same rows of the single-GPU matrix, and columns come out sorted.
// n_topics = 0 selects the UNSTRUCTURED variant of SURVEY.md 8d (raw kernel throughput only): every
// (row, column) is stored with probability `density`, value 1 + Poisson(0.5) capped at 4.
#include "common.hpp"

struct SynthRow {
  float depth;
  int topic;
};

__device__ __forceinline__ SynthRow synth_row(uint64_t seed, int64_t grow, int64_t n_cols,
                                              int n_topics, double density) {
  const uint64_t h = splitmix64(seed * 0x9E3779B97F4A7C15ull + 0x1000000000ull + (uint64_t)grow);
  const uint64_t h2 = splitmix64(h);
  SynthRow r;
  r.topic = n_topics > 0 ? (int)(h2 % (uint64_t)n_topics) : -1;
  if (n_topics <= 0) {
    r.depth = 0.f;
    return r;
  }
  const float u1 = u01(h), u2 = u01(h << 24 | (h2 >> 40));
  const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
  const float d = expf(logf((float)(1.15 * density * (double)n_cols)) + 0.3f * z);
  r.depth = fmaxf(n_cols >= 2000 ? 50.0f : 5.0f, d);
  return r;
}

// Poisson mean of (row, column j) divided by depth
__device__ __forceinline__ float synth_p(uint64_t seed, int topic, int64_t j, float inv2d, float inv01d) {
  const uint64_t hb = splitmix64(seed * 0xD1342543DE82EF95ull + 0x2000000000ull + (uint64_t)j);
  const float bg = -logf(u01(hb)) - logf(u01(hb << 24));
  float p = 0.5f * bg * inv2d;
  const uint64_t ht = splitmix64((seed + 0x51ull) * 0x9E3779B97F4A7C15ull +
                                 ((uint64_t)topic << 40) + (uint64_t)j);
  if ((ht & 0xFFFFull) < 3277ull) {  // 5 % of the peaks
    const float w = -logf(u01(ht)) - logf(u01(ht << 24));
    p += 0.5f * w * inv01d;
  }
  return p;
}

__device__ __forceinline__ int synth_count(uint64_t seed, int64_t grow, int64_t j, float lam) {
  const uint64_t hu = splitmix64((seed + 0xA5ull) * 0xBF58476D1CE4E5B9ull +
                                 (uint64_t)grow * 0x100000001B3ull + (uint64_t)j);
  const float u = u01(hu);
  const float p0 = expf(-lam);
  if (u < p0) return 0;
  float cdf = p0 * (1.0f + lam);
  if (u < cdf) return 1;
  cdf += p0 * lam * lam * 0.5f;
  if (u < cdf) return 2;
  cdf += p0 * lam * lam * lam * (1.0f / 6.0f);
  if (u < cdf) return 3;
  return 4;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_synth(int64_t row0, int64_t n_rows, int64_t n_cols,
                                               int n_topics, double density, uint64_t seed,
                                               int64_t* __restrict__ row_nnz,
                                               const int64_t* __restrict__ indptr,
                                               int32_t* __restrict__ indices,
                                               float* __restrict__ values) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = uniform64(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float inv2d = 1.0f / (2.0f * (float)n_cols), inv01d = 1.0f / (0.1f * (float)n_cols);
  for (int64_t row = wave0; row < n_rows; row += n_waves) {
    const int64_t grow = row0 + row;
    const SynthRow r = synth_row(seed, grow, n_cols, n_topics, density);
    int64_t pos = FILL ? indptr[row] : 0;
    for (int64_t j0 = 0; j0 < n_cols; j0 += 64) {
      const int64_t j = j0 + lane;
      int cnt = 0;
      if (j < n_cols) {
        if (r.topic >= 0) {
          cnt = synth_count(seed, grow, j, r.depth * synth_p(seed, r.topic, j, inv2d, inv01d));
        } else {
          const uint64_t hu = splitmix64((seed + 0xA5ull) * 0xBF58476D1CE4E5B9ull +
                                         (uint64_t)grow * 0x100000001B3ull + (uint64_t)j);
          if (u01(hu) < (float)density) cnt = 1 + synth_count(seed + 7, grow, j, 0.5f);
          cnt = cnt > 4 ? 4 : cnt;
        }
      }
      const unsigned long long m = __ballot(cnt > 0);
      if (FILL && cnt > 0) {
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        indices[pos + rank] = (int32_t)j;
        values[pos + rank] = (float)cnt;
      }
      pos += __popcll(m);
    }
    if (!FILL && lane == 0) row_nnz[row] = pos;
  }
}

static inline unsigned synth_blocks(int64_t n_rows) {
  int64_t blocks = (n_rows + 3) / 4;
  const int64_t cap = (int64_t)mu_num_cus() * 32;
  if (blocks > cap) blocks = cap;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

extern "C" {

int mu_synth_row_nnz(int64_t row0, int64_t n_rows, int64_t n_cols, int n_topics, double density,
                     uint64_t seed, int64_t* d_row_nnz, void* stream) {
  MU_REQUIRE(n_rows >= 0 && n_cols > 0 && n_topics >= 0 && density > 0, "bad arguments");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_row_nnz, "null pointer");
  hipLaunchKernelGGL(k_synth<false>, dim3(synth_blocks(n_rows)), dim3(256), 0, (hipStream_t)stream,
                     row0, n_rows, n_cols, n_topics, density, seed, d_row_nnz, nullptr, nullptr,
                     nullptr);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_synth_fill(int64_t row0, int64_t n_rows, int64_t n_cols, int n_topics, double density,
                  uint64_t seed, const int64_t* d_indptr, int32_t* d_indices, float* d_values,
                  void* stream) {
  MU_REQUIRE(n_rows >= 0 && n_cols > 0 && n_topics >= 0 && density > 0, "bad arguments");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_indices && d_values, "null pointer");
  hipLaunchKernelGGL(k_synth<true>, dim3(synth_blocks(n_rows)), dim3(256), 0, (hipStream_t)stream,
                     row0, n_rows, n_cols, n_topics, density, seed, nullptr, d_indptr, d_indices,
                     d_values);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
