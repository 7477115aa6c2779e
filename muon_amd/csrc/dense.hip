// Dense blocks of the LSI solver: the only MFMA work on the path.
//   gram  : G = A^T A (f64 accumulate, v_mfma_f64_16x16x4_f64) + column sums
//   apply : Out = A * M + bias          (v_mfma_f32_16x16x4_f32)
//   randn : counter-based standard normals for the start block
// Together gram + (host Cholesky of the B x B Gram) + apply form the CholeskyQR step that
// replaces the dense QR / SVD tail of scipy svds (_svds.py:513-539); A is n x B with
// B <= 64, so these kernels stream A once and are bound by HBM, not by the matrix cores.
#include "common.hpp"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kGramThreads = 256;  // 4 waves

// MFMA f64 16x16x4 operand maps (cdna_hip_programming.md §3):
//   A operand: lane l holds A[i = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D      : reg r of lane l is C[row = (l >> 4) + 4 r][col = l & 15]
// For G = A^T A a 16x16 tile (ti, tj) takes both operands from the same four rows of A:
//   a = A[r + (l >> 4)][16 ti + (l & 15)],  b = A[r + (l >> 4)][16 tj + (l & 15)].
template <int B>
__global__ __launch_bounds__(kGramThreads) void k_gram_partial(int64_t n_rows,
                                                               const float* __restrict__ A,
                                                               double* __restrict__ partial) {
  constexpr int T = B / 16;
  constexpr int NT = T * (T + 1) / 2;  // upper-triangular tiles only
  __shared__ double red[B * B + B];
  const int lane = threadIdx.x & 63, wave = uniform32(threadIdx.x >> 6);
  const int lr = lane >> 4, lc = lane & 15;

  d4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double cs[T];
#pragma unroll
  for (int t = 0; t < T; ++t) cs[t] = 0.0;

  // rows are dealt to waves in groups of four
  const int64_t n_groups = (n_rows + 3) / 4;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
  const int64_t gstride = (int64_t)gridDim.x * 4;
  float xn[T];
  {
    const int64_t row = gw * 4 + lr;
#pragma unroll
    for (int t = 0; t < T; ++t) xn[t] = (gw < n_groups && row < n_rows) ? A[row * B + 16 * t + lc] : 0.f;
  }
  for (int64_t grp = gw; grp < n_groups; grp += gstride) {
    double x[T];
#pragma unroll
    for (int t = 0; t < T; ++t) x[t] = (double)xn[t];
    const int64_t nrow = (grp + gstride) * 4 + lr;
    const bool more = (grp + gstride < n_groups) && (nrow < n_rows);
#pragma unroll
    for (int t = 0; t < T; ++t) xn[t] = more ? A[nrow * B + 16 * t + lc] : 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) cs[t] += x[t];
    int k = 0;
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
      for (int tj = ti; tj < T; ++tj) {
        acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ti], x[tj], acc[k], 0, 0, 0);
        ++k;
      }
  }

  // reduce the four waves through LDS (fixed order), then write this workgroup's partial
  for (int i = threadIdx.x; i < B * B + B; i += kGramThreads) red[i] = 0.0;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      int k = 0;
#pragma unroll
      for (int ti = 0; ti < T; ++ti)
#pragma unroll
        for (int tj = ti; tj < T; ++tj) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int gi = 16 * ti + lr + 4 * r, gj = 16 * tj + lc;
            red[gi * B + gj] += acc[k][r];
          }
          ++k;
        }
      // column sums: add the four row-groups of the wave
#pragma unroll
      for (int t = 0; t < T; ++t) {
        double v = cs[t];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lr == 0) red[B * B + 16 * t + lc] += v;
      }
    }
    __syncthreads();
  }
  double* dst = partial + (int64_t)blockIdx.x * (B * B + B);
  for (int i = threadIdx.x; i < B * B + B; i += kGramThreads) dst[i] = red[i];
}

template <int B>
__global__ __launch_bounds__(256) void k_gram_reduce(int n_partials, const double* __restrict__ partial,
                                                     double* __restrict__ G,
                                                     double* __restrict__ colsum) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * B + B) return;
  int src = e;
  if (e < B * B) {
    const int i = e / B, j = e % B;
    // only tiles with ti <= tj were computed; mirror the rest
    if ((i / 16) > (j / 16)) src = j * B + i;
  }
  double acc = 0.0;
  for (int p = 0; p < n_partials; ++p) acc += partial[(int64_t)p * (B * B + B) + src];
  if (e < B * B) G[e] = acc; else if (colsum) colsum[e - B * B] = acc;
}

// C = A^T Bm (both n x B f32, f64 accumulate): every one of the T x T tiles, no column sums.
// Used to measure the angle between the Ritz subspaces of consecutive iterations (lsi stopping rule).
template <int B>
__global__ __launch_bounds__(kGramThreads) void k_gram_cross_partial(int64_t n_rows,
                                                                     const float* __restrict__ A,
                                                                     const float* __restrict__ Bm,
                                                                     double* __restrict__ partial) {
  constexpr int T = B / 16;
  __shared__ double red[B * B + B];
  const int lane = threadIdx.x & 63, wave = uniform32(threadIdx.x >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  d4 acc[T * T];
#pragma unroll
  for (int i = 0; i < T * T; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  const int64_t n_groups = (n_rows + 3) / 4;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
  const int64_t gstride = (int64_t)gridDim.x * 4;
  // (the loads of the next four rows are in flight while the 16 MFMAs of the current ones issue:
  //  with one dependent load -> MFMA chain per iteration the kernel ran at memory latency)
  float xn[T], yn[T];
  {
    const int64_t row = gw * 4 + lr;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      xn[t] = (gw < n_groups && row < n_rows) ? A[row * B + 16 * t + lc] : 0.f;
      yn[t] = (gw < n_groups && row < n_rows) ? Bm[row * B + 16 * t + lc] : 0.f;
    }
  }
  for (int64_t grp = gw; grp < n_groups; grp += gstride) {
    double x[T], y[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      x[t] = (double)xn[t];
      y[t] = (double)yn[t];
    }
    const int64_t nrow = (grp + gstride) * 4 + lr;
    const bool more = (grp + gstride < n_groups) && (nrow < n_rows);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      xn[t] = more ? A[nrow * B + 16 * t + lc] : 0.f;
      yn[t] = more ? Bm[nrow * B + 16 * t + lc] : 0.f;
    }
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
      for (int tj = 0; tj < T; ++tj)
        acc[ti * T + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ti], y[tj], acc[ti * T + tj], 0, 0, 0);
  }
  for (int i = threadIdx.x; i < B * B + B; i += kGramThreads) red[i] = 0.0;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ti = 0; ti < T; ++ti)
#pragma unroll
        for (int tj = 0; tj < T; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(16 * ti + lr + 4 * r) * B + 16 * tj + lc] += acc[ti * T + tj][r];
    }
    __syncthreads();
  }
  double* dst = partial + (int64_t)blockIdx.x * (B * B + B);
  for (int i = threadIdx.x; i < B * B + B; i += kGramThreads) dst[i] = red[i];
}

// plain sum of the folded partials (no mirroring: the cross-Gram is not symmetric)
template <int B>
__global__ __launch_bounds__(256) void k_gram_cross_reduce(int n_partials, const double* __restrict__ partial,
                                                           double* __restrict__ C) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * B) return;
  double acc = 0.0;
  for (int p = 0; p < n_partials; ++p) acc += partial[(int64_t)p * (B * B + B) + e];
  C[e] = acc;
}

// first level of the partial reduction: fold chunk y of the workgroup partials (fixed order)
constexpr int kGramFold = 32;
template <int B>
__global__ __launch_bounds__(256) void k_gram_fold(int n_partials, const double* __restrict__ partial,
                                                   double* __restrict__ folded) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * B + B) return;
  const int chunk = (n_partials + kGramFold - 1) / kGramFold;
  const int p0 = blockIdx.y * chunk;
  const int p1 = (p0 + chunk) < n_partials ? (p0 + chunk) : n_partials;
  double acc = 0.0;
  for (int p = p0; p < p1; ++p) acc += partial[(int64_t)p * (B * B + B) + e];
  folded[(int64_t)blockIdx.y * (B * B + B) + e] = acc;
}

static inline int gram_blocks(int64_t n_rows) {
  int64_t groups = (n_rows + 15) / 16;  // one workgroup step = 16 rows
  int64_t blocks = groups < 1 ? 1 : groups;
  const int per_cu = mu_tune_get("gram_wg") > 0 ? mu_tune_get("gram_wg") : 2;  // (probe: 1, 2, 4, 8 -> 81, 88, 104, 134 us per cross-Gram at 200k rows)
  const int64_t cap = (int64_t)mu_num_cus() * per_cu;
  // (r03: capping small inputs at kGramFold blocks so that the reduction takes the partials directly -
  //  two launches instead of three - made the 30 000-row Grams of a 10k x 30k call slower than the
  //  launch it saved: wait 3.7 -> 4.6 ms per call; the direct path below stays for inputs that have
  //  that few blocks anyway)
  return (int)(blocks > cap ? cap : blocks);
}

// MFMA f32 16x16x4: A operand lane l = A[i = l & 15][k = l >> 4]; B operand = B[k = l >> 4][j = l & 15];
// C/D reg r of lane l = C[row = 4 (l >> 4) + r][col = l & 15].
template <int B>
__global__ __launch_bounds__(256) void k_dense_apply(int64_t n_rows, const float* __restrict__ A,
                                                     const float* __restrict__ M,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ Out) {
  constexpr int T = B / 16;   // output column tiles
  constexpr int KS = B / 4;   // k steps
  const int lane = threadIdx.x & 63, wave = uniform32(threadIdx.x >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  float bm[KS][T];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int t = 0; t < T; ++t) bm[ks][t] = M[(4 * ks + lr) * B + 16 * t + lc];
  float bb[T];
#pragma unroll
  for (int t = 0; t < T; ++t) bb[t] = bias ? bias[16 * t + lc] : 0.f;

  const int64_t n_tiles = (n_rows + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t r0 = tile * 16;
    const int64_t arow = r0 + lc;
    float a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = (arow < n_rows) ? A[arow * B + 4 * ks + lr] : 0.f;
    f4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f4{bb[t], bb[t], bb[t], bb[t]};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < T; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bm[ks][t], acc[t], 0, 0, 0);
    // all of this wave's reads of the tile are complete (consumed by the MFMAs above) before
    // the stores below, so Out may alias A
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t orow = r0 + 4 * lr + r;
        if (orow < n_rows) Out[orow * B + 16 * t + lc] = acc[t][r];
      }
  }
}

// Z <- Z - A * C with the B x B coefficients in f64 (the output of mu_gram_cross_f32): one block of a
// Gram-Schmidt projection.  Same arithmetic as k_dense_apply into a temporary followed by a
// subtraction (the product is rounded to f32, then subtracted), without the temporary, the
// conversion of C and the subtraction as separate launches.
template <int B>
__global__ __launch_bounds__(256) void k_dense_project(int64_t n_rows, const float* __restrict__ A,
                                                       const double* __restrict__ C,
                                                       float* __restrict__ Z) {
  constexpr int T = B / 16;
  constexpr int KS = B / 4;
  const int lane = threadIdx.x & 63, wave = uniform32(threadIdx.x >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  float bm[KS][T];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int t = 0; t < T; ++t) bm[ks][t] = (float)C[(4 * ks + lr) * B + 16 * t + lc];
  const int64_t n_tiles = (n_rows + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t r0 = tile * 16;
    const int64_t arow = r0 + lc;
    float a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = (arow < n_rows) ? A[arow * B + 4 * ks + lr] : 0.f;
    f4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < T; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bm[ks][t], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t orow = r0 + 4 * lr + r;
        if (orow < n_rows) Z[orow * B + 16 * t + lc] -= acc[t][r];
      }
  }
}

__global__ __launch_bounds__(256) void k_randn(int64_t count, uint64_t seed, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t h1 = splitmix64(seed ^ (uint64_t)(2 * i) * 0xD6E8FEB86659FD93ull);
    const uint64_t h2 = splitmix64(h1 + (uint64_t)(2 * i + 1));
    const float u1 = u01(h1), u2 = u01(h2);
    out[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
  }
}

// ---------------------------------------------------------------------------------
// CholeskyQR's small step on the device: M = R^-1 (upper triangular, f32, zero outside the leading
// w x w block) for the Gram G = R^T R of a block (f64, B x B, B <= 64).  One wave, LDS-resident:
// the host version (numpy cholesky + inv) costs two device -> host round trips per
// orthonormalisation, which is what a 10k x 30k lsi() spends 20 % of its time in.
// A pivot that is not safely positive (a block with dependent columns: Krylov space exhausted)
// sets *flag and is clamped, so that nothing non-finite is produced; the caller checks the flag
// with its next host read and redoes the call on the host path.
// ---------------------------------------------------------------------------------
constexpr int kCholT = 256;
constexpr int kCholNb = 16;  // block size of the blocked factorisation
// r03: blocked right-looking Cholesky of the (identity-padded) 64 x 64 matrix in LDS, 16-column panels:
// per panel, wave 0 factors the diagonal block and inverts it (16 dependent steps, wave-synchronous: LDS
// operations of a wave execute in order, no workgroup barrier), then all four waves form the panel below it
// (L21 = A21 L11^-T as a product with the inverted block) and the trailing update - THREE workgroup barriers
// per panel, twelve per call: 87 us per call in the kernel statistics (c2: 12 calls per lsi()).  The first r03
// version swept column by column with three barriers per column (192 per call): 109 us, r02's one-wave version
// 118 us.  What is left is the serial chain of wave 0 - 64 pivots, each a sqrt, a divide and three LDS round
// trips.  L^-1: diagonal blocks from the factorisation, the blocks below
// them block column by block column, one wave per block column, no barriers in between.
__device__ __forceinline__ void chol_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(kCholT) void k_chol_rinv(int B, int w, const double* __restrict__ G,
                                                      float* __restrict__ M, int* __restrict__ flag) {
  constexpr int NB = kCholNb;
  __shared__ double A[64][65];          // lower triangle: the matrix, then L
  __shared__ double X[64][65];          // L^-1 (lower)
  __shared__ double P[64][NB + 1];      // the panel below the diagonal block
  __shared__ double Tm[4][NB][NB + 1];  // per-wave scratch of the inverse's block products
  __shared__ int s_bad;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_bad = 0;
  double dmax = 0.0;
  for (int i = 0; i < w; ++i) {
    const double g = G[(int64_t)i * B + i];
    dmax = g > dmax ? g : dmax;
  }
  const double tiny = dmax > 0.0 ? dmax * 1e-13 : 1.0;
  const double pad = dmax > 0.0 ? dmax : 1.0;  // rows past w: a diagonal that passes the pivot test
  for (int e = t; e < 64 * 64; e += kCholT) {
    const int i = e >> 6, j = e & 63;
    A[i][j] = (i < w && j < w) ? (j <= i ? G[(int64_t)i * B + j] : 0.0) : (i == j ? pad : 0.0);
    X[i][j] = 0.0;
  }
  __syncthreads();

  for (int k0 = 0; k0 < 64; k0 += NB) {
    if (wave == 0) {
      // diagonal block: unblocked Cholesky in place
      for (int s = 0; s < NB; ++s) {
        const int k = k0 + s;
        double piv = A[k][k];
        const bool bad = !(piv > tiny);  // also catches NaN: a block with dependent columns
        if (bad) piv = tiny;
        if (bad && lane == 0) s_bad = 1;
        const double r = sqrt(piv);
        chol_wave_sync();  // (every lane has read the pivot before lane s overwrites it)
        if (lane < NB) {
          const int i = k0 + lane;
          if (lane == s) A[k][k] = r;
          else if (lane > s) A[i][k] = A[i][k] / r;
        }
        chol_wave_sync();
        for (int e = lane; e < NB * NB; e += 64) {
          const int i = k0 + e / NB, j = k0 + e % NB;
          if (j > k && j <= i) A[i][j] -= A[i][k] * A[j][k];
        }
        chol_wave_sync();
      }
      // its inverse, column by column: lane c solves D x = e_c by forward substitution
      if (lane < NB) {
        double x[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          double v = (i == lane) ? 1.0 : 0.0;
#pragma unroll
          for (int j = 0; j < NB; ++j)
            if (j < i) v -= A[k0 + i][k0 + j] * x[j];
          x[i] = (i >= lane) ? v / A[k0 + i][k0 + i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) X[k0 + i][k0 + lane] = x[i];
      }
    }
    __syncthreads();
    const int below = 64 - k0 - NB;  // rows under the diagonal block
    // panel: L21[i][c] = sum_{j <= c} A21[i][j] invD[c][j]
    for (int e = t; e < below * NB; e += kCholT) {
      const int i = k0 + NB + e / NB, c = e % NB;
      double v = 0.0;
      for (int j = 0; j <= c; ++j) v += A[i][k0 + j] * X[k0 + c][k0 + j];
      P[i][c] = v;
    }
    __syncthreads();
    // trailing update A22 -= L21 L21^T (lower part), and L21 takes its place in A
    for (int e = t; e < below * below; e += kCholT) {
      const int i = k0 + NB + e / below, j = k0 + NB + e % below;
      if (j <= i) {
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < NB; ++c) v += P[i][c] * P[j][c];
        A[i][j] -= v;
      }
    }
    for (int e = t; e < below * NB; e += kCholT) {
      const int i = k0 + NB + e / NB, c = e % NB;
      A[i][k0 + c] = P[i][c];
    }
    __syncthreads();
  }
  // L^-1 below the diagonal blocks: wave kb owns block column kb;  X_ik = - X_ii (sum_{j = k}^{i-1} L_ij X_jk)
  {
    const int kb = wave;
    for (int ib = kb + 1; ib < 64 / NB; ++ib) {
      for (int e = lane; e < NB * NB; e += 64) {
        const int r = e / NB, c = e % NB;
        double v = 0.0;
        for (int jb = kb; jb < ib; ++jb)
#pragma unroll
          for (int j = 0; j < NB; ++j) v += A[ib * NB + r][jb * NB + j] * X[jb * NB + j][kb * NB + c];
        Tm[wave][r][c] = v;
      }
      chol_wave_sync();
      for (int e = lane; e < NB * NB; e += 64) {
        const int r = e / NB, c = e % NB;
        double v = 0.0;
        for (int j = 0; j <= r; ++j) v += X[ib * NB + r][ib * NB + j] * Tm[wave][j][c];
        X[ib * NB + r][kb * NB + c] = -v;
      }
      chol_wave_sync();
    }
  }
  __syncthreads();
  if (t == 0 && s_bad) *flag = 1;
  // M = R^-1 = (L^-1)^T: M[r][c] = X[c][r] for r <= c < w
  for (int e = t; e < B * B; e += kCholT) {
    const int r = e / B, c = e % B;
    const bool in = (r < w) && (c < w) && (r <= c);
    M[e] = in ? (float)X[c][r] : 0.f;
  }
}

extern "C" {

size_t mu_gram_worksize(int64_t n_rows, int B) {
  return (size_t)(gram_blocks(n_rows) + kGramFold) * (size_t)(B * B + B) * sizeof(double) + 256;
}

int mu_gram_cross_f32(int64_t n_rows, int B, const float* d_A, const float* d_Bm, double* d_C,
                      void* d_work, size_t work_bytes, void* stream) {
  MU_REQUIRE(B == 16 || B == 32 || B == 64, "B must be 16, 32 or 64");
  MU_REQUIRE(n_rows >= 0 && d_C, "bad arguments");
  MU_REQUIRE(n_rows == 0 || (d_A && d_Bm), "null input");
  MU_REQUIRE(d_work && work_bytes >= mu_gram_worksize(n_rows, B), "work buffer too small");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = gram_blocks(n_rows);
  double* partial = (double*)d_work;
  double* folded = partial + (size_t)blocks * (size_t)(B * B + B);
  const unsigned rblocks = (unsigned)((B * B + B + 255) / 256);
#define MU_CROSS(BB)                                                                                  \
  hipLaunchKernelGGL(k_gram_cross_partial<BB>, dim3(blocks), dim3(kGramThreads), 0, st, n_rows, d_A,  \
                     d_Bm, partial);                                                                  \
  MU_CHECK_LAUNCH();                                                                                  \
  if (blocks <= kGramFold) {                                                                          \
    hipLaunchKernelGGL(k_gram_cross_reduce<BB>, dim3(rblocks), dim3(256), 0, st, blocks, partial, d_C); \
  } else {                                                                                            \
    hipLaunchKernelGGL(k_gram_fold<BB>, dim3(rblocks, kGramFold), dim3(256), 0, st, blocks, partial,  \
                       folded);                                                                       \
    MU_CHECK_LAUNCH();                                                                                \
    hipLaunchKernelGGL(k_gram_cross_reduce<BB>, dim3(rblocks), dim3(256), 0, st, kGramFold, folded, d_C); \
  }
  switch (B) {
    case 64: MU_CROSS(64) break;
    case 32: MU_CROSS(32) break;
    default: MU_CROSS(16) break;
  }
#undef MU_CROSS
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_gram_f32(int64_t n_rows, int B, const float* d_A, double* d_G, double* d_colsum, void* d_work,
                size_t work_bytes, void* stream) {
  MU_REQUIRE(B == 16 || B == 32 || B == 64, "B must be 16, 32 or 64");
  MU_REQUIRE(n_rows >= 0 && d_G, "bad arguments");
  MU_REQUIRE(n_rows == 0 || d_A, "null input");
  MU_REQUIRE(d_work && work_bytes >= mu_gram_worksize(n_rows, B), "work buffer too small");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = gram_blocks(n_rows);
  double* partial = (double*)d_work;
  double* folded = partial + (size_t)blocks * (size_t)(B * B + B);
  const unsigned rblocks = (unsigned)((B * B + B + 255) / 256);
  switch (B) {
    case 64:
      hipLaunchKernelGGL(k_gram_partial<64>, dim3(blocks), dim3(kGramThreads), 0, st, n_rows, d_A, partial);
      MU_CHECK_LAUNCH();
      if (blocks <= kGramFold) {
        hipLaunchKernelGGL(k_gram_reduce<64>, dim3(rblocks), dim3(256), 0, st, blocks, partial, d_G, d_colsum);
        break;
      }
      hipLaunchKernelGGL(k_gram_fold<64>, dim3(rblocks, kGramFold), dim3(256), 0, st, blocks, partial, folded);
      MU_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_gram_reduce<64>, dim3(rblocks), dim3(256), 0, st, kGramFold, folded, d_G, d_colsum);
      break;
    case 32:
      hipLaunchKernelGGL(k_gram_partial<32>, dim3(blocks), dim3(kGramThreads), 0, st, n_rows, d_A, partial);
      MU_CHECK_LAUNCH();
      if (blocks <= kGramFold) {
        hipLaunchKernelGGL(k_gram_reduce<32>, dim3(rblocks), dim3(256), 0, st, blocks, partial, d_G, d_colsum);
        break;
      }
      hipLaunchKernelGGL(k_gram_fold<32>, dim3(rblocks, kGramFold), dim3(256), 0, st, blocks, partial, folded);
      MU_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_gram_reduce<32>, dim3(rblocks), dim3(256), 0, st, kGramFold, folded, d_G, d_colsum);
      break;
    default:
      hipLaunchKernelGGL(k_gram_partial<16>, dim3(blocks), dim3(kGramThreads), 0, st, n_rows, d_A, partial);
      MU_CHECK_LAUNCH();
      if (blocks <= kGramFold) {
        hipLaunchKernelGGL(k_gram_reduce<16>, dim3(rblocks), dim3(256), 0, st, blocks, partial, d_G, d_colsum);
        break;
      }
      hipLaunchKernelGGL(k_gram_fold<16>, dim3(rblocks, kGramFold), dim3(256), 0, st, blocks, partial, folded);
      MU_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_gram_reduce<16>, dim3(rblocks), dim3(256), 0, st, kGramFold, folded, d_G, d_colsum);
      break;
  }
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_dense_apply_f32(int64_t n_rows, int B, const float* d_A, const float* d_M,
                       const float* d_bias, float* d_Out, void* stream) {
  MU_REQUIRE(B == 16 || B == 32 || B == 64, "B must be 16, 32 or 64");
  MU_REQUIRE(n_rows >= 0, "negative n_rows");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_A && d_M && d_Out, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (n_rows + 63) / 64;
  const int64_t cap = (int64_t)mu_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  switch (B) {
    case 64:
      hipLaunchKernelGGL(k_dense_apply<64>, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, d_A, d_M, d_bias, d_Out);
      break;
    case 32:
      hipLaunchKernelGGL(k_dense_apply<32>, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, d_A, d_M, d_bias, d_Out);
      break;
    default:
      hipLaunchKernelGGL(k_dense_apply<16>, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, d_A, d_M, d_bias, d_Out);
      break;
  }
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_dense_project_out_f32(int64_t n_rows, int B, const float* d_A, const double* d_C, float* d_Z,
                             void* stream) {
  MU_REQUIRE(B == 16 || B == 32 || B == 64, "B must be 16, 32 or 64");
  MU_REQUIRE(n_rows >= 0, "negative size");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_A && d_C && d_Z, "null pointer");
  MU_REQUIRE(d_A != d_Z, "the projected block must not alias the basis block");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (((n_rows + 15) / 16) + 3) / 4;
  const int64_t cap = (int64_t)mu_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  switch (B) {
    case 64:
      hipLaunchKernelGGL(k_dense_project<64>, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, d_A, d_C, d_Z);
      break;
    case 32:
      hipLaunchKernelGGL(k_dense_project<32>, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, d_A, d_C, d_Z);
      break;
    default:
      hipLaunchKernelGGL(k_dense_project<16>, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, d_A, d_C, d_Z);
      break;
  }
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_randn_f32(int64_t count, uint64_t seed, float* d_out, void* stream) {
  MU_REQUIRE(count >= 0, "negative count");
  if (count == 0) return MU_OK;
  MU_REQUIRE(d_out, "null pointer");
  int64_t blocks = (count + 255) / 256;
  const int64_t cap = (int64_t)mu_num_cus() * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_randn, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, count, seed, d_out);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_chol_rinv_f64(int B, int w, const double* d_G, float* d_M, int* d_flag, void* stream) {
  MU_REQUIRE(B >= 1 && B <= 64 && w >= 0 && w <= B, "B must be 1..64 and 0 <= w <= B");
  MU_REQUIRE(d_G && d_M && d_flag, "null pointer");
  hipLaunchKernelGGL(k_chol_rinv, dim3(1), dim3(kCholT), 0, (hipStream_t)stream, B, w, d_G, d_M, d_flag);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
