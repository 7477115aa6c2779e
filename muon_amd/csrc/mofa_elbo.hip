// MOFA+ : the updates of the small nodes (tau, alpha_w, theta, alpha_z) and the ELBO, fused.
// mofapy2 evaluates them node by node inside ent.run() (/root/reference/muon/_core/tools.py:585);
// as separate tensor operations they were ~250 launches of a few microseconds each per iteration
// (profiles/r02_c4_kernel_stats.md).  Every sum below is a column sum of a D x K or N x K array or
// a sum over the features of one view, so an iteration needs one pass per array and a one-workgroup
// finish per node.  Equations: SURVEY.md 8a row M3 and oracle/mofa_oracle.py; arithmetic in f64 for
// both storage types, partial sums folded in a fixed order (bit-reproducible).
#include "common.hpp"

namespace {

constexpr int kET = 256;        // threads per workgroup
constexpr int kEBlocksMax = 512;  // partials per pass

// digamma for x > 0: upward recurrence to x >= 10, then the asymptotic series (error < 1e-13)
__device__ double digamma_pos(double x) {
  double r = 0.0;
  for (int i = 0; i < 10 && x < 10.0; ++i) {  // (bounded: a NaN or a negative argument must not spin)
    r -= 1.0 / x;
    x += 1.0;
  }
  const double f = 1.0 / (x * x);
  return r + log(x) - 0.5 / x -
         f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132)))));
}

// E_q[ln p - ln q] of a Gamma node: prior (a0, b0), posterior (a, b), <x> = ex, <ln x> = elx
__device__ double gamma_kl(double a0, double b0, double a, double b, double ex, double elx) {
  const double lp = a0 * log(b0) - lgamma(a0) + (a0 - 1.0) * elx - b0 * ex;
  const double lq = a * log(b) - lgamma(a) + (a - 1.0) * elx - b * ex;
  return lp - lq;
}

// sum of v over the workgroup, result valid in thread 0; fixed tree
__device__ double block_sum(double v, double* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = kET / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// ---- tau node of one view + the likelihood term ------------------------------------------------
// one thread per (group, feature):
//   S = yy - 2 <w> . B + <w>^T Gz <w> + <w^2> . Z2 - <w>^2 . diag(Gz)      (expected squared residual)
//   a = a0 + N/2, b = b0 + S/2, tau = a/b, <ln tau> = psi(a) - ln b
template <typename T, int KP>
__global__ __launch_bounds__(kET) void k_mofa_tau(int64_t D, int K, int G, const T* __restrict__ yy,
                                                  const T* __restrict__ Ngm, const T* __restrict__ EW,
                                                  const T* __restrict__ EW2, const T* __restrict__ B,
                                                  const T* __restrict__ Gz, const T* __restrict__ Z2,
                                                  double a0, double b0, T* __restrict__ tau,
                                                  T* __restrict__ ltau, double* __restrict__ partial) {
  extern __shared__ char smem_raw[];
  double* sh = reinterpret_cast<double*>(smem_raw);  // kET
  double* sGz = sh + kET;                            // G*K*K
  double* sZ2 = sGz + G * K * K;                     // G*K
  for (int i = threadIdx.x; i < G * K * K; i += kET) sGz[i] = (double)Gz[i];
  for (int i = threadIdx.x; i < G * K; i += kET) sZ2[i] = (double)Z2[i];
  __syncthreads();
  double acc = 0.0;
  const int64_t total = (int64_t)G * D;
  for (int64_t id = (int64_t)blockIdx.x * kET + threadIdx.x; id < total; id += (int64_t)gridDim.x * kET) {
    const int g = (int)(id / D);
    const int64_t d = id - (int64_t)g * D;
    double w[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) w[k] = (k < K) ? (double)EW[d * K + k] : 0.0;
    const double* gz = sGz + g * K * K;
    double S = (double)yy[id];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      if (k < K) {
        double q = 0.0;
#pragma unroll
        for (int l = 0; l < KP; ++l)
          if (l < K) q += w[l] * gz[l * K + k];
        S += -2.0 * w[k] * (double)B[id * K + k] + q * w[k] + (double)EW2[d * K + k] * sZ2[g * K + k] -
             w[k] * w[k] * gz[k * K + k];
      }
    }
    const double n = (double)Ngm[g];
    const double a = a0 + 0.5 * n, b = b0 + 0.5 * S;
    const double t = a / b, lt = digamma_pos(a) - log(b);
    tau[id] = (T)t;
    ltau[id] = (T)lt;
    acc += 0.5 * n * (lt - 1.8378770664093453) - 0.5 * t * S + gamma_kl(a0, b0, a, b, t, lt);  // ln 2 pi
  }
  const double s = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// ---- the same node for a view whose statistics are per FEATURE (r06: a dense gaussian view with missing entries in
// the general engine, csrc-side of muon_amd/_core/mofa_general.py) -------------------------------------------------------
// With a mask the second moments of the factors do not factor out of the sum over the samples: Q[d] = sum_n M_nd <z z^T>_n
// is a K x K block per feature (one row for all features without a mask).  Expected squared residual of feature d:
//   S_d = yy_d - 2 <w_d> . B_d + sum_kl Q_d[k, l] <w w^T>_d[k, l],   <w w^T>[k, l] = <w_k><w_l> (k != l), <w_k^2> (k = l)
// As tensor operations this and the node's finish below were ~35 launches of 3-8 us per view and iteration.
template <typename T, int KP>
__global__ __launch_bounds__(kET) void k_mofa_stats_resid(int64_t D, int K, int64_t q_rows, const double* __restrict__ yy,
                                                          const T* __restrict__ EW, const T* __restrict__ EW2,
                                                          const T* __restrict__ B, const T* __restrict__ Q,
                                                          double* __restrict__ S) {
  const int64_t d = (int64_t)blockIdx.x * kET + threadIdx.x;
  if (d >= D) return;
  double w[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) w[k] = (k < K) ? (double)EW[d * K + k] : 0.0;
  const T* q = Q + (q_rows > 1 ? d : 0) * (int64_t)K * K;
  double s = yy[d];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    if (k < K) {
      double r = 0.0;
#pragma unroll
      for (int l = 0; l < KP; ++l)
        if (l < K) r += (double)q[k * K + l] * (l == k ? (double)EW2[d * K + k] : w[k] * w[l]);
      s += r - 2.0 * w[k] * (double)B[d * K + k];
    }
  }
  S[d] = s;
}

// a = a0 + N/2, b = b0 + S/2, tau = a/b, <ln tau> = psi(a) - ln b for n = G x D (group, feature) pairs with their own
// counts N (f64, summed over the ranks by the caller like S); the likelihood and tau-node terms into `partial`
template <typename T>
__global__ __launch_bounds__(kET) void k_mofa_tau_finish(int64_t n, const double* __restrict__ S,
                                                         const double* __restrict__ Ngd, double a0, double b0,
                                                         T* __restrict__ tau, T* __restrict__ ltau,
                                                         double* __restrict__ partial) {
  __shared__ double sh[kET];
  double acc = 0.0;
  for (int64_t id = (int64_t)blockIdx.x * kET + threadIdx.x; id < n; id += (int64_t)gridDim.x * kET) {
    const double cnt = Ngd[id], s = S[id];
    const double a = a0 + 0.5 * cnt, b = b0 + 0.5 * s;
    const double t = a / b, lt = digamma_pos(a) - log(b);
    tau[id] = (T)t;
    ltau[id] = (T)lt;
    acc += 0.5 * cnt * (lt - 1.8378770664093453) - 0.5 * t * s + gamma_kl(a0, b0, a, b, t, lt);  // ln 2 pi
  }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// *elbo += sum of the partials, in order
// (r05: 64 lanes take contiguous runs of the partials and the runs are added in lane order - one thread walking up to
//  512 partials was 27 us behind a 15 us pass)
__global__ __launch_bounds__(64) void k_mofa_add_partials(int nb, const double* __restrict__ partial,
                                                          double* __restrict__ elbo) {
  __shared__ double run[64];
  const int per = (nb + 63) / 64;
  const int b0 = threadIdx.x * per, b1 = b0 + per < nb ? b0 + per : nb;
  double s = 0.0;
  for (int i = b0; i < b1; ++i) s += partial[i];
  run[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 64; ++i) t += run[i];
    *elbo += t;
  }
}

// ---- column sums of the weight node's arrays ---------------------------------------------------
// partial[block][q][k], q: 0 <w_hat^2>, 1 gamma, 2 gamma ln sig2, 3 entropy of Bernoulli(gamma)
// thread t: column t % KP of rows t / KP, t / KP + kET / KP, ... (consecutive threads = consecutive memory)
template <typename T, int KP>
__global__ __launch_bounds__(kET) void k_mofa_w_colsums(int64_t D, int K, const T* __restrict__ EWh2,
                                                        const T* __restrict__ gamma,
                                                        const T* __restrict__ sig2,
                                                        double* __restrict__ partial) {
  __shared__ double sh[kET][4];
  const int k = threadIdx.x % KP, r0 = threadIdx.x / KP;
  constexpr int kRows = kET / KP;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (k < K) {
    for (int64_t d = (int64_t)blockIdx.x * kRows + r0; d < D; d += (int64_t)gridDim.x * kRows) {
      const double gm = (double)gamma[d * K + k];
      s[0] += (double)EWh2[d * K + k];
      s[1] += gm;
      s[2] += gm * log((double)sig2[d * K + k]);
      double e = 0.0;  // -(x ln x + (1 - x) ln (1 - x)), 0 ln 0 = 0
      if (gm > 0.0 && gm < 1.0) e = -(gm * log(gm) + (1.0 - gm) * log1p(-gm));
      s[3] += e;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) sh[threadIdx.x][q] = s[q];
  __syncthreads();
  if ((int)threadIdx.x < 4 * KP) {
    const int q = threadIdx.x / KP, kk = threadIdx.x % KP;
    double t = 0.0;
    for (int r = 0; r < kRows; ++r) t += sh[r * KP + kk][q];
    if (kk < K) partial[((int64_t)blockIdx.x * 4 + q) * K + kk] = t;
  }
}

// alpha_w / theta updates from the column sums and the ELBO terms of the W, alpha_w and theta nodes
constexpr int kFinT = 1024;  // (quantity, factor) values x chunks of partial blocks: 64 x 16 for K <= 16, 128 x 8 beyond
template <typename T>
__global__ __launch_bounds__(kFinT) void k_mofa_w_finish(int64_t D, int K, int nb, int ard, int spikeslab,
                                                         const double* __restrict__ partial, double a_alpha,
                                                         double a0, double b0, double th_a0, double th_b0,
                                                         T* __restrict__ alpha, T* __restrict__ lalpha,
                                                         T* __restrict__ lth, T* __restrict__ l1mth,
                                                         double* __restrict__ elbo) {
  __shared__ double term[64];
  __shared__ double red[kFinT];
  __shared__ double cs[4][32];
  {
    // the partial blocks are summed in eight chunks side by side, the chunks in order (fixed order; r02
    // walked all of them on one thread per factor: 31 us behind a 15 us pass)
    const int kFinChunks = K <= 16 ? 16 : 8;
    const int lanes = kFinT / kFinChunks;  // 64 / 128 >= 4 K
    const int c = threadIdx.x / lanes, v = threadIdx.x % lanes;
    const int b0_ = (int)((int64_t)nb * c / kFinChunks), b1_ = (int)((int64_t)nb * (c + 1) / kFinChunks);
    double sacc = 0.0;
    if (v < 4 * K)
      for (int b = b0_; b < b1_; ++b) sacc += partial[(int64_t)b * 4 * K + v];  // v = q * K + k
    red[threadIdx.x] = sacc;
    __syncthreads();
    if (c == 0 && v < 4 * K) {
      double tot = 0.0;
      for (int q = 0; q < kFinChunks; ++q) tot += red[q * lanes + v];
      cs[v / K][v % K] = tot;
    }
    __syncthreads();
  }
  const int k = threadIdx.x;
  double e = 0.0;
  if (k < K) {
    const double c[4] = {cs[0][k], cs[1][k], cs[2][k], cs[3][k]};
    const double Dd = (double)D;
    double aw = 1.0, law = 0.0;
    if (ard) {
      const double b = b0 + 0.5 * c[0];
      aw = a_alpha / b;
      law = digamma_pos(a_alpha) - log(b);
      alpha[k] = (T)aw;
      lalpha[k] = (T)law;
      e += gamma_kl(a0, b0, a_alpha, b, aw, law);
    }
    // sum_d (0.5 ln alpha - 0.5 alpha <w_hat^2>) and the entropy of the slab
    e += 0.5 * Dd * law - 0.5 * aw * c[0];
    e += 0.5 * c[2] + 0.5 * log(1.0 / aw) * (Dd - c[1]) + 0.5 * Dd;
    if (spikeslab) {
      const double a = th_a0 + c[1], b = th_b0 + Dd - c[1];
      const double lt = digamma_pos(a) - digamma_pos(a + b), l1 = digamma_pos(b) - digamma_pos(a + b);
      lth[k] = (T)lt;
      l1mth[k] = (T)l1;
      e += c[1] * lt + (Dd - c[1]) * l1 + c[3];
      const double lb = lgamma(a) + lgamma(b) - lgamma(a + b);
      const double lb0 = lgamma(th_a0) + lgamma(th_b0) - lgamma(th_a0 + th_b0);
      e += (lb - lb0) + (th_a0 - a) * lt + (th_b0 - b) * l1;
    }
  }
  if (threadIdx.x < 64) term[threadIdx.x] = e;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < K; ++i) s += term[i];
    *elbo += s;
  }
}

// ---- factor node: column sums of <z^2> and ln sig2 over the rows [n0, n1) of one group ------------
template <typename T, int KP>
__global__ __launch_bounds__(kET) void k_mofa_z_colsums(int64_t n0, int64_t n1, int K,
                                                        const T* __restrict__ EZ2,
                                                        const T* __restrict__ sig2,
                                                        double* __restrict__ partial) {
  __shared__ double sh[kET][2];
  const int k = threadIdx.x % KP, r0 = threadIdx.x / KP;
  constexpr int kRows = kET / KP;
  double s0 = 0.0, s1 = 0.0;
  if (k < K) {
    for (int64_t n = n0 + (int64_t)blockIdx.x * kRows + r0; n < n1; n += (int64_t)gridDim.x * kRows) {
      s0 += (double)EZ2[n * K + k];
      s1 += log((double)sig2[n * K + k]);
    }
  }
  sh[threadIdx.x][0] = s0;
  sh[threadIdx.x][1] = s1;
  __syncthreads();
  if ((int)threadIdx.x < 2 * KP) {
    const int q = threadIdx.x / KP, kk = threadIdx.x % KP;
    double t = 0.0;
    for (int r = 0; r < kRows; ++r) t += sh[r * KP + kk][q];
    if (kk < K) partial[((int64_t)blockIdx.x * 2 + q) * K + kk] = t;
  }
}

// out[q][k] = sum over the partial blocks in a fixed order: eight chunks of blocks side by side, then the
// chunks (the caller all-reduces the result over the ranks); width <= 64
__global__ __launch_bounds__(512) void k_mofa_fold(int nb, int width, const double* __restrict__ partial,
                                                   double* __restrict__ out) {
  __shared__ double red[512];
  const int c = threadIdx.x / 64, i = threadIdx.x % 64;
  const int b0 = (int)((int64_t)nb * c / 8), b1 = (int)((int64_t)nb * (c + 1) / 8);
  double s = 0.0;
  if (i < width)
    for (int b = b0; b < b1; ++b) s += partial[(int64_t)b * width + i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (c == 0 && i < width) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q * 64 + i];
    out[i] = t;
  }
}

// alpha_z update and the ELBO terms of the Z and alpha_z nodes from zs[G][2][K] (global sums) and Ng[G]
template <typename T>
__global__ __launch_bounds__(64) void k_mofa_z_finish(int K, int G, int ard, const double* __restrict__ zs,
                                                      const double* __restrict__ Ng, double a0, double b0,
                                                      T* __restrict__ alpha_z, T* __restrict__ lalpha_z,
                                                      double* __restrict__ elbo) {
  __shared__ double term[64];
  const int k = threadIdx.x;
  double e = 0.0;
  if (k < K) {
    for (int g = 0; g < G; ++g) {
      const double z2 = zs[(g * 2 + 0) * K + k], lz = zs[(g * 2 + 1) * K + k], n = Ng[g];
      double az = 1.0, laz = 0.0;
      if (ard) {
        const double a = a0 + 0.5 * n, b = b0 + 0.5 * z2;
        az = a / b;
        laz = digamma_pos(a) - log(b);
        alpha_z[g * K + k] = (T)az;
        lalpha_z[g * K + k] = (T)laz;
        e += gamma_kl(a0, b0, a, b, az, laz);
      }
      e += 0.5 * laz * n - 0.5 * az * z2 + 0.5 * lz + 0.5 * n;
    }
  }
  term[threadIdx.x] = e;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < K; ++i) s += term[i];
    *elbo += s;
  }
}

inline int blocks_for(int64_t items, int per_block, int cap = kEBlocksMax) {
  int64_t b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}
// the finish kernels add the partials up one after the other (fixed order): few, fat workgroups for
// the column sums (512 partials made the one-workgroup finish 0.15 ms, ten times the pass itself)
constexpr int kEColBlocks = 256;

template <typename T>
int run_tau(int64_t D, int K, int G, const void* yy, const void* Ngm, const void* EW, const void* EW2,
            const void* B, const void* Gz, const void* Z2, double a0, double b0, void* tau, void* ltau,
            double* elbo, double* work, hipStream_t st) {
  const int nb = blocks_for((int64_t)G * D, kET);
  const size_t sh = (size_t)(kET + G * K * K + G * K) * sizeof(double);
#define ARGS D, K, G, (const T*)yy, (const T*)Ngm, (const T*)EW, (const T*)EW2, (const T*)B, \
             (const T*)Gz, (const T*)Z2, a0, b0, (T*)tau, (T*)ltau, work
  if (K <= 16) hipLaunchKernelGGL((k_mofa_tau<T, 16>), dim3(nb), dim3(kET), sh, st, ARGS);
  else hipLaunchKernelGGL((k_mofa_tau<T, 32>), dim3(nb), dim3(kET), sh, st, ARGS);
#undef ARGS
  MU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_mofa_add_partials, dim3(1), dim3(64), 0, st, nb, work, elbo);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

template <typename T>
int run_w(int64_t D, int K, int ard, int spikeslab, const void* EWh2, const void* gamma, const void* sig2,
          double a_alpha, double a0, double b0, double th_a0, double th_b0, void* alpha, void* lalpha,
          void* lth, void* l1mth, double* elbo, double* work, hipStream_t st) {
  const int KP = K <= 16 ? 16 : 32;
  const int nb = blocks_for(D, kET / KP, kEColBlocks);
  if (K <= 16)
    hipLaunchKernelGGL((k_mofa_w_colsums<T, 16>), dim3(nb), dim3(kET), 0, st, D, K, (const T*)EWh2,
                       (const T*)gamma, (const T*)sig2, work);
  else
    hipLaunchKernelGGL((k_mofa_w_colsums<T, 32>), dim3(nb), dim3(kET), 0, st, D, K, (const T*)EWh2,
                       (const T*)gamma, (const T*)sig2, work);
  MU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_mofa_w_finish<T>, dim3(1), dim3(kFinT), 0, st, D, K, nb, ard, spikeslab, work, a_alpha,
                     a0, b0, th_a0, th_b0, (T*)alpha, (T*)lalpha, (T*)lth, (T*)l1mth, elbo);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

template <typename T>
int run_z_sums(int64_t n0, int64_t n1, int K, const void* EZ2, const void* sig2, double* out, double* work,
               hipStream_t st) {
  const int KP = K <= 16 ? 16 : 32;
  const int nb = blocks_for(n1 - n0, kET / KP, kEColBlocks);
  if (K <= 16)
    hipLaunchKernelGGL((k_mofa_z_colsums<T, 16>), dim3(nb), dim3(kET), 0, st, n0, n1, K, (const T*)EZ2,
                       (const T*)sig2, work);
  else
    hipLaunchKernelGGL((k_mofa_z_colsums<T, 32>), dim3(nb), dim3(kET), 0, st, n0, n1, K, (const T*)EZ2,
                       (const T*)sig2, work);
  MU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_mofa_fold, dim3(1), dim3(512), 0, st, nb, 2 * K, work, out);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // namespace

extern "C" {

size_t mu_mofa_elbo_work_doubles(int K) { return (size_t)kEBlocksMax * 4 * (size_t)(K > 0 ? K : 1); }

int mu_mofa_tau_elbo(int dtype, int64_t D, int K, int G, const void* d_yy, const void* d_Ngm,
                     const void* d_EW, const void* d_EW2, const void* d_B, const void* d_Gz,
                     const void* d_Z2, double a0, double b0, void* d_tau, void* d_ltau, double* d_elbo,
                     double* d_work, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 32, "1 <= n_factors <= 32");
  MU_REQUIRE(G >= 1 && (size_t)(G * K * K + G * K) * 8 <= 60000, "too many groups for one LDS tile");
  if (D == 0) return MU_OK;
  MU_REQUIRE(d_yy && d_Ngm && d_EW && d_EW2 && d_B && d_Gz && d_Z2 && d_tau && d_ltau && d_elbo && d_work,
             "null pointer");
  if (dtype == MU_DTYPE_F32)
    return run_tau<float>(D, K, G, d_yy, d_Ngm, d_EW, d_EW2, d_B, d_Gz, d_Z2, a0, b0, d_tau, d_ltau, d_elbo,
                          d_work, (hipStream_t)stream);
  return run_tau<double>(D, K, G, d_yy, d_Ngm, d_EW, d_EW2, d_B, d_Gz, d_Z2, a0, b0, d_tau, d_ltau, d_elbo,
                         d_work, (hipStream_t)stream);
}

int mu_mofa_stats_resid(int dtype, int64_t D, int K, int64_t q_rows, const double* d_yy, const void* d_EW,
                        const void* d_EW2, const void* d_B, const void* d_Q, double* d_S, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 32, "1 <= n_factors <= 32");
  MU_REQUIRE(D >= 0 && (q_rows == 1 || q_rows == D), "Q has one row or one per feature");
  if (D == 0) return MU_OK;
  MU_REQUIRE(d_yy && d_EW && d_EW2 && d_B && d_Q && d_S, "null pointer");
  const unsigned nb = (unsigned)((D + kET - 1) / kET);
  hipStream_t st = (hipStream_t)stream;
#define MU_R(T_, KP_)                                                                                              \
  hipLaunchKernelGGL((k_mofa_stats_resid<T_, KP_>), dim3(nb), dim3(kET), 0, st, D, K, q_rows, d_yy, (const T_*)d_EW, \
                     (const T_*)d_EW2, (const T_*)d_B, (const T_*)d_Q, d_S)
  if (dtype == MU_DTYPE_F32) {
    if (K <= 16) MU_R(float, 16);
    else MU_R(float, 32);
  } else {
    if (K <= 16) MU_R(double, 16);
    else MU_R(double, 32);
  }
#undef MU_R
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_mofa_tau_finish(int dtype, int64_t n, const double* d_S, const double* d_Ngd, double a0, double b0, void* d_tau,
                       void* d_ltau, double* d_elbo, double* d_work, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(n >= 0, "negative size");
  if (n == 0) return MU_OK;
  MU_REQUIRE(d_S && d_Ngd && d_tau && d_ltau && d_elbo && d_work, "null pointer");
  const int nb = blocks_for(n, kET);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL((k_mofa_tau_finish<float>), dim3(nb), dim3(kET), 0, st, n, d_S, d_Ngd, a0, b0, (float*)d_tau,
                       (float*)d_ltau, d_work);
  else
    hipLaunchKernelGGL((k_mofa_tau_finish<double>), dim3(nb), dim3(kET), 0, st, n, d_S, d_Ngd, a0, b0, (double*)d_tau,
                       (double*)d_ltau, d_work);
  MU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_mofa_add_partials, dim3(1), dim3(64), 0, st, nb, d_work, d_elbo);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_mofa_w_elbo(int dtype, int64_t D, int K, int ard, int spikeslab, const void* d_EWh2,
                   const void* d_gamma, const void* d_sig2, double a_alpha, double a0, double b0,
                   double th_a0, double th_b0, void* d_alpha, void* d_lalpha, void* d_lth, void* d_l1mth,
                   double* d_elbo, double* d_work, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 32, "1 <= n_factors <= 32");
  MU_REQUIRE(D >= 0, "negative size");
  MU_REQUIRE(d_EWh2 && d_gamma && d_sig2 && d_alpha && d_lalpha && d_lth && d_l1mth && d_elbo && d_work,
             "null pointer");
  if (dtype == MU_DTYPE_F32)
    return run_w<float>(D, K, ard, spikeslab, d_EWh2, d_gamma, d_sig2, a_alpha, a0, b0, th_a0, th_b0, d_alpha,
                        d_lalpha, d_lth, d_l1mth, d_elbo, d_work, (hipStream_t)stream);
  return run_w<double>(D, K, ard, spikeslab, d_EWh2, d_gamma, d_sig2, a_alpha, a0, b0, th_a0, th_b0, d_alpha,
                       d_lalpha, d_lth, d_l1mth, d_elbo, d_work, (hipStream_t)stream);
}

int mu_mofa_z_sums(int dtype, int64_t n0, int64_t n1, int K, const void* d_EZ2, const void* d_sig2,
                   double* d_out, double* d_work, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 32, "1 <= n_factors <= 32");
  MU_REQUIRE(n0 >= 0 && n1 >= n0, "bad row range");
  MU_REQUIRE(d_EZ2 && d_sig2 && d_out && d_work, "null pointer");
  if (dtype == MU_DTYPE_F32)
    return run_z_sums<float>(n0, n1, K, d_EZ2, d_sig2, d_out, d_work, (hipStream_t)stream);
  return run_z_sums<double>(n0, n1, K, d_EZ2, d_sig2, d_out, d_work, (hipStream_t)stream);
}

int mu_mofa_z_elbo(int dtype, int K, int G, int ard, const double* d_zs, const double* d_Ng, double a0,
                   double b0, void* d_alpha_z, void* d_lalpha_z, double* d_elbo, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 32 && G >= 1, "1 <= n_factors <= 32, at least one group");
  MU_REQUIRE(d_zs && d_Ng && d_alpha_z && d_lalpha_z && d_elbo, "null pointer");
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_mofa_z_finish<float>, dim3(1), dim3(64), 0, (hipStream_t)stream, K, G, ard, d_zs,
                       d_Ng, a0, b0, (float*)d_alpha_z, (float*)d_lalpha_z, d_elbo);
  else
    hipLaunchKernelGGL(k_mofa_z_finish<double>, dim3(1), dim3(64), 0, (hipStream_t)stream, K, G, ard, d_zs,
                       d_Ng, a0, b0, (double*)d_alpha_z, (double*)d_lalpha_z, d_elbo);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
