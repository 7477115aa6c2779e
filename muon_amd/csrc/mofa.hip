// MOFA+ coordinate updates (Gaussian likelihood): the per-feature W sweep and the per-sample
// Z sweep.  These replace the node updates mofapy2 runs inside ent.run()
// (/root/reference/muon/_core/tools.py:585); equations: SURVEY.md §8a row M3 and
// oracle/mofa_oracle.py.  All data-dependent terms enter through the sufficient statistics
//   B_g = Y_g^T <Z_g> (D x K)  and  A = Y (tau o <W>) (N x K)
// which are produced by the SpMM kernel (sparse views) or a dense GEMM (dense views), so a
// sweep only touches D x K / N x K arrays: one thread per feature (sample), the Gauss-Seidel
// loop over the K factors unrolled in registers, the K x K Grams broadcast from LDS.
#include "common.hpp"

constexpr int kMofaThreads = 128;

template <typename T>
__device__ __forceinline__ T sigmoid_t(T x) {
  return (T)1 / ((T)1 + exp(-x));
}

// One thread per feature d of one view.  Arrays: B[G][D][K], tau[G][D], Gz[G][K][K] (means,
// full matrix incl. diagonal), Z2[G][K] = sum_n <z^2>, alpha[K], lth[K], l1mth[K].
// In/out EW[D][K]; out EW2, gamma, EWh2 (= <w_hat^2>), sig2 (posterior variance of the slab).
template <typename T, int KP>
__global__ __launch_bounds__(kMofaThreads) void k_mofa_update_w(
    int64_t D, int K, int G, const T* __restrict__ B, const T* __restrict__ tau,
    const T* __restrict__ Gz, const T* __restrict__ Z2, const T* __restrict__ alpha,
    const T* __restrict__ lth, const T* __restrict__ l1mth, int spikeslab, T* __restrict__ EW,
    T* __restrict__ EW2, T* __restrict__ gamma, T* __restrict__ EWh2, T* __restrict__ sig2) {
  extern __shared__ char smem_raw[];
  T* sGz = reinterpret_cast<T*>(smem_raw);  // G*K*K
  T* sZ2 = sGz + G * K * K;                  // G*K
  for (int i = threadIdx.x; i < G * K * K; i += kMofaThreads) sGz[i] = Gz[i];
  for (int i = threadIdx.x; i < G * K; i += kMofaThreads) sZ2[i] = Z2[i];
  __syncthreads();
  const int64_t d = (int64_t)blockIdx.x * kMofaThreads + threadIdx.x;
  if (d >= D) return;
  T w[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) w[k] = (k < K) ? EW[d * K + k] : (T)0;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    if (k < K) {
      T t = (T)0, q = (T)0;
      for (int g = 0; g < G; ++g) {
        const T tg = tau[(int64_t)g * D + d];
        const T* gz = sGz + (g * K + 0) * K;
        T cross = (T)0;
#pragma unroll
        for (int j = 0; j < KP; ++j)
          if (j < K && j != k) cross += w[j] * gz[j * K + k];
        t += tg * (B[((int64_t)g * D + d) * K + k] - cross);
        q += tg * sZ2[g * K + k];
      }
      const T a = alpha[k];
      const T prec = q + a;
      const T s2 = (T)1 / prec;
      const T mu = t * s2;
      T gam = (T)1;
      if (spikeslab) {
        const T lam = lth[k] - l1mth[k] + (T)0.5 * log(a) - (T)0.5 * log(prec) + (T)0.5 * t * t * s2;
        gam = sigmoid_t(lam);
      }
      w[k] = gam * mu;
      const T m2 = gam * (mu * mu + s2);
      EW[d * K + k] = w[k];
      EW2[d * K + k] = m2;
      gamma[d * K + k] = gam;
      EWh2[d * K + k] = m2 + ((T)1 - gam) / a;
      sig2[d * K + k] = s2;
    }
  }
}

// One thread per sample n.  A[M][N][K], pres[M][N] (1 = sample observed in view m), grp[N],
// Gw[M][G][K][K] = <W>^T diag(tau_g) <W>, dw2[M][G][K] = sum_d tau_gd <w_dk^2>, alphaz[G][K],
// corr[M][G][K] (nullable) = mu_g^T (tau_g o <W>): the implicit centring of a sparse view, taken off A.
template <typename T, int KP>
__global__ __launch_bounds__(kMofaThreads) void k_mofa_update_z(
    int64_t N, int K, int M, int G, const T* __restrict__ A, const T* __restrict__ pres,
    const int32_t* __restrict__ grp, const T* __restrict__ Gw, const T* __restrict__ dw2,
    const T* __restrict__ alphaz, const T* __restrict__ corr, T* __restrict__ EZ, T* __restrict__ EZ2,
    T* __restrict__ sig2) {
  extern __shared__ char smem_raw[];
  T* sGw = reinterpret_cast<T*>(smem_raw);  // M*G*K*K
  T* sdw = sGw + M * G * K * K;              // M*G*K
  T* sco = sdw + M * G * K;                  // M*G*K
  for (int i = threadIdx.x; i < M * G * K * K; i += kMofaThreads) sGw[i] = Gw[i];
  for (int i = threadIdx.x; i < M * G * K; i += kMofaThreads) {
    sdw[i] = dw2[i];
    sco[i] = corr ? corr[i] : (T)0;
  }
  __syncthreads();
  const int64_t n = (int64_t)blockIdx.x * kMofaThreads + threadIdx.x;
  if (n >= N) return;
  const int g = grp[n];
  T z[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) z[k] = (k < K) ? EZ[n * K + k] : (T)0;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    if (k < K) {
      T num = (T)0, prec = alphaz[g * K + k];
      for (int m = 0; m < M; ++m) {
        const T pm = pres[(int64_t)m * N + n];
        const T* gw = sGw + ((m * G + g) * K) * K;
        T cross = (T)0;
#pragma unroll
        for (int j = 0; j < KP; ++j)
          if (j < K && j != k) cross += z[j] * gw[j * K + k];
        num += pm * (A[((int64_t)m * N + n) * K + k] - sco[(m * G + g) * K + k] - cross);
        prec += pm * sdw[(m * G + g) * K + k];
      }
      const T s2 = (T)1 / prec;
      z[k] = num * s2;
      EZ[n * K + k] = z[k];
      EZ2[n * K + k] = z[k] * z[k] + s2;
      sig2[n * K + k] = s2;
    }
  }
}

template <typename T>
static int launch_w(int64_t D, int K, int G, const void* B, const void* tau, const void* Gz,
                    const void* Z2, const void* alpha, const void* lth, const void* l1mth,
                    int spikeslab, void* EW, void* EW2, void* gamma, void* EWh2, void* sig2,
                    hipStream_t st) {
  const unsigned blocks = (unsigned)((D + kMofaThreads - 1) / kMofaThreads);
  const size_t sh = (size_t)(G * K * K + G * K) * sizeof(T);
#define ARGS D, K, G, (const T*)B, (const T*)tau, (const T*)Gz, (const T*)Z2, (const T*)alpha, \
             (const T*)lth, (const T*)l1mth, spikeslab, (T*)EW, (T*)EW2, (T*)gamma, (T*)EWh2, (T*)sig2
  if (K <= 16) hipLaunchKernelGGL((k_mofa_update_w<T, 16>), dim3(blocks), dim3(kMofaThreads), sh, st, ARGS);
  else hipLaunchKernelGGL((k_mofa_update_w<T, 32>), dim3(blocks), dim3(kMofaThreads), sh, st, ARGS);
#undef ARGS
  MU_CHECK_LAUNCH();
  return MU_OK;
}

template <typename T>
static int launch_z(int64_t N, int K, int M, int G, const void* A, const void* pres,
                    const int32_t* grp, const void* Gw, const void* dw2, const void* alphaz,
                    const void* corr, void* EZ, void* EZ2, void* sig2, hipStream_t st) {
  const unsigned blocks = (unsigned)((N + kMofaThreads - 1) / kMofaThreads);
  const size_t sh = (size_t)(M * G * K * K + 2 * M * G * K) * sizeof(T);
#define ARGS N, K, M, G, (const T*)A, (const T*)pres, grp, (const T*)Gw, (const T*)dw2, \
             (const T*)alphaz, (const T*)corr, (T*)EZ, (T*)EZ2, (T*)sig2
  if (K <= 16) hipLaunchKernelGGL((k_mofa_update_z<T, 16>), dim3(blocks), dim3(kMofaThreads), sh, st, ARGS);
  else hipLaunchKernelGGL((k_mofa_update_z<T, 32>), dim3(blocks), dim3(kMofaThreads), sh, st, ARGS);
#undef ARGS
  MU_CHECK_LAUNCH();
  return MU_OK;
}

extern "C" {

int mu_mofa_update_w(int dtype, int64_t D, int K, int G, const void* d_B, const void* d_tau,
                     const void* d_Gz, const void* d_Z2, const void* d_alpha, const void* d_lth,
                     const void* d_l1mth, int spikeslab, void* d_EW, void* d_EW2, void* d_gamma,
                     void* d_EWh2, void* d_sig2, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 32, "1 <= n_factors <= 32");
  MU_REQUIRE(G >= 1 && (size_t)(G * K * K + G * K) * 8 <= 60000, "too many groups for one LDS tile");
  if (D == 0) return MU_OK;
  MU_REQUIRE(d_B && d_tau && d_Gz && d_Z2 && d_alpha && d_EW && d_EW2 && d_gamma && d_EWh2 && d_sig2,
             "null pointer");
  if (dtype == MU_DTYPE_F32)
    return launch_w<float>(D, K, G, d_B, d_tau, d_Gz, d_Z2, d_alpha, d_lth, d_l1mth, spikeslab, d_EW,
                           d_EW2, d_gamma, d_EWh2, d_sig2, (hipStream_t)stream);
  return launch_w<double>(D, K, G, d_B, d_tau, d_Gz, d_Z2, d_alpha, d_lth, d_l1mth, spikeslab, d_EW,
                          d_EW2, d_gamma, d_EWh2, d_sig2, (hipStream_t)stream);
}

int mu_mofa_update_z(int dtype, int64_t N, int K, int M, int G, const void* d_A, const void* d_pres,
                     const int32_t* d_grp, const void* d_Gw, const void* d_dw2, const void* d_alphaz,
                     const void* d_corr, void* d_EZ, void* d_EZ2, void* d_sig2, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 32, "1 <= n_factors <= 32");
  MU_REQUIRE(M >= 1 && G >= 1 && (size_t)(M * G * K * K + 2 * M * G * K) * 8 <= 60000,
             "too many views x groups for one LDS tile");
  if (N == 0) return MU_OK;
  MU_REQUIRE(d_A && d_pres && d_grp && d_Gw && d_dw2 && d_alphaz && d_EZ && d_EZ2 && d_sig2,
             "null pointer");
  if (dtype == MU_DTYPE_F32)
    return launch_z<float>(N, K, M, G, d_A, d_pres, d_grp, d_Gw, d_dw2, d_alphaz, d_corr, d_EZ, d_EZ2,
                           d_sig2, (hipStream_t)stream);
  return launch_z<double>(N, K, M, G, d_A, d_pres, d_grp, d_Gw, d_dw2, d_alphaz, d_corr, d_EZ, d_EZ2,
                          d_sig2, (hipStream_t)stream);
}

}  // extern "C"
