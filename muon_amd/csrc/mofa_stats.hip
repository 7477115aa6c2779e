// MOFA+: the K x K statistics of a factor / weight block in ONE pass over the block (r03).
//
// Between the two passes over the data an iteration needs, per view and group,
//   W side:  TW = tau_g o <W> (the dense operand of A = Y (tau o W)),  Gw = <W>^T diag(tau_g) <W>,
//            dw2 = sum_d tau_gd <w_dk^2>,  corr = mu_g^T (tau_g o <W>) (implicit centring of a sparse view)
//   Z side:  Zst = <Z_g> (the dense operand of B = Y^T Z),  Gz = <Z_g>^T diag(pres) <Z_g>,
//            Z2 = sum_n pres <z_nk^2>,  Zs = sum_n pres <z_nk>
// r02 assembled them from ~110 small tensor operations per iteration (zero fills, paddings, slices,
// element-wise products, reductions, two tall-skinny Gram launches each: ~0.9 of 6.8 ms on
// BASELINE configs[3]).  Both sides are the same computation on an R x K block E with second moments
// E2, a row weight w and an auxiliary row weight a:
//   pad[r, col0 + k] = (scale_out ? w_r : 1) E[r, k]        (and its transpose, optionally)
//   gram[i, j] = sum_r w_r E[r, i] E[r, j],  s2[k] = sum_r w_r E2[r, k],  s1[k] = sum_r a_r w_r E[r, k]
// f64 accumulation, fixed-order two-level reduction (bit-reproducible).  These sit where mofapy2's node
// updates recompute the same moments (/root/reference/muon/_core/tools.py:585 -> ent.run()).
#include "common.hpp"

namespace {

constexpr int kST = 256;          // threads per block
constexpr int kSBlocksMax = 512;  // partial blocks.  r05: 96 -> 512 - a thread of the 96-block launch walked 65 row groups one
                                  // dependent load batch after the other (86 us for 4 MB of factors, three times per MOFA
                                  // iteration = 6 % of c4); the fold below takes the blocks in 32 chunks instead of 8

// thread t: column j = t % KP of rows t / KP, t / KP + kST / KP, ...
template <typename T, int KP>
__global__ __launch_bounds__(kST) void k_rowstats(int64_t r0, int64_t r1, int K, const T* __restrict__ E,
                                                  const T* __restrict__ E2, const T* __restrict__ wgt,
                                                  const T* __restrict__ aux, int scale_out,
                                                  T* __restrict__ out_pad, int ld, int col0,
                                                  T* __restrict__ out_t, int64_t ld_t,
                                                  double* __restrict__ partial) {
  constexpr int kRows = kST / KP;
  __shared__ double sh[kST];
  const int j = threadIdx.x % KP, rr = threadIdx.x / KP;
  double g[KP];
#pragma unroll
  for (int i = 0; i < KP; ++i) g[i] = 0.0;
  double s2 = 0.0, s1 = 0.0;
  if (j < K) {
    for (int64_t r = r0 + (int64_t)blockIdx.x * kRows + rr; r < r1; r += (int64_t)gridDim.x * kRows) {
      const T w = wgt ? wgt[r] : (T)1;
      const T a = aux ? aux[r] : (T)1;
      const T* e = E + r * K;
      const T ej = e[j];
      const double wej = (double)w * (double)ej;
#pragma unroll
      for (int i = 0; i < KP; ++i)
        if (i < K) g[i] += wej * (double)e[i];
      s2 += (double)w * (double)E2[r * K + j];
      s1 += (double)a * wej;
      const T o = scale_out ? (T)(w * ej) : ej;
      if (out_pad) out_pad[r * ld + col0 + j] = o;
      if (out_t) out_t[(int64_t)j * ld_t + r] = o;
    }
  }
  // block reduction over the kRows row groups, value by value (K + 2 values per column)
  const int width = K * K + 2 * K;
  double* dst = partial + (int64_t)blockIdx.x * width;
  for (int v = 0; v < KP + 2; ++v) {
    double x = 0.0;
    if (v < KP) {
#pragma unroll
      for (int i = 0; i < KP; ++i)
        if (i == v) x = g[i];
    } else {
      x = v == KP ? s2 : s1;
    }
    __syncthreads();
    sh[threadIdx.x] = x;
    __syncthreads();
    if (rr == 0 && j < K && (v >= KP || v < K)) {
      double s = 0.0;
      for (int q = 0; q < kRows; ++q) s += sh[q * KP + j];
      if (v < KP) dst[v * K + j] = s;              // gram[v][j]
      else dst[K * K + (v - KP) * K + j] = s;      // s2[j], s1[j]
    }
  }
}

// out = sum over the partial blocks in a fixed order: kFoldChunks chunks of blocks in parallel (the serial
// walk over all blocks took 94 us at 512 blocks: a third of what the fusion had saved), then the chunks
constexpr int kFoldChunks = 32, kFoldT = 1024;
template <typename T>
__global__ __launch_bounds__(kFoldT) void k_rowstats_fold(int nb, int K, const double* __restrict__ partial,
                                                          T* __restrict__ gram, T* __restrict__ s2,
                                                          T* __restrict__ s1) {
  __shared__ double sh[kFoldT];
  const int width = K * K + 2 * K;
  const int lanes = kFoldT / kFoldChunks;  // values handled per sweep
  const int c = threadIdx.x / lanes, t = threadIdx.x % lanes;
  const int b0 = (int)((int64_t)nb * c / kFoldChunks), b1 = (int)((int64_t)nb * (c + 1) / kFoldChunks);
  for (int i0 = 0; i0 < width; i0 += lanes) {
    const int i = i0 + t;
    double s = 0.0;
    if (i < width)
      for (int b = b0; b < b1; ++b) s += partial[(int64_t)b * width + i];
    __syncthreads();
    sh[threadIdx.x] = s;
    __syncthreads();
    if (c == 0 && i < width) {
      double tot = 0.0;
#pragma unroll
      for (int q = 0; q < kFoldChunks; ++q) tot += sh[q * lanes + t];
      if (i < K * K) { if (gram) gram[i] = (T)tot; }
      else if (i < K * K + K) { if (s2) s2[i - K * K] = (T)tot; }
      else if (s1) s1[i - K * K - K] = (T)tot;
    }
  }
}

template <typename T>
int run(int64_t r0, int64_t r1, int K, const void* E, const void* E2, const void* wgt, const void* aux,
        int scale_out, void* out_pad, int ld, int col0, void* out_t, int64_t ld_t, void* gram, void* s2,
        void* s1, double* work, hipStream_t st) {
  const int KP = K <= 16 ? 16 : 32;
  const int64_t rows = r1 - r0, per = kST / KP;
  int nb = (int)((rows + per * 8 - 1) / (per * 8));  // >= 8 rows per thread group
  if (nb < 1) nb = 1;
  if (nb > kSBlocksMax) nb = kSBlocksMax;
#define ARGS r0, r1, K, (const T*)E, (const T*)E2, (const T*)wgt, (const T*)aux, scale_out, (T*)out_pad, ld, \
             col0, (T*)out_t, ld_t, work
  if (K <= 16) hipLaunchKernelGGL((k_rowstats<T, 16>), dim3(nb), dim3(kST), 0, st, ARGS);
  else hipLaunchKernelGGL((k_rowstats<T, 32>), dim3(nb), dim3(kST), 0, st, ARGS);
#undef ARGS
  MU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_rowstats_fold<T>, dim3(1), dim3(kFoldT), 0, st, nb, K, work, (T*)gram, (T*)s2, (T*)s1);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // namespace

extern "C" {

size_t mu_mofa_rowstats_work_doubles(int K) {
  const int k = K > 0 ? K : 1;
  return (size_t)kSBlocksMax * (size_t)(k * k + 2 * k);
}

int mu_mofa_rowstats(int dtype, int64_t r0, int64_t r1, int K, const void* d_E, const void* d_E2,
                     const void* d_wgt, const void* d_aux, int scale_out, void* d_out_pad, int ld, int col0,
                     void* d_out_t, int64_t ld_t, void* d_gram, void* d_s2, void* d_s1, double* d_work,
                     void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 32, "1 <= n_factors <= 32");
  MU_REQUIRE(r0 >= 0 && r1 >= r0, "bad row range");
  MU_REQUIRE(d_E && d_E2 && d_work, "null pointer");
  MU_REQUIRE(!d_out_pad || (ld >= col0 + K && col0 >= 0), "padded block too narrow");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MU_DTYPE_F32)
    return run<float>(r0, r1, K, d_E, d_E2, d_wgt, d_aux, scale_out, d_out_pad, ld, col0, d_out_t, ld_t,
                      d_gram, d_s2, d_s1, d_work, st);
  return run<double>(r0, r1, K, d_E, d_E2, d_wgt, d_aux, scale_out, d_out_pad, ld, col0, d_out_t, ld_t,
                     d_gram, d_s2, d_s1, d_work, st);
}

}  // extern "C"

// ---- Gauss-Seidel sweep over the K factors of every row with ROW-WISE K x K statistics (r03) --------------
// The nodes of the element-wise-precision model (non-gaussian likelihoods, NaN entries: muon_amd/_core/
// mofa_general.py; mofapy2's W / Z nodes reached from /root/reference/muon/_core/tools.py:585) update factor k of
// row r from  t = b[r][k] - sum_j E[r][j] T[r][k][j] + E[r][k] T[r][k][k],  prec = T[r][k][k] + a_k  with the
// row's OWN K x K matrix T[r] (sum_n Omega_nd <z z^T> for a weight row, sum_d Omega_nd <w w^T> for a sample),
// one factor after the other with the fresh values of the earlier ones.  As tensor operations that was K x ~15
// launches over n x K slices per view; here a thread takes a row.  Arithmetic in f64 for both storage types.
namespace {

template <typename T, int KP>
__global__ __launch_bounds__(256) void k_gs_update(int64_t n, int K, const T* __restrict__ Tm, const T* __restrict__ b,
                                                   const double* __restrict__ prior, const double* __restrict__ lth,
                                                   const double* __restrict__ l1mth, int spikeslab, T* __restrict__ E,
                                                   T* __restrict__ E2, T* __restrict__ gamma, T* __restrict__ Eh2,
                                                   T* __restrict__ sig2) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double e[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) e[k] = k < K ? (double)E[r * K + k] : 0.0;
  const T* Tr = Tm + r * (int64_t)K * K;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    if (k < K) {
    double t = (double)b[r * K + k];
    double tkk = 0.0;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      if (j < K) {
        const double v = (double)Tr[k * K + j];
        if (j == k) tkk = v;
        else t -= e[j] * v;
      }
    }
    const double a = prior[k];
    const double prec = tkk + a;
    const double s2 = 1.0 / prec;
    const double mu = t * s2;
    double gam = 1.0;
    if (spikeslab) {
      const double lam = lth[k] - l1mth[k] + 0.5 * log(a) - 0.5 * log(prec) + 0.5 * t * t * s2;
      gam = 1.0 / (1.0 + exp(-lam));
    }
    e[k] = gam * mu;
    const double m2 = gam * (mu * mu + s2);
    E[r * K + k] = (T)e[k];
    E2[r * K + k] = (T)m2;
    if (gamma) gamma[r * K + k] = (T)gam;
    if (Eh2) Eh2[r * K + k] = (T)(m2 + (1.0 - gam) / a);
    sig2[r * K + k] = (T)s2;
    }
  }
}

}  // namespace

extern "C" int mu_mofa_gs_update(int dtype, int64_t n, int K, const void* d_T, const void* d_b, const double* d_prior,
                                 const double* d_lth, const double* d_l1mth, int spikeslab, void* d_E, void* d_E2,
                                 void* d_gamma, void* d_Eh2, void* d_sig2, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 32 && n >= 0, "1 <= n_factors <= 32");
  if (n == 0) return MU_OK;
  MU_REQUIRE(d_T && d_b && d_prior && d_E && d_E2 && d_sig2, "null pointer");
  MU_REQUIRE(!spikeslab || (d_lth && d_l1mth), "spike-and-slab needs the theta expectations");
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
#define MU_GS(T_, KP_)                                                                                          \
  hipLaunchKernelGGL((k_gs_update<T_, KP_>), dim3(blocks), dim3(256), 0, st, n, K, (const T_*)d_T, (const T_*)d_b,  \
                     d_prior, d_lth, d_l1mth, spikeslab, (T_*)d_E, (T_*)d_E2, (T_*)d_gamma, (T_*)d_Eh2, (T_*)d_sig2)
  if (dtype == MU_DTYPE_F32) {
    if (K <= 16) MU_GS(float, 16); else MU_GS(float, 32);
  } else {
    if (K <= 16) MU_GS(double, 16); else MU_GS(double, 32);
  }
#undef MU_GS
  MU_CHECK_LAUNCH();
  return MU_OK;
}

// ---- poisson pseudo-data of a dense chunk, element-wise (r03) ---------------------------------------------------------
// mofapy2's Poisson pseudo-data node (Seeger bound; reached from tools.py:585) on a chunk of predictions zeta = <Z><W>^T:
//   rate = softplus(zeta);  mode 0:  R = kappa_d zeta - sigmoid(zeta) (1 - y / rate)   (precision x pseudo-data)
//                           mode 1:  R = y ln(rate) - rate                            (the likelihood term of the ELBO)
// As tensor operations each was eight passes over the N x D chunk.  Arithmetic in the storage type.
namespace {

__device__ __forceinline__ float pp_exp(float x) { return __expf(x); }
__device__ __forceinline__ double pp_exp(double x) { return exp(x); }
__device__ __forceinline__ float pp_log(float x) { return __logf(x); }
__device__ __forceinline__ double pp_log(double x) { return log(x); }
__device__ __forceinline__ float pp_log1p(float x) { return log1pf(x); }
__device__ __forceinline__ double pp_log1p(double x) { return log1p(x); }

// (arithmetic in the storage type, as the tensor operations it replaces: f64 transcendentals on an f32 model made
//  this kernel 41 % of an iteration - 1.05 ms per 1.3e8-element chunk)
template <typename T>
__global__ __launch_bounds__(256) void k_poisson_pseudo(int64_t n, int64_t D, int mode, const T* __restrict__ zeta,
                                                        const T* __restrict__ Y, const T* __restrict__ kappa,
                                                        T* __restrict__ out, T tiny) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const T z = zeta[i], y = Y[i];
    // softplus as torch computes it (threshold 20), clamped away from zero
    T rate = z > (T)20 ? z : pp_log1p(pp_exp(z));
    rate = rate > tiny ? rate : tiny;
    T r;
    if (mode == 0) {
      const T sg = (T)1 / ((T)1 + pp_exp(-z));
      r = kappa[i % D] * z - sg * ((T)1 - y / rate);
    } else {
      r = y * pp_log(rate) - rate;
    }
    out[i] = r;
  }
}

}  // namespace

extern "C" int mu_mofa_poisson_pseudo(int dtype, int64_t n_rows, int64_t D, int mode, const void* d_zeta, const void* d_Y,
                                      const void* d_kappa, void* d_out, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(n_rows >= 0 && D >= 1 && (mode == 0 || mode == 1), "shape / mode");
  const int64_t n = n_rows * D;
  if (n == 0) return MU_OK;
  MU_REQUIRE(d_zeta && d_Y && d_out && (mode == 1 || d_kappa), "null pointer");
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = (int64_t)mu_num_cus() * 32;
  if (blocks > cap) blocks = cap;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_poisson_pseudo<float>, dim3((unsigned)blocks), dim3(256), 0, st, n, D, mode, (const float*)d_zeta,
                       (const float*)d_Y, (const float*)d_kappa, (float*)d_out, 1e-30f);
  else
    hipLaunchKernelGGL(k_poisson_pseudo<double>, dim3((unsigned)blocks), dim3(256), 0, st, n, D, mode,
                       (const double*)d_zeta, (const double*)d_Y, (const double*)d_kappa, (double*)d_out, 1e-300);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

// ---- rows [r0, r1) of a CSR as a dense chunk (zeros are data for a count likelihood): one pass, a workgroup per row ----
namespace {

template <typename T>
__global__ __launch_bounds__(256) void k_densify_rows(int64_t r0, int64_t D, const int64_t* __restrict__ indptr,
                                                      const int32_t* __restrict__ indices, const T* __restrict__ values,
                                                      T* __restrict__ out) {
  const int64_t row = r0 + blockIdx.x;
  T* o = out + (int64_t)blockIdx.x * D;
  for (int64_t j = threadIdx.x; j < D; j += blockDim.x) o[j] = (T)0;
  __syncthreads();
  const int64_t a = indptr[row], b = indptr[row + 1];
  for (int64_t p = a + threadIdx.x; p < b; p += blockDim.x) o[indices[p]] = values[p];
}

}  // namespace

extern "C" int mu_csr_densify_rows(int dtype, int64_t r0, int64_t r1, int64_t D, const int64_t* d_indptr,
                                   const int32_t* d_indices, const void* d_values, void* d_out, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(r0 >= 0 && r1 >= r0 && D >= 1 && r1 - r0 < ((int64_t)1 << 31), "row range");
  if (r1 == r0) return MU_OK;
  MU_REQUIRE(d_indptr && d_out, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_densify_rows<float>, dim3((unsigned)(r1 - r0)), dim3(256), 0, st, r0, D, d_indptr, d_indices,
                       (const float*)d_values, (float*)d_out);
  else
    hipLaunchKernelGGL(k_densify_rows<double>, dim3((unsigned)(r1 - r0)), dim3(256), 0, st, r0, D, d_indptr, d_indices,
                       (const double*)d_values, (double*)d_out);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

// ---- bernoulli pseudo-data precision of a dense chunk (Jaakkola bound), element-wise (r03) -----------------------------
//   xi^2 = zeta^2 + a - b  (a = <Z^2> <W^2>^T, b = <Z>^2 (<W>^2)^T: the variance of the prediction),
//   Omega = 2 lambda(xi) = tanh(xi / 2) / (2 xi),  xi clamped to >= 1e-8     (mofapy2's Bernoulli node, tools.py:585)
// As tensor operations: twelve passes over the N x D chunk.  Arithmetic in the storage type.  out may alias zeta.
namespace {

__device__ __forceinline__ float jj_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ double jj_tanh(double x) { return tanh(x); }
__device__ __forceinline__ float jj_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double jj_sqrt(double x) { return sqrt(x); }

template <typename T>
__global__ __launch_bounds__(256) void k_jaakkola(int64_t n, const T* __restrict__ zeta, const T* __restrict__ a,
                                                  const T* __restrict__ b, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const T z = zeta[i];
    T xi2 = z * z + a[i] - b[i];
    xi2 = xi2 > (T)0 ? xi2 : (T)0;
    T x = jj_sqrt(xi2);
    x = x > (T)1e-8 ? x : (T)1e-8;
    out[i] = (T)2 * (jj_tanh((T)0.5 * x) / ((T)4 * x));
  }
}

}  // namespace

extern "C" int mu_mofa_jaakkola(int dtype, int64_t n, const void* d_zeta, const void* d_a, const void* d_b, void* d_out,
                                void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(n >= 0, "size");
  if (n == 0) return MU_OK;
  MU_REQUIRE(d_zeta && d_a && d_b && d_out, "null pointer");
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = (int64_t)mu_num_cus() * 32;
  if (blocks > cap) blocks = cap;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_jaakkola<float>, dim3((unsigned)blocks), dim3(256), 0, st, n, (const float*)d_zeta,
                       (const float*)d_a, (const float*)d_b, (float*)d_out);
  else
    hipLaunchKernelGGL(k_jaakkola<double>, dim3((unsigned)blocks), dim3(256), 0, st, n, (const double*)d_zeta,
                       (const double*)d_a, (const double*)d_b, (double*)d_out);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// per-feature first and second moments of a dense view's rows r0 .. r1-1 (the per-(group, feature) means and the yy
// term of a fit's set-up: mofapy2 centres every view per group before training, /root/reference/muon/_core/tools.py:
// 283-286 keeps those means as the intercepts).  One pass over the block, f64 sums whatever the storage type; a
// workgroup = 1024 (vector loads) / 256 columns x one chunk of rows, partial[chunk][0 / 1][column] folded by the caller
// in a fixed order.  r04 took these sums with tensor reductions: three passes and an 8 GB temporary for the squares in
// f32, 16384-row slabs widened to f64 for the f32-stored view of an f64 fit (18 ms at 100 000 x 20 000).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

template <typename T, int V>
__global__ __launch_bounds__(256) void k_col_moments(int64_t r0, int64_t r1, int64_t D, const T* __restrict__ Y,
                                                     int64_t rows_per_chunk, double* __restrict__ partial) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (c >= D) return;
  const int64_t a = r0 + (int64_t)blockIdx.y * rows_per_chunk;
  int64_t b = a + rows_per_chunk;
  if (b > r1) b = r1;
  double s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) s1[j] = s2[j] = 0.0;
  struct alignas(sizeof(T) * V) Vec { T v[V]; };
  const T* p = Y + a * D + c;
  int64_t r = a;
  for (; r + 4 <= b; r += 4, p += 4 * D) {
    Vec x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const Vec*>(p + u * D);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const double y = (double)x[u].v[j];
        s1[j] += y;
        s2[j] = __builtin_fma(y, y, s2[j]);
      }
  }
  for (; r < b; ++r, p += D) {
    const Vec x = *reinterpret_cast<const Vec*>(p);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const double y = (double)x.v[j];
      s1[j] += y;
      s2[j] = __builtin_fma(y, y, s2[j]);
    }
  }
  double* out = partial + (int64_t)blockIdx.y * 2 * D + c;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    out[j] = s1[j];
    out[D + j] = s2[j];
  }
}

}  // namespace

extern "C" int mu_dense_col_moments_chunks(int64_t n_rows, int64_t D) {
  if (n_rows <= 0 || D <= 0) return 0;
  const int64_t ctiles = (D + 1023) / 1024;
  int64_t chunks = (8 * (int64_t)mu_num_cus() + ctiles - 1) / ctiles;
  const int64_t most = (n_rows + 31) / 32;  // (at least 32 rows per chunk)
  if (chunks > most) chunks = most;
  if (chunks > 4096) chunks = 4096;
  return (int)(chunks < 1 ? 1 : chunks);
}

extern "C" int mu_dense_col_moments(int dtype, int64_t r0, int64_t r1, int64_t D, const void* d_Y, int chunks,
                                    double* d_partial, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(r0 >= 0 && r1 >= r0 && D >= 0 && chunks >= 1 && chunks <= 65535, "shape");
  if (D == 0) return MU_OK;
  MU_REQUIRE(d_Y && d_partial, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (r1 == r0) {
    MU_CHECK_HIP(hipMemsetAsync(d_partial, 0, sizeof(double) * 2 * (size_t)chunks * (size_t)D, st));
    return MU_OK;
  }
  const int64_t rpc = (r1 - r0 + chunks - 1) / chunks;
  const bool vec = dtype == MU_DTYPE_F32 && D % 4 == 0 && (reinterpret_cast<uintptr_t>(d_Y) & 15) == 0;
  const int64_t per_block = vec ? 1024 : 256;
  const dim3 grid((unsigned)((D + per_block - 1) / per_block), (unsigned)chunks);
  if (vec)
    hipLaunchKernelGGL((k_col_moments<float, 4>), grid, dim3(256), 0, st, r0, r1, D, (const float*)d_Y, rpc, d_partial);
  else if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL((k_col_moments<float, 1>), grid, dim3(256), 0, st, r0, r1, D, (const float*)d_Y, rpc, d_partial);
  else
    hipLaunchKernelGGL((k_col_moments<double, 1>), grid, dim3(256), 0, st, r0, r1, D, (const double*)d_Y, rpc, d_partial);
  MU_CHECK_LAUNCH();
  return MU_OK;
}
