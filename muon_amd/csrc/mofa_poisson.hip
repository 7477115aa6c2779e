// MOFA+ with a poisson view, without anything of size N x D (r04).
//
// mofapy2 fits count data through pseudo-data (Seeger's bound; the Poisson node reached from
// /root/reference/muon/_core/tools.py:583-585, likelihoods guessed at :272-280): with zeta = <Z><W>^T, rate =
// softplus(zeta) and the per-feature bound kappa_d, every iteration needs
//     R      = kappa_d zeta - sigmoid(zeta) (1 - y / rate)          (precision x pseudo-data, N x D)
//     b      = R^T <Z>   (D x K: W update),      a = R <W>   (N x K: Z update)
//     L      = sum y ln(rate) - rate                                 (likelihood term of the ELBO)
// The reference densifies the view for it; r03's engine walked dense CHUNKS (densify, zeta by GEMM, an element-wise
// kernel, the reductions by GEMM: ~4.5 GB of traffic per chunk and pass, 5 ms per pass at 20 000 x 20 000).  But a count
// matrix is zeros except for its stored entries, and for y = 0 the element depends on (z_n, w_d) only:
//     R = [kappa_d zeta - sigmoid(zeta)]  +  [y > 0] sigmoid(zeta) y / rate
//     L = [-rate]                         +  [y > 0] y ln(rate)
// so a pass is a DENSE sweep over all (n, d) that reads nothing but the two K-column factor blocks - zeta in registers,
// the transform in registers, the reduction in registers - plus a SPARSE correction over the stored entries.
//
//   * dense sweep (`k_pois_dense`): a thread owns one row of the "own" block (a sample for a and L, a feature for b),
//     keeps it and its K accumulators in registers and walks the "other" block through LDS tiles of 128 rows that every
//     lane reads at the same address (broadcast).  ~2 KP + 6 vector instructions per (n, d): 0.3 ms per pass at
//     20 000 x 20 000, K = 10, against 5 ms.  The other block is cut into column blocks for parallelism; the partial
//     results [block][own][K] are added by the caller in block order (deterministic).
//   * sparse correction (`k_pois_sparse`): a wave per own row over its stored entries (CSR of the view for a and L, of
//     its transpose for b), the other block's rows gathered through the L2, a wave reduction per accumulator.
// Arithmetic in the storage type (f32 models in f32, f64 models in f64), like the chunk kernels it replaces.
#include "common.hpp"

namespace {

__device__ __forceinline__ float pz_exp(float x) { return __expf(x); }
__device__ __forceinline__ double pz_exp(double x) { return exp(x); }
__device__ __forceinline__ float pz_log(float x) { return __logf(x); }
__device__ __forceinline__ double pz_log(double x) { return log(x); }
__device__ __forceinline__ float pz_log1p(float x) { return log1pf(x); }
__device__ __forceinline__ double pz_log1p(double x) { return log1p(x); }
template <typename T> __device__ __forceinline__ T pz_tiny();
template <> __device__ __forceinline__ float pz_tiny<float>() { return 1e-30f; }
template <> __device__ __forceinline__ double pz_tiny<double>() { return 1e-300; }

template <typename T>
__device__ __forceinline__ T pz_softplus(T z) {  // as torch computes it (threshold 20), clamped away from zero
  T r = z > (T)20 ? z : pz_log1p(pz_exp(z));
  return r > pz_tiny<T>() ? r : pz_tiny<T>();
}
// The dense sweep of the likelihood term SUMS softplus over all (n, d): in f32 the hardware exp / log pair is enough
// there - softplus(z) = max(z, 0) + ln(1 + e^-|z|) with an absolute error of ~1e-7 per element, i.e. ~1e-12 of the sum
// after the f64 reduction over the samples - and a third of libm's log1pf (1.54 -> 0.6 ms per sweep at 20 000 x 20 000).
__device__ __forceinline__ float pz_softplus_sum(float z) {
  return fmaxf(z, 0.f) + __logf(1.0f + __expf(-fabsf(z)));
}
__device__ __forceinline__ double pz_softplus_sum(double z) { return pz_softplus(z); }
template <typename T>
__device__ __forceinline__ T pz_sigmoid(T z) { return (T)1 / ((T)1 + pz_exp(-z)); }

// a row of a factor block PADDED to KP columns (16-byte aligned rows: three 16-byte loads for K = 10 instead of ten
// dword gathers - the sparse corrections are bound by the number of gather instructions)
template <typename T, int KP>
__device__ __forceinline__ void pz_load_row(const T* __restrict__ p, T (&o)[KP]) {
  constexpr int V = 16 / (int)sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const vec_t* q = reinterpret_cast<const vec_t*>(p);
#pragma unroll
  for (int i = 0; i < KP / V; ++i) {
    const vec_t v = q[i];
#pragma unroll
    for (int u = 0; u < V; ++u) o[i * V + u] = v[u];
  }
}

constexpr int kPzTile = 128;     // rows of the other block per LDS tile
constexpr int kPzThreads = 256;

// MODE 0: own = samples, other = features, kappa per OTHER row:  out[own][k] = sum_d (kappa_d zeta - sigmoid zeta) w_dk
// MODE 1: own = features, other = samples, kappa per OWN row:    out[own][k] = sum_n (kappa_own zeta - sigmoid zeta) z_nk
// MODE 2: own = samples, other = features:                       out[own]    = sum_d -softplus(zeta)
// MODE 3 (r05): MODE 1 and the likelihood term in ONE sweep - the tau / ELBO pass of an iteration and the W update of the
//         next one evaluate the same predictions (same factors, same weights): out[own][0 .. K-1] as MODE 1,
//         out[own][K] = sum_n -softplus(zeta); rows of K + 1 values
template <typename T, int KP, int MODE>
__global__ __launch_bounds__(kPzThreads) void k_pois_dense(int64_t n_own, int64_t n_other, int K, int64_t other_block,
                                                           const T* __restrict__ E_own, const T* __restrict__ E_other,
                                                           const T* __restrict__ kappa, T* __restrict__ part) {
  __shared__ T tile[kPzTile][KP];
  __shared__ T kap[kPzTile];
  const int64_t own = (int64_t)blockIdx.x * kPzThreads + threadIdx.x;
  const int64_t o0 = (int64_t)blockIdx.y * other_block;
  const int64_t o1 = o0 + other_block < n_other ? o0 + other_block : n_other;
  T e[KP], acc[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) e[k] = acc[k] = (T)0;
  if (own < n_own) pz_load_row<T, KP>(E_own + own * KP, e);  // (padding columns are zero)
  const T kown = ((MODE == 1 || MODE == 3) && own < n_own) ? kappa[own] : (T)0;
  T lsum = (T)0;
  for (int64_t t0 = o0; t0 < o1; t0 += kPzTile) {
    const int rows = (int)(o1 - t0 < kPzTile ? o1 - t0 : kPzTile);
    __syncthreads();
    for (int i = threadIdx.x; i < kPzTile * KP; i += kPzThreads) {
      const int r = i / KP, k = i - r * KP;
      tile[r][k] = r < rows ? E_other[(t0 + r) * KP + k] : (T)0;
    }
    if (MODE == 0)
      for (int i = threadIdx.x; i < kPzTile; i += kPzThreads) kap[i] = i < rows ? kappa[t0 + i] : (T)0;
    __syncthreads();
    for (int j = 0; j < rows; ++j) {
      T o[KP];
#pragma unroll
      for (int k = 0; k < KP; ++k) o[k] = tile[j][k];  // (every lane reads the same address: a broadcast)
      T zeta = (T)0;
#pragma unroll
      for (int k = 0; k < KP; ++k) zeta += e[k] * o[k];
      if (MODE == 2 || MODE == 3) lsum -= pz_softplus_sum(zeta);
      if (MODE != 2) {
        const T r0 = (MODE == 0 ? kap[j] : kown) * zeta - pz_sigmoid(zeta);
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[k] += r0 * o[k];
      }
    }
  }
  if (own >= n_own) return;
  if (MODE == 2) {
    part[(int64_t)blockIdx.y * n_own + own] = lsum;
  } else {
    const int ostride = MODE == 3 ? K + 1 : K;
    T* out = part + ((int64_t)blockIdx.y * n_own + own) * ostride;
    for (int k = 0; k < K; ++k) out[k] = acc[k];
    if (MODE == 3) out[K] = lsum;
  }
}

// the stored entries: a wave per own row.  MODE as above; (indptr, indices, values) = CSR of the view (MODE 0, 2) or of
// its transpose (MODE 1); the result is ADDED to out (which holds the dense part).
template <typename T, int KP, int MODE>
__global__ __launch_bounds__(256) void k_pois_sparse(int64_t n_own, int K, const int64_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices, const T* __restrict__ values,
                                                     const T* __restrict__ E_own, const T* __restrict__ E_other,
                                                     T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t own = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (own >= n_own) return;
  T e[KP], acc[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) acc[k] = (T)0;
  pz_load_row<T, KP>(E_own + own * KP, e);
  T lsum = (T)0;
  const int64_t lo = indptr[own], hi = indptr[own + 1];
  for (int64_t p = lo + lane; p < hi; p += 64) {
    const int64_t j = indices[p];
    const T y = values[p];
    T o[KP];
    pz_load_row<T, KP>(E_other + j * KP, o);
    T zeta = (T)0;
#pragma unroll
    for (int k = 0; k < KP; ++k) zeta += e[k] * o[k];
    const T rate = pz_softplus(zeta);
    if (MODE == 2 || MODE == 3) lsum += y * pz_log(rate);
    if (MODE != 2) {
      const T c = pz_sigmoid(zeta) * y / rate;
#pragma unroll
      for (int k = 0; k < KP; ++k) acc[k] += c * o[k];
    }
  }
  if (MODE == 2) {
    lsum = wave_sum(lsum);
    if (lane == 0) out[own] += lsum;
  } else {
    const int ostride = MODE == 3 ? K + 1 : K;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      if (k < K) {  // (K is wave-uniform)
        const T s = wave_sum(acc[k]);
        if (lane == 0) out[own * ostride + k] += s;
      }
    }
    if (MODE == 3) {
      lsum = wave_sum(lsum);
      if (lane == 0) out[own * ostride + K] += lsum;
    }
  }
}

template <typename T, int KP>
int pois_dense_launch(int mode, int64_t n_own, int64_t n_other, int K, int64_t other_block, const void* E_own,
                      const void* E_other, const void* kappa, void* part, hipStream_t st) {
  const dim3 grid((unsigned)((n_own + kPzThreads - 1) / kPzThreads), (unsigned)((n_other + other_block - 1) / other_block));
#define MU_GO(MD)                                                                                                      \
  hipLaunchKernelGGL((k_pois_dense<T, KP, MD>), grid, dim3(kPzThreads), 0, st, n_own, n_other, K, other_block,          \
                     (const T*)E_own, (const T*)E_other, (const T*)kappa, (T*)part)
  if (mode == 0) MU_GO(0);
  else if (mode == 1) MU_GO(1);
  else if (mode == 2) MU_GO(2);
  else MU_GO(3);
#undef MU_GO
  MU_CHECK_LAUNCH();
  return MU_OK;
}

template <typename T, int KP>
int pois_sparse_launch(int mode, int64_t n_own, int K, const int64_t* indptr, const int32_t* indices, const void* values,
                       const void* E_own, const void* E_other, void* out, hipStream_t st) {
  const unsigned blocks = (unsigned)((n_own + 3) / 4);
#define MU_GO(MD)                                                                                                    \
  hipLaunchKernelGGL((k_pois_sparse<T, KP, MD>), dim3(blocks), dim3(256), 0, st, n_own, K, indptr, indices,          \
                     (const T*)values, (const T*)E_own, (const T*)E_other, (T*)out)
  if (mode == 0) MU_GO(0);
  else if (mode == 1) MU_GO(1);
  else if (mode == 2) MU_GO(2);
  else MU_GO(3);
#undef MU_GO
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // namespace

extern "C" {

int64_t mu_mofa_poisson_blocks(int64_t n_own, int64_t n_other) {
  // column blocks of the dense sweep: ~8 workgroups per CU in total, blocks of whole LDS tiles
  const int64_t own_wgs = (n_own + kPzThreads - 1) / kPzThreads;
  int64_t want = ((int64_t)mu_num_cus() * 8 + own_wgs - 1) / (own_wgs > 0 ? own_wgs : 1);
  const int64_t tiles = (n_other + kPzTile - 1) / kPzTile;
  if (want < 1) want = 1;
  if (want > tiles) want = tiles > 0 ? tiles : 1;
  const int64_t per = (tiles + want - 1) / want;  // tiles per block
  return per * kPzTile;                           // rows of the other block per column block
}

int mu_mofa_poisson_dense(int dtype, int mode, int64_t n_own, int64_t n_other, int K, int64_t other_block,
                          const void* d_E_own, const void* d_E_other, const void* d_kappa, void* d_part, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(mode >= 0 && mode <= 3 && K >= 1 && K <= 32, "mode 0..3, 1 <= K <= 32");
  MU_REQUIRE(n_own >= 0 && n_other >= 0 && other_block >= 1, "shape");
  if (n_own == 0 || n_other == 0) return MU_OK;
  MU_REQUIRE(d_E_own && d_E_other && d_part && (mode == 2 || d_kappa), "null pointer");
  hipStream_t st = (hipStream_t)stream;
#define MU_D(T_)                                                                                                   \
  (K <= 4    ? pois_dense_launch<T_, 4>(mode, n_own, n_other, K, other_block, d_E_own, d_E_other, d_kappa, d_part, st)  \
   : K <= 8  ? pois_dense_launch<T_, 8>(mode, n_own, n_other, K, other_block, d_E_own, d_E_other, d_kappa, d_part, st)  \
   : K <= 12 ? pois_dense_launch<T_, 12>(mode, n_own, n_other, K, other_block, d_E_own, d_E_other, d_kappa, d_part, st) \
   : K <= 16 ? pois_dense_launch<T_, 16>(mode, n_own, n_other, K, other_block, d_E_own, d_E_other, d_kappa, d_part, st) \
             : pois_dense_launch<T_, 32>(mode, n_own, n_other, K, other_block, d_E_own, d_E_other, d_kappa, d_part, st))
  return dtype == MU_DTYPE_F32 ? MU_D(float) : MU_D(double);
#undef MU_D
}

int mu_mofa_poisson_sparse(int dtype, int mode, int64_t n_own, int K, const int64_t* d_indptr, const int32_t* d_indices,
                           const void* d_values, const void* d_E_own, const void* d_E_other, void* d_out, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(mode >= 0 && mode <= 3 && K >= 1 && K <= 32, "mode 0..3, 1 <= K <= 32");
  if (n_own <= 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_E_own && d_E_other && d_out, "null pointer");
  hipStream_t st = (hipStream_t)stream;
#define MU_S(T_)                                                                                                       \
  (K <= 4    ? pois_sparse_launch<T_, 4>(mode, n_own, K, d_indptr, d_indices, d_values, d_E_own, d_E_other, d_out, st)  \
   : K <= 8  ? pois_sparse_launch<T_, 8>(mode, n_own, K, d_indptr, d_indices, d_values, d_E_own, d_E_other, d_out, st)  \
   : K <= 12 ? pois_sparse_launch<T_, 12>(mode, n_own, K, d_indptr, d_indices, d_values, d_E_own, d_E_other, d_out, st) \
   : K <= 16 ? pois_sparse_launch<T_, 16>(mode, n_own, K, d_indptr, d_indices, d_values, d_E_own, d_E_other, d_out, st) \
             : pois_sparse_launch<T_, 32>(mode, n_own, K, d_indptr, d_indices, d_values, d_E_own, d_E_other, d_out, st))
  return dtype == MU_DTYPE_F32 ? MU_S(float) : MU_S(double);
#undef MU_S
}

}  // extern "C"
