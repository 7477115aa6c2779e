// MOFA+ with a bernoulli view, without anything of size N x D (r06).
//
// mofapy2 fits binary data through the Jaakkola bound (the Bernoulli node reached from
// /root/reference/muon/_core/tools.py:583-585, likelihoods guessed at :272-280): with zeta = <z_n> . <w_d> and
// xi_nd^2 = E[(z_n . w_d)^2] = zeta^2 + sum_k (<z_k^2><w_k^2> - <z_k>^2 <w_k>^2), every iteration needs
//     Omega_nd = 2 lambda(xi_nd) = tanh(xi / 2) / (2 xi)        (precision of the pseudo-data, N x D)
//     R_nd     = Omega x pseudo-data = y_nd - 1/2                (does NOT depend on xi)
//     T_d = sum_n Omega_nd <z z^T>_n  (K x K per feature, W update),   S_n = sum_d Omega_nd <w w^T>_d  (Z update)
//     b = R^T <Z>,  a = R <W>,  L = sum y zeta - ln(1 + e^zeta)
// r03-r05 walked dense CHUNKS of the view for it (densify, three N x D x K products, an element-wise kernel, two
// N x D x K^2 products per pass).  But the data enter through R and the likelihood only:
//     b = Y^T <Z> - 1/2 sum_n <z_n>,   a = Y <W> - 1/2 sum_d <w_d>,   sum y zeta = sum_n <z_n> . (Y <W>)_n
// are sparse products the library has, sum ln(1 + e^zeta) is the poisson view's likelihood sweep (mofa_poisson.hip,
// mode 2) - and Omega depends on the two factor blocks alone.  What is new is ONE kernel:
//     out[own][c] = sum_other Omega(own, other) M_other[c]      c over the K (K + 1) / 2 distinct entries of <m m^T>
// (own = features, other = samples, M = <z z^T>: T;  own = samples, other = features, M = <w w^T>: S), a dense sweep over
// all (own, other) pairs that reads the K-column blocks and the packed moment block of the other side.
//
// `k_jaakkola_sweep` is the tile kernel of mofa_poisson.hip's k_pois_mfma with a wider second product:
//   1. zeta^T and q^T tiles [16 other x 16 own] on the matrix cores: zeta = E_other . E_own (KP / 4 instructions), and
//      q = <e^2>-terms as  E_own^2 . var_other + var_own . E2_other  (2 KP / 4): every term of q is >= 0, nothing cancels
//   2. Omega = tanh(xi / 2) / (2 xi), xi = sqrt(zeta^2 + q), in the accumulator registers (a series below xi = 0.2)
//   3. out tile [16 own x 16 c] += Omega^T M_other for each of the CT column tiles of the packed moments (4 CT
//      instructions): step 3's A operand is accumulator register s of the lane that computed it (the reduction index
//      walked as row(j, s), the map of the instruction's C/D layout), B reads row row(j, s) of the LDS tile.
// f32 models use v_mfma_f32_16x16x4_f32, f64 models v_mfma_f64_16x16x4_f64 (C/D rows (lane >> 4) + 4 r instead of
// 4 (lane >> 4) + r: the same kernel with another row map).  The other block is cut into column blocks for
// parallelism; the partial results [block][own][pc] are added by the caller in block order (deterministic).
#include "common.hpp"

namespace {

template <typename T> struct BjMap;
template <> struct BjMap<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ int row(int q, int r) { return 4 * q + r; }
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // 2 lambda(xi) from xi^2: hardware sqrt / exp2 / rcp (1 ulp each); the series where 1 - e^-xi would cancel
  static __device__ __forceinline__ float omega(float x2) {
    x2 = fmaxf(x2, 0.f);
    const float xi = __builtin_amdgcn_sqrtf(x2);
    const float t = __builtin_amdgcn_exp2f(xi * -1.4426950408889634f);
    const float big = (1.0f - t) * __builtin_amdgcn_rcpf((1.0f + t) * (2.0f * xi));
    const float small = 0.25f + x2 * (-1.0f / 48.0f + x2 * (1.0f / 480.0f - x2 * (17.0f / 80640.0f)));
    return x2 < 0.04f ? small : big;
  }
};
template <> struct BjMap<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ int row(int q, int r) { return q + 4 * r; }
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ double omega(double x2) {
    x2 = x2 > 0.0 ? x2 : 0.0;
    if (x2 < 1e-4)  // (xi < 0.01: the next term of the series is 4e-20)
      return 0.25 + x2 * (-1.0 / 48.0 + x2 * (1.0 / 480.0 - x2 * (17.0 / 80640.0)));
    const double xi = sqrt(x2);
    return tanh(0.5 * xi) / (2.0 * xi);
  }
};

constexpr int kBjThreads = 256;

template <typename T, int KP, int CT>
struct BjShape {
  static constexpr int OWN = (sizeof(T) == 4 && CT <= 5) ? 2 : 1;  // 16-row own tiles per wave
  static constexpr int LSE = 2 * KP + 1;                          // LDS row of the factor tile: E | E2
  static constexpr int LSM = 16 * CT + 4;                         // LDS row of the moment tile
  static constexpr int kRowBytes = (LSE + LSM) * (int)sizeof(T);
  // rows of the other block per LDS stage: ~48 KB, a power of two in 32 .. 128
  static constexpr int SR = 49152 / kRowBytes >= 128 ? 128 : (49152 / kRowBytes >= 64 ? 64 : 32);
};

template <typename T, int KP, int CT>
__global__ __launch_bounds__(kBjThreads) void k_jaakkola_sweep(int64_t n_own, int64_t n_other, int K, int pc,
                                                               int64_t other_block, const T* __restrict__ E_own,
                                                               const T* __restrict__ E2_own,
                                                               const T* __restrict__ E_other,
                                                               const T* __restrict__ E2_other,
                                                               const T* __restrict__ M_other, int ldm,
                                                               T* __restrict__ part) {
  typedef BjShape<T, KP, CT> Sh;
  typedef BjMap<T> Mp;
  typedef typename Mp::acc_t acc_t;
  constexpr int KS = KP / 4, OWN = Sh::OWN, LSE = Sh::LSE, LSM = Sh::LSM, SR = Sh::SR;
  __shared__ T tile_e[SR * LSE];
  __shared__ T tile_m[SR * LSM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int64_t own0 = (int64_t)blockIdx.x * (64 * OWN) + wave * (16 * OWN);
  const int64_t o0 = (int64_t)blockIdx.y * other_block;
  const int64_t o1 = o0 + other_block < n_other ? o0 + other_block : n_other;
  // own side of the first products (B operands): <e>, <e>^2 and var = <e^2> - <e>^2 of columns KS lj .. KS lj + KS - 1
  T eo[OWN][KS], eo_sq[OWN][KS], eo_var[OWN][KS];
  acc_t acc[OWN][CT];
#pragma unroll
  for (int u = 0; u < OWN; ++u) {
    const int64_t rowi = own0 + 16 * u + li;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = KS * lj + s;
      const bool ok = rowi < n_own && k < K;
      const T e = ok ? E_own[rowi * K + k] : (T)0, e2 = ok ? E2_own[rowi * K + k] : (T)0;
      eo[u][s] = e;
      eo_sq[u][s] = e * e;
      const T v = e2 - e * e;
      eo_var[u][s] = v > (T)0 ? v : (T)0;
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[u][c] = (acc_t){(T)0, (T)0, (T)0, (T)0};
  }
  for (int64_t t0 = o0; t0 < o1; t0 += SR) {
    const int rows = (int)(o1 - t0 < SR ? o1 - t0 : SR);
    __syncthreads();
    for (int i = threadIdx.x; i < SR * KP; i += kBjThreads) {
      const int r = i / KP, k = i - r * KP;
      const bool ok = r < rows && k < K;
      tile_e[r * LSE + k] = ok ? E_other[(t0 + r) * K + k] : (T)0;
      tile_e[r * LSE + KP + k] = ok ? E2_other[(t0 + r) * K + k] : (T)0;
    }
    for (int i = threadIdx.x; i < SR * 16 * CT; i += kBjThreads) {
      const int r = i / (16 * CT), c = i - r * (16 * CT);
      tile_m[r * LSM + c] = (r < rows && c < ldm) ? M_other[(t0 + r) * (int64_t)ldm + c] : (T)0;  // (padding rows: 0)
    }
    __syncthreads();
    for (int tt = 0; tt < rows; tt += 16) {
      // other side of the first products (A operands): row tt + li, columns KS lj ..
      T a_e[KS], a_var[KS], a_e2[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const T e = tile_e[(tt + li) * LSE + KS * lj + s], e2 = tile_e[(tt + li) * LSE + KP + KS * lj + s];
        a_e[s] = e;
        a_e2[s] = e2;
        const T v = e2 - e * e;
        a_var[s] = v > (T)0 ? v : (T)0;
      }
      acc_t om[OWN];
#pragma unroll
      for (int u = 0; u < OWN; ++u) {
        acc_t z = {(T)0, (T)0, (T)0, (T)0}, q = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll
        for (int s = 0; s < KS; ++s) z = Mp::mfma(a_e[s], eo[u][s], z);
#pragma unroll
        for (int s = 0; s < KS; ++s) q = Mp::mfma(a_var[s], eo_sq[u][s], q);  // <e_own>^2 var_other
#pragma unroll
        for (int s = 0; s < KS; ++s) q = Mp::mfma(a_e2[s], eo_var[u][s], q);  // var_own <e_other^2>
#pragma unroll
        for (int r = 0; r < 4; ++r) om[u][r] = Mp::omega(z[r] * z[r] + q[r]);
      }
      // second product: accumulator register r of a lane is Omega[other = tt + row(lj, r)][own = li]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T* mrow = tile_m + (tt + Mp::row(lj, r)) * LSM + li;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const T b = mrow[16 * c];
#pragma unroll
          for (int u = 0; u < OWN; ++u) acc[u][c] = Mp::mfma(om[u][r], b, acc[u][c]);
        }
      }
    }
  }
  T* out = part + (int64_t)blockIdx.y * n_own * pc;
#pragma unroll
  for (int u = 0; u < OWN; ++u) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int col = 16 * c + li;
      if (col < pc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // acc register r of lane 16 q + i: out[own = row(q, r)][col = i]
          const int64_t rowi = own0 + 16 * u + Mp::row(lj, r);
          if (rowi < n_own) out[rowi * pc + col] = acc[u][c][r];
        }
      }
    }
  }
}

// rows -> the K (K + 1) / 2 distinct entries of <e e^T> (k <= l, row-major; the diagonal holds the second moments), padded
// with zeros to ldm columns: a thread per row
template <typename T>
__global__ __launch_bounds__(256) void k_pack_moments(int64_t n, int K, int ldm, const T* __restrict__ E,
                                                      const T* __restrict__ E2, T* __restrict__ M) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  T* m = M + i * ldm;
  int c = 0;
  for (int k = 0; k < K; ++k) {
    const T ek = E[i * K + k];
    m[c++] = E2[i * K + k];
    for (int l = k + 1; l < K; ++l) m[c++] = ek * E[i * K + l];
  }
  for (; c < ldm; ++c) m[c] = (T)0;
}

int bj_ct(int K) { return (K * (K + 1) / 2 + 15) / 16; }
// instantiated column-tile counts per padded width: the smallest one that holds K (K + 1) / 2 columns
int bj_ct_inst(int K) {
  const int need = bj_ct(K);
  if (K <= 4) return 1;
  if (K <= 8) return need <= 2 ? 2 : 3;
  if (K <= 12) return need <= 4 ? 4 : 5;
  return need <= 7 ? 7 : 9;
}

template <typename T, int KP, int CT>
const void* bj_kernel() { return (const void*)k_jaakkola_sweep<T, KP, CT>; }

template <typename T>
const void* bj_pick(int K, int* own, int* sr) {
  const int ct = bj_ct_inst(K);
#define MU_B(KP_, CT_)                                   \
  do {                                                   \
    *own = BjShape<T, KP_, CT_>::OWN;                    \
    *sr = BjShape<T, KP_, CT_>::SR;                      \
    return bj_kernel<T, KP_, CT_>();                     \
  } while (0)
  if (K <= 4) MU_B(4, 1);
  if (K <= 8) { if (ct == 2) MU_B(8, 2); MU_B(8, 3); }
  if (K <= 12) { if (ct == 4) MU_B(12, 4); MU_B(12, 5); }
  if (ct == 7) MU_B(16, 7);
  MU_B(16, 9);
#undef MU_B
}

template <typename T>
int bj_launch(int64_t n_own, int64_t n_other, int K, int64_t other_block, const void* E_own, const void* E2_own,
              const void* E_other, const void* E2_other, const void* M_other, int ldm, void* part, hipStream_t st) {
  int own = 1, sr = 32;
  (void)bj_pick<T>(K, &own, &sr);
  const int pc = K * (K + 1) / 2, ct = bj_ct_inst(K);
  const dim3 grid((unsigned)((n_own + 64 * own - 1) / (64 * own)), (unsigned)((n_other + other_block - 1) / other_block));
#define MU_GO(KP_, CT_)                                                                                               \
  hipLaunchKernelGGL((k_jaakkola_sweep<T, KP_, CT_>), grid, dim3(kBjThreads), 0, st, n_own, n_other, K, pc, other_block, \
                     (const T*)E_own, (const T*)E2_own, (const T*)E_other, (const T*)E2_other, (const T*)M_other, ldm,   \
                     (T*)part)
  if (K <= 4) MU_GO(4, 1);
  else if (K <= 8) { if (ct == 2) MU_GO(8, 2); else MU_GO(8, 3); }
  else if (K <= 12) { if (ct == 4) MU_GO(12, 4); else MU_GO(12, 5); }
  else if (ct == 7) MU_GO(16, 7);
  else MU_GO(16, 9);
#undef MU_GO
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // namespace

extern "C" {

int mu_mofa_jaakkola_cols(int K) { return K >= 1 && K <= 16 ? 16 * bj_ct_inst(K) : 0; }

int mu_mofa_pack_moments(int dtype, int64_t n, int K, int ldm, const void* d_E, const void* d_E2, void* d_M,
                         void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 16 && ldm >= K * (K + 1) / 2 && n >= 0, "1 <= K <= 16, ldm >= K (K + 1) / 2");
  if (n == 0) return MU_OK;
  MU_REQUIRE(d_E && d_E2 && d_M, "null pointer");
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_pack_moments<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, K, ldm, (const float*)d_E,
                       (const float*)d_E2, (float*)d_M);
  else
    hipLaunchKernelGGL(k_pack_moments<double>, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, K, ldm,
                       (const double*)d_E, (const double*)d_E2, (double*)d_M);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int64_t mu_mofa_jaakkola_blocks(int dtype, int K, int64_t n_own, int64_t n_other) {
  // rows of the other block per column block: whole LDS stages, the workgroups in whole rounds of the places the
  // kernel's occupancy gives (the rule of mu_mofa_poisson_blocks_for)
  if (dtype != MU_DTYPE_F32 && dtype != MU_DTYPE_F64) dtype = MU_DTYPE_F32;
  if (K < 1) K = 1;
  if (K > 16) K = 16;
  int own = 1, sr = 32;
  const void* kern = dtype == MU_DTYPE_F32 ? bj_pick<float>(K, &own, &sr) : bj_pick<double>(K, &own, &sr);
  const int64_t own_wgs = n_own > 0 ? (n_own + 64 * own - 1) / (64 * own) : 1;
  const int64_t tiles = n_other > 0 ? (n_other + sr - 1) / sr : 1;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kBjThreads, 0) != hipSuccess || occ < 1) {
    (void)hipGetLastError();
    occ = 2;
  }
  const int64_t slots = (int64_t)mu_num_cus() * occ;
  int64_t best_per = tiles, best_cost = -1, best_nb = 1;
  for (int r = 1; r <= 3; ++r) {
    int64_t nb = r * slots / own_wgs;
    if (nb < 1) nb = 1;
    if (nb > tiles) nb = tiles;
    const int64_t per = (tiles + nb - 1) / nb, nb_real = (tiles + per - 1) / per;
    const int64_t rounds = (own_wgs * nb_real + slots - 1) / slots;
    const int64_t cost = rounds * per;
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && nb_real < best_nb)) {
      best_cost = cost;
      best_per = per;
      best_nb = nb_real;
    }
  }
  return best_per * sr;
}

int mu_mofa_jaakkola_sweep(int dtype, int64_t n_own, int64_t n_other, int K, int64_t other_block, const void* d_E_own,
                           const void* d_E2_own, const void* d_E_other, const void* d_E2_other, const void* d_M_other,
                           int ldm, void* d_part, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(K >= 1 && K <= 16, "1 <= K <= 16");
  MU_REQUIRE(n_own >= 0 && n_other >= 0 && other_block >= 1 && ldm >= K * (K + 1) / 2, "shape");
  if (n_own == 0 || n_other == 0) return MU_OK;
  MU_REQUIRE(d_E_own && d_E2_own && d_E_other && d_E2_other && d_M_other && d_part, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  return dtype == MU_DTYPE_F32
             ? bj_launch<float>(n_own, n_other, K, other_block, d_E_own, d_E2_own, d_E_other, d_E2_other, d_M_other,
                                ldm, d_part, st)
             : bj_launch<double>(n_own, n_other, K, other_block, d_E_own, d_E2_own, d_E_other, d_E2_other, d_M_other,
                                 ldm, d_part, st);
}

}  // extern "C"
