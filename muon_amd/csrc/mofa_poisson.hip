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

// The stored entries of an f32 model (k_pois_sparse): rate, sigmoid and ln(rate) of a prediction on the hardware exp2 /
// log2 / rcp (r06; libm's log1pf + __logf and two IEEE divisions were ~110 of the ~170 vector instructions per 64
// entries, and the kernel was 57 % VALU-busy).  Unlike the dense sweep, which only SUMS softplus, an entry needs the
// rate to a RELATIVE accuracy (it divides by it and takes its logarithm), also where the prediction is far below zero
// and the rate is e^zeta: ln(1 + e) is taken as ln(d) e / (d - 1) with d = fl(1 + e) (Kahan's log1p: the rounding of
// d cancels in the quotient), e = exp(-|zeta|) <= 1.
struct PzEntry { float rate, sig, lnrate; };
__device__ __forceinline__ PzEntry pz_entry(float z) {
  const float e = __builtin_amdgcn_exp2f(fabsf(z) * -1.4426950408889634f);
  const float d = 1.0f + e, inv = __builtin_amdgcn_rcpf(d), dm1 = d - 1.0f;
  const float lnd = 0.6931471805599453f * __builtin_amdgcn_logf(d);
  const float l1p = dm1 == 0.f ? e : lnd * (e * __builtin_amdgcn_rcpf(dm1));
  PzEntry r;
  r.sig = z >= 0.f ? inv : e * inv;
  r.rate = fmaxf(fmaxf(z, 0.f) + l1p, pz_tiny<float>());
  r.lnrate = 0.6931471805599453f * __builtin_amdgcn_logf(r.rate);  // (rate >= 1e-30: a normal number)
  return r;
}
struct PzEntryD { double rate, sig, lnrate; };
__device__ __forceinline__ PzEntryD pz_entry(double z) {
  PzEntryD r;
  r.rate = pz_softplus(z);
  r.sig = pz_sigmoid(z);
  r.lnrate = pz_log(r.rate);
  return r;
}
// sum over the 64 lanes in six DPP additions, the total in lane 63 (wave_sum's __shfl_down is an LDS permute plus its
// index arithmetic per step: ~500 instructions for the 12 accumulators of a row)
__device__ __forceinline__ float pz_wave_sum63(float v) {
#define MU_DPP(CTRL, RM) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xf, true))
  v += MU_DPP(0xB1, 0xf);   // quad_perm [1, 0, 3, 2]
  v += MU_DPP(0x4E, 0xf);   // quad_perm [2, 3, 0, 1]
  v += MU_DPP(0x141, 0xf);  // row_half_mirror
  v += MU_DPP(0x140, 0xf);  // row_mirror: every lane of a row of 16 holds the row's sum
  v += MU_DPP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  v += MU_DPP(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef MU_DPP
  return v;
}
__device__ __forceinline__ double pz_wave_sum63(double v) { return wave_sum_all(v); }

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
__global__ __launch_bounds__(kPzThreads) void k_pois_dense(int64_t n_own, int64_t n_other, int K, int ld, int64_t other_block,
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
  if (own < n_own) pz_load_row<T, KP>(E_own + own * ld, e);  // (padding columns are zero)
  const T kown = ((MODE == 1 || MODE == 3) && own < n_own) ? kappa[own] : (T)0;
  T lsum = (T)0;
  for (int64_t t0 = o0; t0 < o1; t0 += kPzTile) {
    const int rows = (int)(o1 - t0 < kPzTile ? o1 - t0 : kPzTile);
    __syncthreads();
    for (int i = threadIdx.x; i < kPzTile * KP; i += kPzThreads) {
      const int r = i / KP, k = i - r * KP;
      tile[r][k] = r < rows ? E_other[(t0 + r) * ld + k] : (T)0;
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

// ---- the dense sweep on the matrix cores (r06; f32 models, K <= 16) ------------------------------------------------
// `k_pois_dense` is bound by its vector ALU: 2 KP multiply-adds and ~10 other instructions per (own, other) pair, 0.50 ms
// per sweep at 20 000 x 20 000, K = 10 (profiles/r05_mofa_ng_kernel_stats.md).  Both halves of that arithmetic are small
// matrix products, and v_mfma_f32_16x16x4_f32 is exact f32 at the vector rate on its OWN pipe, next to the transform:
//   1. zeta^T tile [16 other x 16 own] = E_other[16 x KP] E_own^T[KP x 16]   (KP / 4 instructions; A = other, B = own)
//      -> lane 16 q + c holds zeta[other = 4 q + r][own = c] in register r = 0..3
//   2. the transform in those registers (4 pairs per lane, every lane busy)
//   3. out tile [16 own x 16 k] += R^T[16 own x 16 other] E_other[16 other x 16 k]  (4 instructions; A = R, B = E_other)
// Step 3's A operand wants lane 16 j + i to hold R[other = pi(s, j)][own = i] at reduction step s - and a reduction index
// may be walked in any order as long as A and B agree: with pi(s, j) = 4 j + s that value IS register s of the lane
// that computed it, so the prediction never leaves its registers and no lane exchanges anything (B then reads row
// 4 j + s of the LDS tile).  The same freedom makes step 1's operands one contiguous run of KP / 4 columns per lane.
// A wave owns kPmOwn tiles of 16 own rows (their E_own operands and accumulators stay in registers), a workgroup 256 own
// rows; the other block is staged through the same 128-row LDS tiles as above (row stride padded where the step-3
// reads of four row groups would meet in the same banks).  Modes and the layout of `part` as k_pois_dense.
// r06, later: f64 models sweep on the f64 matrix cores with the same kernel - v_mfma_f64_16x16x4_f64's C/D rows are
// (lane >> 4) + 4 r instead of 4 (lane >> 4) + r, so the reduction index of step 3 is walked as row(j, s) of the
// instruction's own map (PmMap<T>::row; csrc/mofa_bernoulli.hip does the same) and the transform is libm's in f64.
template <typename T> struct PmMap;
template <> struct PmMap<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static constexpr int OWN = 4;  // 16-row own tiles per wave
  static __device__ __forceinline__ int row(int q, int r) { return 4 * q + r; }
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // sigmoid and softplus of a prediction; softplus in units of `unit()`, a padding row (zeta = 0) adds `pad()` units
  static __device__ __forceinline__ void transform(float z, bool clamp, float& sig, float& sp) {
    float a = z * -1.4426950408889634f;
    if (clamp) a = fminf(a, 126.0f);
    const float d = 1.0f + __builtin_amdgcn_exp2f(a);
    sig = __builtin_amdgcn_rcpf(d);
    sp = __builtin_amdgcn_logf(d) - a;
  }
  static __device__ __forceinline__ float unit() { return 0.6931471805599453f; }
  static __device__ __forceinline__ float pad() { return 1.0f; }
};
template <> struct PmMap<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static constexpr int OWN = 2;
  static __device__ __forceinline__ int row(int q, int r) { return q + 4 * r; }
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void transform(double z, bool, double& sig, double& sp) {
    sig = pz_sigmoid(z);
    sp = pz_softplus(z);
  }
  static __device__ __forceinline__ double unit() { return 1.0; }
  static __device__ __forceinline__ double pad() { return pz_softplus(0.0); }
};

template <typename T, int KP, int MODE>
__global__ __launch_bounds__(kPzThreads) void k_pois_mfma(int64_t n_own, int64_t n_other, int K, int ld, int64_t other_block,
                                                          const T* __restrict__ E_own, const T* __restrict__ E_other,
                                                          const T* __restrict__ kappa, T* __restrict__ part) {
  typedef PmMap<T> Mp;
  typedef typename Mp::acc_t acc_t;
  constexpr int kPmOwn = Mp::OWN;
  constexpr int KS = KP / 4;                                   // reduction steps of the prediction
  constexpr int LS = KP == 8 ? 12 : (KP == 16 ? 20 : KP);      // LDS row stride (dwords): 4 LS mod 64 in {16, 48}
  __shared__ T tile[kPzTile * LS + 16];  // (+16: step 3 reads 16 columns of every row, the last row's run past it)
  __shared__ T kap[kPzTile];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int64_t own0 = (int64_t)blockIdx.x * (64 * kPmOwn) + wave * (16 * kPmOwn);
  const int64_t o0 = (int64_t)blockIdx.y * other_block;
  const int64_t o1 = o0 + other_block < n_other ? o0 + other_block : n_other;
  T eo[kPmOwn][KS], kown[kPmOwn], lsum[kPmOwn];
  acc_t acc[kPmOwn];
#pragma unroll
  for (int u = 0; u < kPmOwn; ++u) {
    const int64_t row = own0 + 16 * u + li;  // (own = the tile's column = lane & 15)
#pragma unroll
    for (int s = 0; s < KS; ++s) eo[u][s] = row < n_own ? E_own[row * ld + KS * lj + s] : (T)0;
    kown[u] = ((MODE == 1 || MODE == 3) && row < n_own) ? kappa[row] : (T)0;
    lsum[u] = (T)0;
    acc[u] = (acc_t){(T)0, (T)0, (T)0, (T)0};
  }
  if (threadIdx.x < 16) tile[kPzTile * LS + threadIdx.x] = (T)0;
  if (LS != KP)
    for (int i = threadIdx.x; i < kPzTile; i += kPzThreads)
      for (int k = KP; k < LS; ++k) tile[i * LS + k] = (T)0;  // (the stride padding is read as columns >= KP too)
  for (int64_t t0 = o0; t0 < o1; t0 += kPzTile) {
    const int rows = (int)(o1 - t0 < kPzTile ? o1 - t0 : kPzTile);
    __syncthreads();
    for (int i = threadIdx.x; i < kPzTile * KP; i += kPzThreads) {
      const int r = i / KP, k = i - r * KP;
      tile[r * LS + k] = r < rows ? E_other[(t0 + r) * ld + k] : (T)0;
    }
    if (MODE == 0)
      for (int i = threadIdx.x; i < kPzTile; i += kPzThreads) kap[i] = i < rows ? kappa[t0 + i] : (T)0;
    __syncthreads();
    for (int tt = 0; tt < rows; tt += 16) {
      T a1[KS], b2[4], kp[4];
#pragma unroll
      for (int s = 0; s < KS; ++s) a1[s] = tile[(tt + li) * LS + KS * lj + s];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // (columns >= KP of B are the next row's values: finite, and output columns >= K are never stored)
        b2[s] = MODE != 2 ? tile[(tt + Mp::row(lj, s)) * LS + li] : (T)0;
        kp[s] = MODE == 0 ? kap[tt + Mp::row(lj, s)] : (T)0;
      }
      // all the first products of the step, then the transform and the second product register by register across the
      // own tiles: four independent chains instead of one (0.291 -> 0.265 ms in scripts/probes/pois_mfma_bench.hip;
      // the f32 matrix instructions and the vector ALU do not overlap there - first product alone 0.098 ms, transform
      // alone 0.112, both products alone 0.206 - so the order only trims dependency stalls).
      // The transform on the hardware exp2 / rcp / log2 (1 ulp each; libm's __logf alone is 14 instructions, an IEEE
      // division 10): with a = -zeta log2(e) and d = 1 + 2^a,  sigmoid = 1 / d  and  softplus = ln2 (log2 d - a).
      // a is clamped so that 2^a stays finite (zeta < -87: sigmoid 1e-38, softplus ln2 (126 - 126) = 0); MODE 1 as
      // MODE 3, so that the two give the same b bit for bit.  A padding row has zeta = 0: it adds exactly 1 to the
      // softplus sum (2^0 = 1, log2 2 = 1), taken off after the loop instead of a select per element.
      acc_t z[kPmOwn];
#pragma unroll
      for (int u = 0; u < kPmOwn; ++u) {
        z[u] = (acc_t){(T)0, (T)0, (T)0, (T)0};
#pragma unroll
        for (int s = 0; s < KS; ++s) z[u] = Mp::mfma(a1[s], eo[u][s], z[u]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int u = 0; u < kPmOwn; ++u) {
          T sig, sp;
          Mp::transform(z[u][r], MODE != 0, sig, sp);
          if (MODE == 2 || MODE == 3) lsum[u] += sp;
          if (MODE != 2) {
            const T rr = (MODE == 0 ? kp[r] : kown[u]) * z[u][r] - sig;
            acc[u] = Mp::mfma(rr, b2[r], acc[u]);
          }
        }
      }
    }
  }
  if (MODE == 2 || MODE == 3) {
    // padding rows: only the last 16-row tile of the block can be partial; this lane holds rows row(lj, 0 .. 3) of it
    const int rem = (int)((o1 - o0) & 15);
    int npad = 0;
    if (o1 > o0 && rem)
      for (int r = 0; r < 4; ++r) npad += Mp::row(lj, r) >= rem;
#pragma unroll
    for (int u = 0; u < kPmOwn; ++u) lsum[u] = -Mp::unit() * (lsum[u] - Mp::pad() * (T)npad);
  }
  if (MODE == 2 || MODE == 3) {
#pragma unroll
    for (int u = 0; u < kPmOwn; ++u) {  // the four row groups of a tile hold the four quarters of an own row's sum
      lsum[u] += __shfl_xor(lsum[u], 16);
      lsum[u] += __shfl_xor(lsum[u], 32);
    }
  }
  const int ostride = MODE == 2 ? 1 : (MODE == 3 ? K + 1 : K);
  T* out = part + (int64_t)blockIdx.y * n_own * ostride;
#pragma unroll
  for (int u = 0; u < kPmOwn; ++u) {
    if (MODE != 2 && li < K) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // acc register r of lane 16 q + c: out[own = row(q, r)][k = c]
        const int64_t row = own0 + 16 * u + Mp::row(lj, r);
        if (row < n_own) out[row * ostride + li] = acc[u][r];
      }
    }
    if ((MODE == 2 || MODE == 3) && lj == 0) {
      const int64_t row = own0 + 16 * u + li;
      if (row < n_own) out[row * ostride + (MODE == 3 ? K : 0)] = lsum[u];
    }
  }
}

// The stored entries: a wave per own row, a lane per entry.  MODE as above; (indptr, indices, values) = CSR of the view
// (MODE 0, 2) or of its transpose (MODE 1, 3); the result is ADDED to out (which holds the dense part).
//
// r06: the kernel was bound by its own dependency chain, not by anything the machine runs out of.  A step of 64 entries
// reads (index, value) from the CSR stream (HBM: ~2 us under load), then gathers the other block's rows through the L2
// (~0.7 us), then computes - and the next step's reads were issued after that, so a row of ~1100 entries was 18 round
// trips one after the other and the 8 waves a SIMD can hold were all there is to overlap them: 0.22 ms per pass at
// 22.8 M entries whatever else changed (170 -> 60 vector instructions per step: nothing; a third of the cache-line
// look-ups per entry with four lanes per entry: nothing; the next step's loads issued one step ahead: -8 %).  The steps
// of a row do not depend on each other, so they now go in BATCHES of kPsBatch: all the batch's gathers are issued
// together, the (index, value) pairs of the next batch right behind them, then the batch is computed - a row is
// ~ceil(18 / 4) round trips to the L2, the HBM latency behind the arithmetic.
// A lane past the end of the row carries y = 0 and row 0 of the other block: it adds 0 x (finite) everywhere, so no
// step needs a guard.
constexpr int kPsBatch = 4;

template <typename T, int KP, int MODE>
__global__ __launch_bounds__(256) void k_pois_sparse(int64_t n_own, int K, int ld, const int64_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices, const T* __restrict__ values,
                                                     const T* __restrict__ E_own, const T* __restrict__ E_other,
                                                     T* __restrict__ out) {
  constexpr int kRowRegs = KP * (int)sizeof(T) / 4;  // registers of one gathered row
  constexpr int U = 48 / kRowRegs >= kPsBatch ? kPsBatch : (48 / kRowRegs >= 1 ? 48 / kRowRegs : 1);  // <= 48 registers of rows in flight
  const int lane = threadIdx.x & 63;
  const int64_t own = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (own >= n_own) return;
  T e[KP], acc[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) acc[k] = (T)0;
  pz_load_row<T, KP>(E_own + own * ld, e);
  T lsum = (T)0;
  const int64_t lo = indptr[own], hi = indptr[own + 1];
  int32_t jn[U];
  T yn[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t p = lo + 64 * u + lane;
    jn[u] = 0;
    yn[u] = (T)0;
    if (p < hi) { jn[u] = indices[p]; yn[u] = values[p]; }
  }
  for (int64_t base = lo; base < hi; base += 64 * U) {  // (wave-uniform trip count)
    T o[U][KP], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      y[u] = yn[u];
      pz_load_row<T, KP>(E_other + (int64_t)jn[u] * ld, o[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + 64 * (U + u) + lane;
      jn[u] = 0;
      yn[u] = (T)0;
      if (p < hi) { jn[u] = indices[p]; yn[u] = values[p]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      T zeta = (T)0;
#pragma unroll
      for (int k = 0; k < KP; ++k) zeta += e[k] * o[u][k];
      const auto en = pz_entry(zeta);
      if (MODE == 2 || MODE == 3) lsum += y[u] * en.lnrate;
      if (MODE != 2) {
        const T c = sizeof(T) == 4 ? (T)(en.sig * y[u] * (T)__builtin_amdgcn_rcpf((float)en.rate)) : en.sig * y[u] / en.rate;
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[k] += c * o[u][k];
      }
    }
  }
  if (MODE == 2) {
    lsum = pz_wave_sum63(lsum);
    if (lane == 63) out[own] += lsum;
  } else {
    const int ostride = MODE == 3 ? K + 1 : K;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      if (k < K) {  // (K is wave-uniform)
        const T s = pz_wave_sum63(acc[k]);
        if (lane == 63) out[own * ostride + k] += s;
      }
    }
    if (MODE == 3) {
      lsum = pz_wave_sum63(lsum);
      if (lane == 63) out[own * ostride + K] += lsum;
    }
  }
}

// The same pass with FOUR lanes per entry (r06, KP = 12 / 16, i.e. 9 <= K <= 16): what bounds the kernel above once its
// dependency chain is out of the way is the cache-line look-up rate of the gathers - the texture addresser was 77 % busy
// (profiles/r06_pois_pmc.txt), one look-up per lane and 16-byte piece, three per entry, and every line it finds
// is used for 16 of its 64 bytes.  Here the four lanes of an entry read four consecutive 16-byte pieces of the row: with
// rows padded to ld = 16 columns (64-byte aligned) an entry is ONE look-up, a third of the traffic through the addresser.
// The price is that a prediction is now spread over four lanes.  Summing it into all four and repeating the transform
// in each would quadruple the vector work (and was no faster); instead four steps are reduced together - a 4 x 4
// transpose-reduce inside the quad, two select / DPP-add rounds - which leaves lane q of a quad with the complete
// prediction of ITS step q, so the transform runs once per entry, on every lane, and the four scale factors go back
// to the quad by DPP broadcast.  Batches as above: kPqGroups x 4 steps x 16 entries in flight per wave.
template <typename T>
__device__ __forceinline__ void pz_load4(const T* __restrict__ p, T (&o)[4]) {
  constexpr int V = 16 / (int)sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const vec_t* q = reinterpret_cast<const vec_t*>(p);
#pragma unroll
  for (int i = 0; i < 4 / V; ++i) {
    const vec_t v = q[i];
#pragma unroll
    for (int u = 0; u < V; ++u) o[i * V + u] = v[u];
  }
}
template <int CTRL>
__device__ __forceinline__ float pz_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double pz_dpp(double v) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xf, 0xf, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

template <typename T, int KP, int MODE>
__global__ __launch_bounds__(256) void k_pois_sparse_quad(int64_t n_own, int K, int ld, const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices,
                                                          const T* __restrict__ values, const T* __restrict__ E_own,
                                                          const T* __restrict__ E_other, T* __restrict__ out) {
  static_assert(KP == 12 || KP == 16, "four lanes of four columns per entry");
  constexpr int G = sizeof(T) == 4 ? 2 : 1;  // groups of four steps per batch
  constexpr int S = 4 * G;                   // steps per batch, 16 entries each
  const int lane = threadIdx.x & 63, q = lane & 3, slot = lane >> 2;
  const int64_t own = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (own >= n_own) return;
  const bool act = 4 * q < KP;  // (KP = 12: the fourth lane of an entry holds no columns; it re-reads the first piece)
  const int col = act ? 4 * q : 0;
  const bool odd = q & 1, upper = q & 2;
  T e[4], acc[4];
  pz_load4<T>(E_own + own * ld + col, e);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc[i] = (T)0;
    if (!act) e[i] = (T)0;
  }
  T lsum = (T)0;
  const int64_t lo = indptr[own], hi = indptr[own + 1];
  // the CSR stream is read a lane per entry - one instruction per 64 entries and array, not one per step of 16: the
  // addresser pays ~16 cycles per memory instruction whatever it fetches (profiles/r06_pois_pmc.txt) - and a
  // step's 16 (index, value) pairs reach its quads through the LDS crossbar (ds_bpermute)
  constexpr int C = (S + 3) / 4;  // 64-entry chunks per batch
  int32_t jn[C];
  T yn[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int64_t p = lo + 64 * c + lane;
    jn[c] = 0;
    yn[c] = (T)0;
    if (p < hi) { jn[c] = indices[p]; yn[c] = values[p]; }
  }
  for (int64_t base = lo; base < hi; base += 16 * S) {  // (wave-uniform trip count)
    T o[S][4], y[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int src = 16 * (s & 3) + slot;  // the lane that read entry 16 s + slot of the batch
      const int32_t j = __shfl(jn[s >> 2], src, 64);
      y[s] = __shfl(yn[s >> 2], src, 64);
      pz_load4<T>(E_other + (int64_t)j * ld + col, o[s]);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t p = base + 16 * S + 64 * c + lane;
      jn[c] = 0;
      yn[c] = (T)0;
      if (p < hi) { jn[c] = indices[p]; yn[c] = values[p]; }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      T z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        z[u] = e[0] * o[4 * g + u][0];
#pragma unroll
        for (int i = 1; i < 4; ++i) z[u] += e[i] * o[4 * g + u][i];
      }
      // 4 x 4 transpose-reduce: lane q ends up with the sum over the quad of z[q]
      const T a0 = (odd ? z[1] : z[0]) + pz_dpp<0xB1>(odd ? z[0] : z[1]);   // steps 0 | 1, summed over the lane pair
      const T a1 = (odd ? z[3] : z[2]) + pz_dpp<0xB1>(odd ? z[2] : z[3]);   // steps 2 | 3
      const T zeta = (upper ? a1 : a0) + pz_dpp<0x4E>(upper ? a0 : a1);     // step q
      const T y01 = odd ? y[4 * g + 1] : y[4 * g + 0], y23 = odd ? y[4 * g + 3] : y[4 * g + 2];
      const T yq = upper ? y23 : y01;
      const auto en = pz_entry(zeta);
      if (MODE == 2 || MODE == 3) lsum += yq * en.lnrate;
      if (MODE != 2) {
        const T c = sizeof(T) == 4 ? (T)(en.sig * yq * (T)__builtin_amdgcn_rcpf((float)en.rate)) : en.sig * yq / en.rate;
        const T c0 = pz_dpp<0x00>(c), c1 = pz_dpp<0x55>(c), c2 = pz_dpp<0xAA>(c), c3 = pz_dpp<0xFF>(c);  // quad broadcasts
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] += c0 * o[4 * g + 0][i];
          acc[i] += c1 * o[4 * g + 1][i];
          acc[i] += c2 * o[4 * g + 2][i];
          acc[i] += c3 * o[4 * g + 3][i];
        }
      }
    }
  }
  const int ostride = MODE == 2 ? 1 : (MODE == 3 ? K + 1 : K);
  if (MODE != 2) {
    // the 16 entry slots fold: lanes with the same q hold the same four columns
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] += pz_dpp<0x124>(acc[i]);  // row_ror:4
      acc[i] += pz_dpp<0x128>(acc[i]);  // row_ror:8: every lane holds the sum over the four quads of its row of 16
      acc[i] += __shfl_xor(acc[i], 16, 64);
      acc[i] += __shfl_xor(acc[i], 32, 64);
    }
    if (slot == 0 && act) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (col + i < K) out[own * ostride + col + i] += acc[i];
    }
  }
  if (MODE == 2 || MODE == 3) {
    lsum = wave_sum(lsum);
    if (lane == 0) out[own * ostride + (MODE == 3 ? K : 0)] += lsum;
  }
}

template <typename T, int KP>
int pois_dense_launch(int mode, int64_t n_own, int64_t n_other, int K, int ld, int64_t other_block, const void* E_own,
                      const void* E_other, const void* kappa, void* part, hipStream_t st) {
  const dim3 grid((unsigned)((n_own + kPzThreads - 1) / kPzThreads), (unsigned)((n_other + other_block - 1) / other_block));
#define MU_GO(MD)                                                                                                      \
  hipLaunchKernelGGL((k_pois_dense<T, KP, MD>), grid, dim3(kPzThreads), 0, st, n_own, n_other, K, ld, other_block,      \
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
int pois_mfma_launch(int mode, int64_t n_own, int64_t n_other, int K, int ld, int64_t other_block, const void* E_own,
                     const void* E_other, const void* kappa, void* part, hipStream_t st) {
  const int own_rows = 64 * PmMap<T>::OWN;
  const dim3 grid((unsigned)((n_own + own_rows - 1) / own_rows), (unsigned)((n_other + other_block - 1) / other_block));
#define MU_GO(MD)                                                                                                 \
  hipLaunchKernelGGL((k_pois_mfma<T, KP, MD>), grid, dim3(kPzThreads), 0, st, n_own, n_other, K, ld, other_block, \
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
int pois_sparse_launch(int mode, int64_t n_own, int K, int ld, const int64_t* indptr, const int32_t* indices,
                       const void* values, const void* E_own, const void* E_other, void* out, hipStream_t st) {
  const unsigned blocks = (unsigned)((n_own + 3) / 4);
#define MU_GO(KERNEL, MD)                                                                                            \
  hipLaunchKernelGGL((KERNEL<T, KP, MD>), dim3(blocks), dim3(256), 0, st, n_own, K, ld, indptr, indices,             \
                     (const T*)values, (const T*)E_own, (const T*)E_other, (T*)out)
  if constexpr (KP == 12 || KP == 16) {
    if (mu_tune_get("pois_lane") <= 0) {  // four lanes per entry (k_pois_sparse_quad)
      if (mode == 0) MU_GO(k_pois_sparse_quad, 0);
      else if (mode == 1) MU_GO(k_pois_sparse_quad, 1);
      else if (mode == 2) MU_GO(k_pois_sparse_quad, 2);
      else MU_GO(k_pois_sparse_quad, 3);
      MU_CHECK_LAUNCH();
      return MU_OK;
    }
  }
  if (mode == 0) MU_GO(k_pois_sparse, 0);
  else if (mode == 1) MU_GO(k_pois_sparse, 1);
  else if (mode == 2) MU_GO(k_pois_sparse, 2);
  else MU_GO(k_pois_sparse, 3);
#undef MU_GO
  MU_CHECK_LAUNCH();
  return MU_OK;
}

template <typename T, int KP>
const void* pois_mfma_ptr(int mode) {
  return mode == 0 ? (const void*)k_pois_mfma<T, KP, 0> : mode == 1 ? (const void*)k_pois_mfma<T, KP, 1>
       : mode == 2 ? (const void*)k_pois_mfma<T, KP, 2> : (const void*)k_pois_mfma<T, KP, 3>;
}
template <typename T, int KP>
const void* pois_dense_ptr(int mode) {
  return mode == 0 ? (const void*)k_pois_dense<T, KP, 0> : mode == 1 ? (const void*)k_pois_dense<T, KP, 1>
       : mode == 2 ? (const void*)k_pois_dense<T, KP, 2> : (const void*)k_pois_dense<T, KP, 3>;
}
bool pois_use_mfma(int dtype, int K) { return K <= 16 && mu_tune_get("pois_valu") <= 0; }
int pois_own_rows(int dtype, int K) {  // own rows of a workgroup of the dense sweep
  return pois_use_mfma(dtype, K) ? 64 * (dtype == MU_DTYPE_F32 ? PmMap<float>::OWN : PmMap<double>::OWN) : kPzThreads;
}
const void* pois_dense_kernel(int dtype, int mode, int K) {
  if (pois_use_mfma(dtype, K)) {
#define MU_Q(T_)                                                                                                  \
  (K <= 4 ? pois_mfma_ptr<T_, 4>(mode) : K <= 8 ? pois_mfma_ptr<T_, 8>(mode) : K <= 12 ? pois_mfma_ptr<T_, 12>(mode) \
                                                                                       : pois_mfma_ptr<T_, 16>(mode))
    return dtype == MU_DTYPE_F32 ? MU_Q(float) : MU_Q(double);
#undef MU_Q
  }
#define MU_P(T_)                                                                                                   \
  (K <= 4 ? pois_dense_ptr<T_, 4>(mode) : K <= 8 ? pois_dense_ptr<T_, 8>(mode) : K <= 12 ? pois_dense_ptr<T_, 12>(mode) \
   : K <= 16 ? pois_dense_ptr<T_, 16>(mode) : pois_dense_ptr<T_, 32>(mode))
  return dtype == MU_DTYPE_F32 ? MU_P(float) : MU_P(double);
#undef MU_P
}

}  // namespace

extern "C" {

int64_t mu_mofa_poisson_blocks_for(int dtype, int mode, int K, int64_t n_own, int64_t n_other) {
  // Column blocks of the dense sweep for the kernel that (dtype, mode, K) selects.  Workgroups do equal work, so the
  // sweep lasts (rounds of resident workgroups) x (tiles per workgroup): r05's "~8 workgroups per CU" put 2054 workgroups
  // on the 1792 places of a kernel that fits 7 per CU - a second round for 13 % of them, 1.4x the time.  Here the
  // number of places comes from the kernel's own occupancy and the split with the smallest rounds x tiles wins (ties:
  // fewer blocks, i.e. less partial-result traffic).
  if (dtype != MU_DTYPE_F32 && dtype != MU_DTYPE_F64) dtype = MU_DTYPE_F32;
  if (mode < 0 || mode > 3) mode = 0;
  if (K < 1) K = 1;
  if (K > 32) K = 32;
  const int own_rows = pois_own_rows(dtype, K);
  const int64_t own_wgs = n_own > 0 ? (n_own + own_rows - 1) / own_rows : 1;
  const int64_t tiles = n_other > 0 ? (n_other + kPzTile - 1) / kPzTile : 1;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, pois_dense_kernel(dtype, mode, K), kPzThreads, 0) != hipSuccess ||
      occ < 1) {
    (void)hipGetLastError();
    occ = 4;
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
  return best_per * kPzTile;
}

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

static int pois_padded_width(int K) { return K <= 4 ? 4 : K <= 8 ? 8 : K <= 12 ? 12 : K <= 16 ? 16 : 32; }

int mu_mofa_poisson_dense_ld(int dtype, int mode, int64_t n_own, int64_t n_other, int K, int ld, int64_t other_block,
                             const void* d_E_own, const void* d_E_other, const void* d_kappa, void* d_part,
                             void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(mode >= 0 && mode <= 3 && K >= 1 && K <= 32, "mode 0..3, 1 <= K <= 32");
  MU_REQUIRE(ld >= pois_padded_width(K) && ld % 4 == 0, "ld: a multiple of 4, at least the padded width of K");
  MU_REQUIRE(n_own >= 0 && n_other >= 0 && other_block >= 1, "shape");
  if (n_own == 0 || n_other == 0) return MU_OK;
  MU_REQUIRE(d_E_own && d_E_other && d_part && (mode == 2 || d_kappa), "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (pois_use_mfma(dtype, K)) {  // the matrix-core sweep (k_pois_mfma)
#define MU_M(T_, KP_) pois_mfma_launch<T_, KP_>(mode, n_own, n_other, K, ld, other_block, d_E_own, d_E_other, d_kappa, d_part, st)
#define MU_MK(T_) (K <= 4 ? MU_M(T_, 4) : K <= 8 ? MU_M(T_, 8) : K <= 12 ? MU_M(T_, 12) : MU_M(T_, 16))
    return dtype == MU_DTYPE_F32 ? MU_MK(float) : MU_MK(double);
#undef MU_MK
#undef MU_M
  }
#define MU_D(T_, KP_) pois_dense_launch<T_, KP_>(mode, n_own, n_other, K, ld, other_block, d_E_own, d_E_other, d_kappa, d_part, st)
#define MU_DK(T_) (K <= 4 ? MU_D(T_, 4) : K <= 8 ? MU_D(T_, 8) : K <= 12 ? MU_D(T_, 12) : K <= 16 ? MU_D(T_, 16) : MU_D(T_, 32))
  return dtype == MU_DTYPE_F32 ? MU_DK(float) : MU_DK(double);
#undef MU_DK
#undef MU_D
}

int mu_mofa_poisson_dense(int dtype, int mode, int64_t n_own, int64_t n_other, int K, int64_t other_block,
                          const void* d_E_own, const void* d_E_other, const void* d_kappa, void* d_part, void* stream) {
  MU_REQUIRE(K >= 1 && K <= 32, "1 <= K <= 32");
  return mu_mofa_poisson_dense_ld(dtype, mode, n_own, n_other, K, pois_padded_width(K), other_block, d_E_own, d_E_other,
                                  d_kappa, d_part, stream);
}

int mu_mofa_poisson_sparse_ld(int dtype, int mode, int64_t n_own, int K, int ld, const int64_t* d_indptr,
                              const int32_t* d_indices, const void* d_values, const void* d_E_own, const void* d_E_other,
                              void* d_out, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(mode >= 0 && mode <= 3 && K >= 1 && K <= 32, "mode 0..3, 1 <= K <= 32");
  MU_REQUIRE(ld >= pois_padded_width(K) && ld % 4 == 0, "ld: a multiple of 4, at least the padded width of K");
  if (n_own <= 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_E_own && d_E_other && d_out, "null pointer");
  hipStream_t st = (hipStream_t)stream;
#define MU_S(T_, KP_) pois_sparse_launch<T_, KP_>(mode, n_own, K, ld, d_indptr, d_indices, d_values, d_E_own, d_E_other, d_out, st)
#define MU_SK(T_) (K <= 4 ? MU_S(T_, 4) : K <= 8 ? MU_S(T_, 8) : K <= 12 ? MU_S(T_, 12) : K <= 16 ? MU_S(T_, 16) : MU_S(T_, 32))
  return dtype == MU_DTYPE_F32 ? MU_SK(float) : MU_SK(double);
#undef MU_SK
#undef MU_S
}

int mu_mofa_poisson_sparse(int dtype, int mode, int64_t n_own, int K, const int64_t* d_indptr, const int32_t* d_indices,
                           const void* d_values, const void* d_E_own, const void* d_E_other, void* d_out, void* stream) {
  MU_REQUIRE(K >= 1 && K <= 32, "1 <= K <= 32");
  return mu_mofa_poisson_sparse_ld(dtype, mode, n_own, K, pois_padded_width(K), d_indptr, d_indices, d_values, d_E_own,
                                   d_E_other, d_out, stream);
}

}  // extern "C"
