// Tall-skinny products of a dense MOFA view Y [n x D] (row-major, leading dimension ldY) with the
// factor blocks, on the matrix cores.  These are the two passes over a dense view that one MOFA+
// iteration needs (the reference hands the arithmetic to mofapy2 through
// /root/reference/muon/_core/tools.py:583-585):
//
//   nn :  A[n x 16]  = Y * T          T [D x 16] = tau o <W>  (zero padded to 16 columns)
//   tn :  B[D x 16]  = Y^T * Z        Z [n x 16] = <Z>        (zero padded to 16 columns)
//
// Both stream Y exactly once (HBM bound: 2 x 16 flop per byte of f64 against ~10 needed to hide the
// f64 matrix cores behind 8 TB/s), which the library GEMMs do not at K = 10: rocprof on
// configs[3] in f64 shows 26 ms per product for Tensile's 128x64 / 128x128 macro tiles against
// 2.5 ms of streaming.  v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32; operand maps as in
// dense.hip:  A operand lane l = A[i = l & 15][k = l >> 4],  B operand lane l = B[k = l >> 4][j = l & 15],
// C/D reg r of lane l = C[row = (l >> 4) + 4 r][col = l & 15] (f64) / C[row = 4 (l >> 4) + r][col] (f32).
#include "common.hpp"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma;
template <> struct Mma<double> {
  typedef d4 acc_t;
  static __device__ __forceinline__ acc_t fma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lr, int r) { return lr + 4 * r; }
};
template <> struct Mma<float> {
  typedef f4 acc_t;
  static __device__ __forceinline__ acc_t fma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lr, int r) { return 4 * lr + r; }
};

// ---- nn: out[n x 16] = Y[n x D] * Tm[D x 16] --------------------------------------------------
// MFMA roles i = row, j = output column, k = d: the A operand wants 16 ROWS x 4 consecutive d per
// instruction, i.e. 16 different lines of Y.  Loading it like that runs at the texture unit's
// line-lookup rate (measured 5.0 ms for 8 GB in f32, 1.6 TB/s), so a wave stages a 16-row x
// 128-byte tile through LDS instead: two coalesced global_load_dwordx4 per tile (8 rows x one full
// line each), ds_write_b128, then the transposed operand reads (row stride 144 B: conflict free).
// r03: a workgroup owns kSub = 4 such tiles (64 rows) and its four waves split the columns (r04: every fourth tile
// each; partial products, added up through LDS in a fixed order): per wave the eight loads of the NEXT step
// over d are in flight while the 4 x CW/4 MFMAs of this one run (r02 had one tile per wave in flight
// and wrote it to LDS right behind its loads: no overlap inside a wave, 3.6 TB/s in f64), the CW/4
// operands of Tm (D x 16, a few MB: L2) are loaded once per 64 rows instead of once per 16, one step
// ahead, and 100 000 rows still give every CU six workgroups.  A wave's LDS writes and reads execute
// in order: one staging buffer per wave, no barrier inside the sweep.
// TY: the type Y is STORED in.  TY = float with T = double (r04): a view whose values are exact in f32 - AnnData's
// default dtype - streams half the bytes and is widened on its way out of LDS; every product and sum is the f64 one.
constexpr int kSub = 4;
// FAST (r04): Y 16-byte aligned with a leading dimension of whole pieces and D a whole number of tiles - the launch
// checks - so no load of the sweep is predicated: a row past the end reads the last row (its product is not stored).
// The general build predicates every piece AND every element of its fallback; unrolled eight times that is ~90
// conditional blocks a step, no 16-byte load survives (61 global_load_dword in the f32-storage instance) and the
// step runs at a fifth of the matrix cores' rate.
template <typename T, typename TY = T, bool FAST = false>
__global__ __launch_bounds__(256) void k_skinny_nn(int64_t n_rows, int64_t D, int64_t ldY,
                                                   const TY* __restrict__ Y, const T* __restrict__ Tm,
                                                   T* __restrict__ out, int quarters) {
  typedef typename Mma<T>::acc_t acc_t;
  constexpr int CW = 128 / (int)sizeof(TY);  // columns per tile: 32 (f32 storage) / 16 (f64)
  constexpr int PE = 16 / (int)sizeof(TY);   // elements per 16-byte piece: 4 / 2
  constexpr int RS = 144;                    // LDS row stride in bytes
  constexpr int NU = CW / 4;                 // MFMA steps per tile: 8 / 4
  constexpr int kStage = 4 * kSub * 16 * RS;                      // staging: 36 KiB
  constexpr int kRed = 4 * kSub * 16 * 16 * (int)sizeof(T);       // the waves' partial products
  __shared__ __attribute__((aligned(16))) char smem[kStage > kRed ? kStage : kRed];
  const int lane = threadIdx.x & 63;
  // (wave-uniform, and known to be: a loop whose trip count hangs on a per-lane `wave` is a divergent loop to the
  //  compiler, and it then moves every MFMA accumulator to VGPRs and back around each step - 128 instructions a step)
  const int wave = uniform32(threadIdx.x >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int prow = lane >> 3, piece = lane & 7;  // loader view: 8 rows x 8 pieces per instruction
  const bool vec_ok = ((ldY * (int64_t)sizeof(TY)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(Y) % 16) == 0);
  char* my_tiles = smem + wave * (kSub * 16 * RS);
  const int64_t n_tiles = (n_rows + 16 * kSub - 1) / (16 * kSub);
  // this wave's share of the columns, in whole tiles: every fourth tile (r04) - the four waves of a workgroup then
  // ask for 512 consecutive bytes of a row at about the same time instead of for four lines 40 KB apart (DRAM page
  // locality: the tn kernel, which reads 1 KiB per row and wave, streams the same matrix 23 % faster).  `quarters`
  // (tune nn_interleave = 1): r03's contiguous quarter per wave, kept for comparison
  const int64_t n_ct = (D + CW - 1) / CW;
  const int64_t dq = (n_ct * wave / 4) * CW, dq1 = (n_ct * (wave + 1) / 4) * CW;
  const int64_t dlo = quarters ? dq : (int64_t)wave * CW;
  const int64_t dhi = quarters ? (dq1 < D ? dq1 : D) : D;
  const int64_t dstep = quarters ? CW : 4 * CW;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t r0 = tile * (16 * kSub);
    acc_t acc[kSub];
#pragma unroll
    for (int q = 0; q < kSub; ++q) acc[q] = acc_t{0, 0, 0, 0};
    f4 yq[kSub][2];
    T bq[NU];
    auto gload = [&](int64_t d0) {  // this lane's two 16-byte pieces of every sub-tile at columns d0 ..
#pragma unroll
      for (int q = 0; q < kSub; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int64_t row = r0 + 16 * q + prow + 8 * h;
          const int64_t d = d0 + piece * PE;
          if constexpr (FAST) {
            const int64_t rc = row < n_rows ? row : n_rows - 1;
            // from asm: the compiler, left to itself, lands these prefetches in registers it reuses a few
            // instructions later and waits for each one right behind its issue (`s_waitcnt vmcnt(15)` x 8, then
            // copies) - the sweep then runs at the memory LATENCY (measured: 4.3 ms against 3.8 with the branchy
            // general build).  The wait is the one at the top of the next step (nn_wait)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(yq[q][h]) : "v"(Y + rc * ldY + d) : "memory");
            continue;
          }
          TY v[PE];
          if (row < n_rows && vec_ok && d + PE <= D) {
            yq[q][h] = *reinterpret_cast<const f4*>(Y + row * ldY + d);
          } else {
#pragma unroll
            for (int e = 0; e < PE; ++e) v[e] = (row < n_rows && d + e < D) ? Y[row * ldY + d + e] : (TY)0;
            __builtin_memcpy(&yq[q][h], v, 16);
          }
        }
    };
    auto bload = [&](int64_t d0) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int64_t d = d0 + 4 * u + lr;
        if constexpr (FAST) {  // (from asm like the pieces of Y: the wait is nn_wait's)
          if constexpr (sizeof(T) == 8)
            asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(bq[u]) : "v"(Tm + d * 16 + lc) : "memory");
          else
            asm volatile("global_load_dword %0, %1, off" : "=&v"(bq[u]) : "v"(Tm + d * 16 + lc) : "memory");
        } else {
          bq[u] = (d < D) ? Tm[d * 16 + lc] : (T)0;
        }
      }
    };
    if (dlo < dhi) {
      gload(dlo);
      bload(dlo);
    }
    for (int64_t d0 = dlo; d0 < dhi; d0 += dstep) {
      if constexpr (FAST) {
        static_assert(kSub == 4, "nn_wait names 2 kSub registers");
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(yq[0][0]), "+v"(yq[0][1]), "+v"(yq[1][0]), "+v"(yq[1][1]), "+v"(yq[2][0]), "+v"(yq[2][1]),
                       "+v"(yq[3][0]), "+v"(yq[3][1])
                     :
                     : "memory");
#pragma unroll
        for (int u = 0; u < NU; ++u) asm volatile("" : "+v"(bq[u]));  // (ordered behind the wait)
      }
#pragma unroll
      for (int q = 0; q < kSub; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          *reinterpret_cast<f4*>(&my_tiles[(q * 16 + prow + 8 * h) * RS + piece * 16]) = yq[q][h];
      T b[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) b[u] = bq[u];
      if (d0 + dstep < dhi) {  // the next step's operands: in flight under this step's MFMAs
        gload(d0 + dstep);
        bload(d0 + dstep);
      }
#pragma unroll
      for (int q = 0; q < kSub; ++q)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const T a = (T)*reinterpret_cast<const TY*>(&my_tiles[(q * 16 + lc) * RS + (4 * u + lr) * (int)sizeof(TY)]);
          acc[q] = Mma<T>::fma(a, b[u], acc[q]);
        }
    }
    // the four quarters, added in wave order
    __syncthreads();  // everyone is done with the staging buffers
    T* red = reinterpret_cast<T*>(smem);
#pragma unroll
    for (int q = 0; q < kSub; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[((wave * kSub + q) * 16 + Mma<T>::row(lr, r)) * 16 + lc] = acc[q][r];
    __syncthreads();
    for (int e = threadIdx.x; e < kSub * 16 * 16; e += 256) {
      const int64_t orow = r0 + e / 16;
      if (orow < n_rows)
        out[orow * 16 + (e & 15)] = ((red[e] + red[kSub * 256 + e]) + red[2 * kSub * 256 + e]) + red[3 * kSub * 256 + e];
    }
    __syncthreads();  // before the next step's staging writes
  }
}

// ---- tn: C[D x 16] = Y^T[D x n] * Z[n x 16] ------------------------------------------------------
// MFMA roles i = d (column of Y), j = output column, k = row n.  A wave owns kCT column tiles (128
// columns of Y: 1 KiB contiguous per row) and a strided share of the workgroup's row range; both
// operands load coalesced (4 rows x 16 consecutive elements per instruction).  Row splits give the
// grid its width; partial blocks are reduced in a fixed order (bit-reproducible).
constexpr int kCT = 8;

template <typename T, typename TY = T, bool PIPE = false>
__global__ __launch_bounds__(256) void k_skinny_tn_partial(int64_t n_rows, int64_t D, int64_t ldY,
                                                           const TY* __restrict__ Y,
                                                           const T* __restrict__ Z, int n_splits,
                                                           T* __restrict__ partial) {
  typedef typename Mma<T>::acc_t acc_t;
  __shared__ T red[16 * kCT * 16];
  const int lane = threadIdx.x & 63;
  // (wave-uniform, and known to be: a loop whose trip count hangs on a per-lane `wave` is a divergent loop to the
  //  compiler, and it then moves every MFMA accumulator to VGPRs and back around each step - 128 instructions a step)
  const int wave = uniform32(threadIdx.x >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int64_t cb = blockIdx.x;            // column block: columns [128 cb, 128 cb + 128)
  const int split = blockIdx.y;
  const int64_t c0 = cb * (16 * kCT);
  const int64_t n_groups = (n_rows + 3) / 4;  // 4 rows per MFMA step
  const int64_t g0 = n_groups * split / n_splits, g1 = n_groups * (split + 1) / n_splits;
  acc_t acc[kCT];
#pragma unroll
  for (int t = 0; t < kCT; ++t) acc[t] = acc_t{0, 0, 0, 0};
  // No predicated load in the loop (r04).  With `rk && cok[t] ? Y[..] : 0` the loop was nine conditional blocks and
  // the compiler kept the accumulators in VGPRs across them: 64 v_accvgpr_write before and 64 v_accvgpr_read after
  // the eight MFMAs of every step - more issue slots than the MFMAs themselves.  Addresses are clamped instead: a row
  // past the end reads the last row against z = 0, a column past D reads column D - 1 into an output row that is
  // never stored.  One basic block, the accumulators stay where the MFMAs leave them.
  int64_t coff[kCT];
#pragma unroll
  for (int t = 0; t < kCT; ++t) {
    const int64_t c = c0 + 16 * t + lc;
    coff[t] = c < D ? c : D - 1;
  }
  const int64_t last_row = n_rows - 1;
  // PIPE (tune tn_pipe = 1; NOT the default): two steps in flight per wave - the loads of step i + 1 issued before the
  // MFMAs of step i, two register sets, the loop unrolled by two.  It costs 50 registers (170-182 against 119-128:
  // two waves per SIMD instead of four) and measured 1-3 % slower on c4 in all three instances.
  TY xa[kCT], xb[kCT];
  T za, zb;
  auto load = [&](TY (&x)[kCT], T& z, int64_t grp) {
    const int64_t row = grp * 4 + lr;
    const bool rk = row < n_rows;
    const int64_t rc = rk ? row : last_row;
    const T zz = Z[rc * 16 + lc];
    z = rk ? zz : (T)0;
    const TY* yrow = Y + rc * ldY;
#pragma unroll
    for (int t = 0; t < kCT; ++t) x[t] = yrow[coff[t]];
  };
  auto mma = [&](const TY (&x)[kCT], T z) {
#pragma unroll
    for (int t = 0; t < kCT; ++t) acc[t] = Mma<T>::fma((T)x[t], z, acc[t]);
  };
  int64_t grp = g0 + wave;
  if constexpr (!PIPE) {
    for (; grp < g1; grp += 4) {
      load(xa, za, grp);
      mma(xa, za);
    }
  } else if (grp < g1) {
    load(xa, za, grp);
    for (;;) {
      const int64_t g2 = grp + 4;
      load(xb, zb, g2 < g1 ? g2 : grp);
      mma(xa, za);
      if (g2 >= g1) break;
      grp = g2 + 4;
      load(xa, za, grp < g1 ? grp : g2);
      mma(xb, zb);
      if (grp >= g1) break;
    }
  }
  // reduce the four waves through LDS in a fixed order
  for (int i = threadIdx.x; i < 16 * kCT * 16; i += 256) red[i] = (T)0;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < kCT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(16 * t + Mma<T>::row(lr, r)) * 16 + lc] += acc[t][r];
    }
    __syncthreads();
  }
  T* dst = partial + ((int64_t)split * D + c0) * 16;
  for (int i = threadIdx.x; i < 16 * kCT * 16; i += 256) {
    const int64_t c = c0 + (i >> 4);
    if (c < D) dst[i] = red[i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_skinny_tn_reduce(int64_t total, int n_splits,
                                                          const T* __restrict__ partial,
                                                          T* __restrict__ C) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  T acc = (T)0;
  for (int s = 0; s < n_splits; ++s) acc += partial[(int64_t)s * total + e];
  C[e] = acc;
}

// the branch-free build of k_skinny_nn: aligned pieces, whole tiles of 128 bytes
inline bool nn_fast(const void* Y, int64_t ldY, int64_t D, int64_t elem) {
  return (reinterpret_cast<uintptr_t>(Y) % 16) == 0 && (ldY * elem) % 16 == 0 && D > 0 && (D * elem) % 128 == 0;
}

inline int tn_splits(int64_t n_rows, int64_t D) {
  // enough workgroups for ~8 per CU, at least 256 rows per split
  const int64_t cbs = (D + 16 * kCT - 1) / (16 * kCT);
  int64_t s = ((int64_t)mu_num_cus() * 8 + cbs - 1) / cbs;
  const int64_t cap = (n_rows + 255) / 256;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

}  // namespace

extern "C" {

size_t mu_skinny_tn_worksize(int dtype, int64_t n_rows, int64_t D) {
  const size_t es = dtype == MU_DTYPE_F64 ? 8 : 4;
  return (size_t)tn_splits(n_rows, D) * (size_t)D * 16 * es + 256;
}

int mu_skinny_nn(int dtype, int64_t n_rows, int64_t D, int64_t ldY, const void* d_Y, const void* d_T,
                 void* d_out, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(n_rows >= 0 && D >= 0 && ldY >= D, "bad shape");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_out && (D == 0 || (d_Y && d_T)), "null pointer");
  int64_t blocks = (n_rows + 16 * kSub - 1) / (16 * kSub);  // 64-row steps, one per workgroup
  const int64_t cap = (int64_t)mu_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipStream_t st = (hipStream_t)stream;
  const int quarters = mu_tune_get("nn_interleave") == 1;
  const bool fast = nn_fast(d_Y, ldY, D, dtype == MU_DTYPE_F64 ? 8 : 4) && mu_tune_get("nn_fast_off") == 0;
  if (dtype == MU_DTYPE_F64 && fast)
    hipLaunchKernelGGL((k_skinny_nn<double, double, true>), dim3((unsigned)blocks), dim3(256), 0, st, n_rows, D, ldY,
                       (const double*)d_Y, (const double*)d_T, (double*)d_out, quarters);
  else if (dtype == MU_DTYPE_F64)
    hipLaunchKernelGGL(k_skinny_nn<double>, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, D, ldY,
                       (const double*)d_Y, (const double*)d_T, (double*)d_out, quarters);
  else if (fast)
    hipLaunchKernelGGL((k_skinny_nn<float, float, true>), dim3((unsigned)blocks), dim3(256), 0, st, n_rows, D, ldY,
                       (const float*)d_Y, (const float*)d_T, (float*)d_out, quarters);
  else
    hipLaunchKernelGGL(k_skinny_nn<float>, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, D, ldY,
                       (const float*)d_Y, (const float*)d_T, (float*)d_out, quarters);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_skinny_nn_f64_f32(int64_t n_rows, int64_t D, int64_t ldY, const float* d_Y, const double* d_T, double* d_out,
                         void* stream) {
  MU_REQUIRE(n_rows >= 0 && D >= 0 && ldY >= D, "bad shape");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_out && (D == 0 || (d_Y && d_T)), "null pointer");
  int64_t blocks = (n_rows + 16 * kSub - 1) / (16 * kSub);
  const int64_t cap = (int64_t)mu_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  const int quarters = mu_tune_get("nn_interleave") == 1;
  if (nn_fast(d_Y, ldY, D, sizeof(float)))
    hipLaunchKernelGGL((k_skinny_nn<double, float, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_rows,
                       D, ldY, d_Y, d_T, d_out, quarters);
  else
    hipLaunchKernelGGL((k_skinny_nn<double, float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_rows, D,
                       ldY, d_Y, d_T, d_out, quarters);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_skinny_tn_f64_f32(int64_t n_rows, int64_t D, int64_t ldY, const float* d_Y, const double* d_Z, double* d_C,
                         void* d_work, size_t work_bytes, void* stream) {
  MU_REQUIRE(n_rows >= 0 && D >= 0 && ldY >= D, "bad shape");
  if (D == 0) return MU_OK;
  MU_REQUIRE(d_C && d_work && (n_rows == 0 || (d_Y && d_Z)), "null pointer");
  MU_REQUIRE(work_bytes >= mu_skinny_tn_worksize(MU_DTYPE_F64, n_rows, D), "work buffer too small");
  const int S = tn_splits(n_rows, D);
  const int64_t cbs = (D + 16 * kCT - 1) / (16 * kCT);
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = D * 16;
  if (mu_tune_get("tn_pipe") == 1)
    hipLaunchKernelGGL((k_skinny_tn_partial<double, float, true>), dim3((unsigned)cbs, (unsigned)S), dim3(256), 0, st,
                       n_rows, D, ldY, d_Y, d_Z, S, (double*)d_work);
  else
    hipLaunchKernelGGL((k_skinny_tn_partial<double, float>), dim3((unsigned)cbs, (unsigned)S), dim3(256), 0, st, n_rows, D,
                       ldY, d_Y, d_Z, S, (double*)d_work);
  MU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_skinny_tn_reduce<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, S,
                     (const double*)d_work, d_C);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_skinny_tn(int dtype, int64_t n_rows, int64_t D, int64_t ldY, const void* d_Y, const void* d_Z,
                 void* d_C, void* d_work, size_t work_bytes, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(n_rows >= 0 && D >= 0 && ldY >= D, "bad shape");
  if (D == 0) return MU_OK;
  MU_REQUIRE(d_C && d_work && (n_rows == 0 || (d_Y && d_Z)), "null pointer");
  MU_REQUIRE(work_bytes >= mu_skinny_tn_worksize(dtype, n_rows, D), "work buffer too small");
  const int S = tn_splits(n_rows, D);
  const int64_t cbs = (D + 16 * kCT - 1) / (16 * kCT);
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = D * 16;
  const bool piped = mu_tune_get("tn_pipe") == 1;
  if (dtype == MU_DTYPE_F64) {
    if (piped)
      hipLaunchKernelGGL((k_skinny_tn_partial<double, double, true>), dim3((unsigned)cbs, (unsigned)S), dim3(256), 0, st,
                         n_rows, D, ldY, (const double*)d_Y, (const double*)d_Z, S, (double*)d_work);
    else
      hipLaunchKernelGGL(k_skinny_tn_partial<double>, dim3((unsigned)cbs, (unsigned)S), dim3(256), 0, st,
                         n_rows, D, ldY, (const double*)d_Y, (const double*)d_Z, S, (double*)d_work);
    MU_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_skinny_tn_reduce<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       total, S, (const double*)d_work, (double*)d_C);
  } else {
    if (piped)
      hipLaunchKernelGGL((k_skinny_tn_partial<float, float, true>), dim3((unsigned)cbs, (unsigned)S), dim3(256), 0, st,
                         n_rows, D, ldY, (const float*)d_Y, (const float*)d_Z, S, (float*)d_work);
    else
      hipLaunchKernelGGL(k_skinny_tn_partial<float>, dim3((unsigned)cbs, (unsigned)S), dim3(256), 0, st,
                         n_rows, D, ldY, (const float*)d_Y, (const float*)d_Z, S, (float*)d_work);
    MU_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_skinny_tn_reduce<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       total, S, (const float*)d_work, (float*)d_C);
  }
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
