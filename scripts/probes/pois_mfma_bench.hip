// Stand-alone experiments on the matrix-core sweep of csrc/mofa_poisson.hip (k_pois_mfma, mode 0 shape): which part of
// the loop the time belongs to.  hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form pois_mfma_bench.hip -o /tmp/pmb
//   VAR 0: the kernel as shipped (mode 0)      1: no transform (R = kappa zeta)      2: transform, no second product
//   VAR 3: both products, transform = 1 fma     4: first product only                 5: transform only (zeta from VALU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kTile = 128, kThreads = 256;

template <int KP, int VAR, int OWN, int WPE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k(int64_t n_own, int64_t n_other, int K, int64_t other_block, const float* __restrict__ E_own,
                                          const float* __restrict__ E_other, const float* __restrict__ kappa,
                                          float* __restrict__ part) {
  constexpr int KS = KP / 4, LS = KP;
  __shared__ float tile[kTile * LS + 16];
  __shared__ float kap[kTile];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lj = lane >> 4;
  const int64_t own0 = (int64_t)blockIdx.x * (4 * 16 * OWN) + wave * (16 * OWN);
  const int64_t o0 = (int64_t)blockIdx.y * other_block, o1 = o0 + other_block < n_other ? o0 + other_block : n_other;
  float eo[OWN][KS];
  f32x4 acc[OWN];
  for (int u = 0; u < OWN; ++u) {
    const int64_t row = own0 + 16 * u + li;
    for (int s = 0; s < KS; ++s) eo[u][s] = row < n_own ? E_own[row * KP + KS * lj + s] : 0.f;
    acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (threadIdx.x < 16) tile[kTile * LS + threadIdx.x] = 0.f;
  for (int64_t t0 = o0; t0 < o1; t0 += kTile) {
    const int rows = (int)(o1 - t0 < kTile ? o1 - t0 : kTile);
    __syncthreads();
    for (int i = threadIdx.x; i < kTile * KP; i += kThreads) {
      const int r = i / KP, kk = i - r * KP;
      tile[r * LS + kk] = r < rows ? E_other[(t0 + r) * KP + kk] : 0.f;
    }
    for (int i = threadIdx.x; i < kTile; i += kThreads) kap[i] = i < rows ? kappa[t0 + i] : 0.f;
    __syncthreads();
    for (int tt = 0; tt < rows; tt += 16) {
      float a1[KS], b2[4], kp[4];
#pragma unroll
      for (int s = 0; s < KS; ++s) a1[s] = tile[(tt + li) * LS + KS * lj + s];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        b2[s] = tile[(tt + 4 * lj + s) * LS + li];
        kp[s] = kap[tt + 4 * lj + s];
      }
      if (VAR == 8 || VAR == 9) {
        f32x4 zz[OWN];
#pragma unroll
        for (int u = 0; u < OWN; ++u) {
          zz[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KS; ++s) zz[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], eo[u][s], zz[u], 0, 0, 0);
        }
        if (VAR == 9) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int u = 0; u < OWN; ++u) {
            const float rr = kp[r] * zz[u][r] - __builtin_amdgcn_rcpf(1.0f + __expf(-zz[u][r]));
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(rr, b2[r], acc[u], 0, 0, 0);
          }
        }
        continue;
      }
#pragma unroll
      for (int u = 0; u < OWN; ++u) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (VAR == 5) {
#pragma unroll
          for (int r = 0; r < 4; ++r) z[r] = a1[r % KS] * eo[u][r % KS] + b2[r];
        } else {
#pragma unroll
          for (int s = 0; s < KS; ++s) z = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], eo[u][s], z, 0, 0, 0);
        }
        if (VAR == 4) {
          acc[u] += z;
          continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float rr;
          if (VAR == 1 || VAR == 3) rr = kp[r] * z[r];
          else rr = kp[r] * z[r] - __builtin_amdgcn_rcpf(1.0f + __expf(-z[r]));
          if (VAR == 2 || VAR == 5) acc[u][r] += rr * b2[r];
          else acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(rr, b2[r], acc[u], 0, 0, 0);
        }
      }
      if (VAR == 6 || VAR == 7) {
#pragma unroll
        for (int i = 0; i < 7 * OWN; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VAR == 6 ? 4 : 3, 0);
        }
      }
    }
  }
  float* out = part + (int64_t)blockIdx.y * n_own * K;
  for (int u = 0; u < OWN; ++u)
    if (li < K)
      for (int r = 0; r < 4; ++r) {
        const int64_t row = own0 + 16 * u + 4 * lj + r;
        if (row < n_own) out[row * K + li] = acc[u][r];
      }
}

template <int VAR, int OWN, int WPE>
void run(const char* what, int64_t N, int64_t D, int K, const float* Eo, const float* Et, const float* kap, float* part, int per_tiles) {
  const int64_t blk = (int64_t)per_tiles * kTile;
  dim3 grid((unsigned)((N + 64 * OWN - 1) / (64 * OWN)), (unsigned)((D + blk - 1) / blk));
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k<12, VAR, OWN, WPE>, kThreads, 0);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<12, VAR, OWN, WPE>), grid, dim3(kThreads), 0, 0, N, D, K, blk, Eo, Et, kap, part);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<12, VAR, OWN, WPE>), grid, dim3(kThreads), 0, 0, N, D, K, blk, Eo, Et, kap, part);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("%-44s own tiles %d  wpe %d  occ %d WG/CU  grid %u x %u = %u WGs  %.3f ms\n", what, OWN, WPE, occ, grid.x, grid.y, grid.x * grid.y, ms / reps);
}

int main(int argc, char** argv) {
  const int64_t N = 20000, D = 20000;
  const int K = 10, KP = 12;
  std::vector<float> ho(N * KP, 0.f), ht(D * KP, 0.f), hk(D);
  srand(1);
  for (int64_t i = 0; i < N; ++i) for (int k2 = 0; k2 < K; ++k2) ho[i * KP + k2] = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  for (int64_t i = 0; i < D; ++i) { for (int k2 = 0; k2 < K; ++k2) ht[i * KP + k2] = (rand() / (float)RAND_MAX - 0.5f); hk[i] = 0.25f + rand() / (float)RAND_MAX; }
  float *Eo, *Et, *kap, *part;
  hipMalloc(&Eo, ho.size() * 4); hipMalloc(&Et, ht.size() * 4); hipMalloc(&kap, hk.size() * 4); hipMalloc(&part, (size_t)160 * N * K * 4);
  hipMemcpy(Eo, ho.data(), ho.size() * 4, hipMemcpyHostToDevice); hipMemcpy(Et, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(kap, hk.data(), hk.size() * 4, hipMemcpyHostToDevice);
  run<0, 4, 8>("as shipped", N, D, K, Eo, Et, kap, part, 7);
  run<1, 4, 8>("no transform", N, D, K, Eo, Et, kap, part, 7);
  run<2, 4, 8>("transform, second product on the VALU", N, D, K, Eo, Et, kap, part, 7);
  run<3, 4, 8>("both products, 1-op transform", N, D, K, Eo, Et, kap, part, 7);
  run<4, 4, 8>("first product only", N, D, K, Eo, Et, kap, part, 7);
  run<5, 4, 8>("transform only", N, D, K, Eo, Et, kap, part, 7);
  run<6, 4, 8>("shipped + sched groups 1 MFMA : 4 VALU", N, D, K, Eo, Et, kap, part, 7);
  run<7, 4, 8>("shipped + sched groups 1 MFMA : 3 VALU", N, D, K, Eo, Et, kap, part, 7);
  run<8, 4, 8>("all first products, then by row r over the tiles", N, D, K, Eo, Et, kap, part, 7);
  run<9, 4, 8>("same + sched barrier between", N, D, K, Eo, Et, kap, part, 7);
  run<8, 8, 4>("same, 8 own tiles, 4 waves", N, D, K, Eo, Et, kap, part, 14);
  run<8, 2, 8>("same, 2 own tiles", N, D, K, Eo, Et, kap, part, 4);
  run<0, 4, 4>("as shipped, 4 waves / SIMD", N, D, K, Eo, Et, kap, part, 14);
  run<0, 4, 2>("as shipped, 2 waves / SIMD", N, D, K, Eo, Et, kap, part, 28);
  run<0, 4, 1>("as shipped, 1 wave / SIMD", N, D, K, Eo, Et, kap, part, 56);
  run<0, 2, 8>("as shipped, 2 own tiles", N, D, K, Eo, Et, kap, part, 4);
  run<0, 8, 4>("as shipped, 8 own tiles, 4 waves", N, D, K, Eo, Et, kap, part, 14);
  run<0, 8, 2>("as shipped, 8 own tiles, 2 waves", N, D, K, Eo, Et, kap, part, 28);
  return 0;
}
