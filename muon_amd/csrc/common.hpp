// Shared helpers for libmuon_amd.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "muon_amd.h"

void mu_set_error(const char* fmt, ...);

#define MU_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      mu_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,     \
                   __LINE__);                                                           \
      return MU_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

#define MU_CHECK_LAUNCH() MU_CHECK_HIP(hipGetLastError())

#define MU_REQUIRE(cond, msg)                        \
  do {                                               \
    if (!(cond)) {                                   \
      mu_set_error("%s: %s", __func__, msg);         \
      return MU_ERR_ARG;                             \
    }                                                \
  } while (0)

constexpr int kWave = 64;

// number of CUs of the current device (cached per device id)
int mu_num_cus();

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// tell the compiler a value is wave-uniform (it cannot see that threadIdx.x >> 6 is)
__device__ __forceinline__ int uniform32(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int64_t uniform64(int64_t x) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(x & 0xffffffffll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
  return ((int64_t)hi << 32) | (int64_t)(uint32_t)lo;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

template <typename T>
__device__ __forceinline__ T wave_sum_all(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;  // valid in every lane
}

// first index i in [lo, hi) with a[i] >= key (a ascending)
__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* a, int64_t lo, int64_t hi,
                                                   int64_t key) {
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// splitmix64: counter-based hashing for reproducible device-side random numbers
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// uniform in (0,1) from the top 24 bits
__host__ __device__ __forceinline__ float u01(uint64_t h) {
  return ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
}
