// TF-IDF kernels: replace the six scipy passes of
// /root/reference/muon/_atac/preproc.py:92-117 (two reductions, two diag x CSR
// SpGEMMs, a scalar multiply and a sparse log1p) by one reduction sweep and one
// fused scale pass.  HBM-bound: 8 B/nnz (sweep) + 12 B/nnz (scale).
#include <type_traits>
#include <utility>

#include "sweep.hpp"

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// ---------------------------------------------------------------------------------
// slab pointers
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_slab_ptr(int64_t n_rows, int64_t S,
                                                  const int64_t* __restrict__ indptr,
                                                  const int32_t* __restrict__ indices,
                                                  int64_t* __restrict__ sp, int64_t width = kSlab) {
  const int64_t total = n_rows * (S + 1);
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = id / (S + 1);
    const int64_t s = id - row * (S + 1);
    int64_t lo = indptr[row], hi = indptr[row + 1];
    if (s == S) {
      sp[id] = hi;
      continue;
    }
    const int64_t key = s * width;
    while (lo < hi) {
      int64_t mid = lo + ((hi - lo) >> 1);
      if ((int64_t)indices[mid] < key) lo = mid + 1; else hi = mid;
    }
    sp[id] = lo;
  }
}

int launch_slab_ptr(int64_t n_rows, int64_t n_cols, const int64_t* indptr, const int32_t* indices,
                    int64_t* sp, hipStream_t stream) {
  const int64_t S = num_slabs(n_cols);
  const int64_t total = n_rows * (S + 1);
  if (total == 0) return MU_OK;
  int64_t blocks = (total + 255) / 256;
  const int64_t cap = (int64_t)mu_num_cus() * 32;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_slab_ptr, dim3((unsigned)blocks), dim3(256), 0, stream, n_rows, S, indptr,
                     indices, sp);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

// ---------------------------------------------------------------------------------
// row sums + per-workgroup column partial sums (LDS bins), one pass over the nnz
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kSweepThreads, sizeof(T) == 4 ? 8 : 4) void k_row_col_sums(
    int64_t n_rows, int64_t n_cols, int64_t S, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const T* __restrict__ values,
    const int64_t* __restrict__ sp, double* __restrict__ rowsum, double* __restrict__ partial) {
  __shared__ double bins[kSlab];  // 64 KiB
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) sweep_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  __syncthreads();
  const int64_t r0 = s_r[0], r1 = s_r[1];
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;

  // Pointers are 32-bit offsets from the row block's first entry.  A wave takes its rows (r0 + wave,
  // + 16, ...) 64 at a time: lane l fetches the slab pointers of the strip's l-th row once, the visits
  // read them with v_readlane, and the strip's row sums go back with one read-modify-write per lane.
  // (r01/r02 loaded the pointers and read-modify-wrote the row sum inside every visit; taking both
  //  out of the visit is worth 5 % of the sweep at 1e6 x 200k, 13.1 -> 12.4 ms: the visits are not
  //  bound by that chain but by the bytes a wave keeps in flight - 2 KiB per visit at 64 VGPRs.)
  const int64_t wg_base = uniform64(indptr[r0 < n_rows ? r0 : n_rows]);
  const int32_t* __restrict__ ib = indices + wg_base;
  const T* __restrict__ vb = values + wg_base;
  for (int64_t s = 0; s < S; ++s) {
    for (int t = threadIdx.x; t < kSlab; t += kSweepThreads) bins[t] = 0.0;
    __syncthreads();
    const int32_t cbase = (int32_t)(s * kSlab);
    for (int64_t strip = r0 + wave; strip < r1; strip += (int64_t)kSweepWaves * 64) {
      const int64_t myrow = strip + (int64_t)kSweepWaves * lane;
      int lo_l = 0, hi_l = 0;
      if (myrow < r1) {
        lo_l = (int)(sp[myrow * (S + 1) + s] - wg_base);
        hi_l = (int)(sp[myrow * (S + 1) + s + 1] - wg_base);
      }
      const int64_t left = (r1 - strip + kSweepWaves - 1) / kSweepWaves;
      const int nrow = left < 64 ? (int)left : 64;  // wave-uniform
      double racc = 0.0;
      for (int l = 0; l < nrow; ++l) {
        const int lo = __builtin_amdgcn_readlane(lo_l, l), hi = __builtin_amdgcn_readlane(hi_l, l);
        double rs = 0.0;
        for (int p = lo + lane; p < hi; p += 256) {
          // four independent chunks in flight per wave
          int32_t c[4];
          T v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = p + 64 * u;
            const bool ok = q < hi;
            c[u] = ok ? ib[q] : -1;
            v[u] = ok ? vb[q] : (T)0;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (c[u] >= 0) {
              atomicAdd(&bins[c[u] - cbase], (double)v[u]);
              rs += (double)v[u];
            }
          }
        }
        rs = wave_sum(rs);  // (lane 0)
        const double tot = __shfl(rs, 0, 64);
        if (lane == l) racc = tot;
      }
      if (myrow < r1) {
        if (s == 0) rowsum[myrow] = racc; else rowsum[myrow] += racc;
      }
    }
    __syncthreads();
    const int64_t ncol_here = (n_cols - (int64_t)cbase) < kSlab ? (n_cols - (int64_t)cbase) : kSlab;
    double* dst = partial + (int64_t)g * n_cols + cbase;
    for (int t = threadIdx.x; t < ncol_here; t += kSweepThreads) dst[t] = bins[t];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------
// the same two sweeps, software pipelined (r04)
// ---------------------------------------------------------------------------------
// What the compiler made of the loops above: the predicated loads (`ok ? ib[q] : -1`) became branches with the
// value's conversion - and an `s_waitcnt vmcnt(0)` - inside, so the sum sweep's "four chunks in flight" were three
// full memory round trips per iteration; and in both sweeps a wave has ONE iteration in flight: it issues its
// loads, waits for all of them (on gfx9 the stores of the iteration before count in the same counter), works,
// and only then asks for more - with four (scale) or eight (sums) waves per SIMD the CU's requests in flight are
// a few tens of KiB, and the sweeps run at the memory latency, not at its bandwidth (4.15 / 5.2 of ~6.3 TB/s).
// Here the visits of a wave's strip of rows are walked as ONE flat sequence of iterations (a row's piece inside
// the slab, 64 CH entries at a time, then the next row's), every load is unconditional with a clamped position
// (no branch for the compiler to hide a wait in; a lane past the end re-reads the piece's last entry: the same
// line), and the loads of iteration i + 1 are issued before iteration i is worked on: two register sets, the
// loop unrolled by two, so the waits the compiler inserts are counted ones.  Same arithmetic in the same order,
// entry by entry and row by row: bit-identical row sums and values.
// The loads of the pipelined f32 sweeps are issued and awaited from asm: left to the compiler, the two register sets
// of the unrolled loop came back with `s_waitcnt vmcnt(1)` in front of every other set of loads (a write-after-write
// wait on registers whose loads had long been consumed), i.e. with half of the overlap.  The compiler does not know
// that these registers are pending: `pipe_wait` takes them as read-write operands, so every use is ordered behind
// it, and the ISA was checked for copies between a load and its wait (none).  Counts: a set is 2 CH loads, issued
// in chunk order; the wait for chunk u of a set allows the loads issued after it - the rest of its own set and the
// whole next set - to be outstanding: 2 (CH - 1 - u) + 2 CH.  (Stores of the scale sweep issued in between only
// make the wait stricter.)  Offsets are 32-bit byte offsets from the row block's first entry (host: < 2^30 entries).
__device__ __forceinline__ void pipe_load(const int32_t* ib, const float* vb, unsigned off, int32_t& c, float& v) {
  asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %4"
               : "=&v"(c), "=&v"(v)
               : "v"(off), "s"(ib), "s"(vb)
               : "memory");
}
template <int N>
__device__ __forceinline__ void pipe_wait(int32_t& c, float& v) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(c), "+v"(v) : "n"(N) : "memory");
}
// The walk ends with one set of loads that nobody reads (issued for an iteration that does not exist): its registers
// stay operands of this wait, so the compiler cannot hand them to anything else while the loads are in flight.
template <typename A, typename B>
__device__ __forceinline__ void pipe_drain(A (&c0)[4], B (&v0)[4], A (&c1)[4], B (&v1)[4]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(v0[0]), "+v"(v0[1]), "+v"(v0[2]), "+v"(v0[3]),
                 "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]), "+v"(v1[0]), "+v"(v1[1]), "+v"(v1[2]), "+v"(v1[3])
               :
               : "memory");
}

struct FlatWalk {
  int l, pb, hi, nrow, step;
  // the next iteration: false when the strip is through (wave-uniform: every member lives in scalar registers)
  __device__ __forceinline__ bool advance(int lo_l, int hi_l) {
    pb += step;
    while (pb >= hi) {
      if (++l >= nrow) return false;
      pb = __builtin_amdgcn_readlane(lo_l, l);
      hi = __builtin_amdgcn_readlane(hi_l, l);
    }
    return true;
  }
};

// ABL (timing ablations, wrong results; tune "tfidf_abl"): 1 no LDS atomics, 2 f32 atomics
// M: bins of M x 8192 columns (M = 2: 128 KiB, one workgroup per CU - pieces of a row twice as long: 14.3 -> 13.1 ms
// at 1e6 x 200k now that a wave keeps two iterations in flight; r04's first try of M = 2, on the unpipelined kernel,
// lost 17 %).
// (r04 also built the slab pointer search INTO this sweep - the lane that owns a row finds the ends of its next
// pieces from the end of the last one - to drop k_slab_ptr's 3.0 ms: the dependent loads, un-overlapped inside a
// wave, cost the sweep 4.3 ms.  Deleted; what the search costs is its 64-byte sectors, wherever it runs.)
template <typename T, int CH, int ABL = 0, int M = 1>
__global__ __launch_bounds__(kSweepThreads, (sizeof(T) == 4 && M == 1) ? 8 : 4) void k_row_col_sums_pipe(  // (CH = 4: pipe_drain)
    int64_t n_rows, int64_t n_cols, int64_t S, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const T* __restrict__ values,
    const int64_t* __restrict__ sp, double* __restrict__ rowsum, double* __restrict__ partial) {
  __shared__ double bins[kSlab * M];  // 64 KiB x M
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) sweep_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  __syncthreads();
  const int64_t r0 = uniform64(s_r[0]), r1 = uniform64(s_r[1]);
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t wg_base = uniform64(indptr[r0 < n_rows ? r0 : n_rows]);
  const int32_t* __restrict__ ib = indices + wg_base;
  const T* __restrict__ vb = values + wg_base;
  constexpr bool kAsm = std::is_same<T, float>::value;
  for (int64_t s = 0; s < S; s += M) {
    for (int t = threadIdx.x; t < kSlab * M; t += kSweepThreads) bins[t] = 0.0;
    __syncthreads();
    const int32_t cbase = (int32_t)(s * kSlab);
    const int64_t s_hi = s + M < S ? s + M : S;
    for (int64_t strip = r0 + wave; strip < r1; strip += (int64_t)kSweepWaves * 64) {
      const int64_t myrow = strip + (int64_t)kSweepWaves * lane;
      int lo_l = 0, hi_l = 0;
      if (myrow < r1) {
        lo_l = (int)(sp[myrow * (S + 1) + s] - wg_base);
        hi_l = (int)(sp[myrow * (S + 1) + s_hi] - wg_base);
      }
      const int64_t left = (r1 - strip + kSweepWaves - 1) / kSweepWaves;
      FlatWalk w{-1, 0, 0, left < 64 ? (int)left : 64, 64 * CH};
      double racc = 0.0, rs = 0.0;
      int32_t ca[CH], cb[CH];
      T va[CH], vb_[CH];
      auto load = [&](int32_t (&c)[CH], T (&v)[CH], int pb, int hi) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          int q = pb + lane + 64 * u;
          q = q < hi ? q : hi - 1;
          if constexpr (kAsm) {
            pipe_load(ib, vb, (unsigned)q * 4u, c[u], v[u]);
          } else {
            c[u] = ib[q];
            v[u] = vb[q];
          }
        }
      };
      auto work = [&](int32_t (&c)[CH], T (&v)[CH], int pb, int hi, int l, bool row_ends) {
        // no branch around a chunk: a register that was loaded but not read on some path makes the compiler wait
        // for it - with `vmcnt(0)`, i.e. for the loads just issued - before the register is written again.  A lane
        // past the end adds 0.0 to a bin of its own (no conflict, no effect)
        static_for<CH>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          if constexpr (kAsm) pipe_wait<2 * (CH - 1 - u) + 2 * CH>(c[u], v[u]);
        });
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const bool ok = pb + lane + 64 * u < hi;
          const double x = ok ? (double)v[u] : 0.0;
          if constexpr (ABL == 0) atomicAdd(&bins[ok ? c[u] - cbase : lane], x);
          if constexpr (ABL == 1) rs += (double)(c[u] & 1);
          if constexpr (ABL == 2) atomicAdd(reinterpret_cast<float*>(bins) + (ok ? c[u] - cbase : lane), (float)x);
          rs += x;
        }
        if (row_ends) {  // (uniform)
          const double tot = __shfl(wave_sum(rs), 0, 64);
          if (lane == l) racc = tot;
          rs = 0.0;
        }
      };
      if (w.advance(lo_l, hi_l)) {
        int a_pb = w.pb, a_hi = w.hi, a_l = w.l, b_pb, b_hi, b_l;
        load(ca, va, a_pb, a_hi);
        for (;;) {
          bool more = w.advance(lo_l, hi_l);
          b_pb = more ? w.pb : a_pb, b_hi = more ? w.hi : a_hi, b_l = w.l;  // (nothing left: a harmless re-read)
          load(cb, vb_, b_pb, b_hi);
          work(ca, va, a_pb, a_hi, a_l, b_l != a_l);
          if (!more) break;
          more = w.advance(lo_l, hi_l);
          a_pb = more ? w.pb : b_pb, a_hi = more ? w.hi : b_hi, a_l = w.l;
          load(ca, va, a_pb, a_hi);
          work(cb, vb_, b_pb, b_hi, b_l, a_l != b_l);
          if (!more) break;
        }
        if constexpr (kAsm) pipe_drain(ca, va, cb, vb_);
      }
      if (myrow < r1) {
        if (s == 0) rowsum[myrow] = racc; else rowsum[myrow] += racc;
      }
    }
    __syncthreads();
    const int64_t ncol_here = (n_cols - (int64_t)cbase) < kSlab * M ? (n_cols - (int64_t)cbase) : kSlab * M;
    double* dst = partial + (int64_t)g * n_cols + cbase;
    for (int t = threadIdx.x; t < ncol_here; t += kSweepThreads) dst[t] = bins[t];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_reduce_partials(int64_t n_cols, int G,
                                                         const double* __restrict__ partial,
                                                         double* __restrict__ colsum) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_cols) return;
  double acc = 0.0;
  for (int g = 0; g < G; ++g) acc += partial[(int64_t)g * n_cols + j];  // fixed order
  colsum[j] = acc;
}

// ---------------------------------------------------------------------------------
// idf and the fused scale pass
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_idf(int64_t n_cols, double n_obs,
                                             const double* __restrict__ colsum, int flags,
                                             T* __restrict__ idf) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_cols) return;
  T v = (T)n_obs / (T)colsum[j];  // preproc.py:106
  if (flags & MU_TFIDF_LOG_IDF) v = log1p(v);  // :107-108
  idf[j] = v;
}

// log1p for the f32 path.  libm's log1pf is ~100 VALU instructions and made the scale pass
// VALU-bound (4.6 ms for 7.8e8 entries, 2 TB/s).  When every lane of the wave holds a value in
// [0.25, 1e30) - the normal case: t = count / row sum * scale_factor - the hardware log2 of
// u = 1 + t times ln 2, plus the first-order correction for the rounding of u, is accurate to
// < 3e-7 relative; anything else (tiny, negative, NaN, inf) takes libm for the whole wave, so
// special values behave exactly as before.
__device__ __forceinline__ float log1p_wave(float t) {
  const bool easy = (t >= 0.25f) && (t < 1e30f);
  if (__all(easy)) {
    const float u = 1.0f + t;
    float r = __builtin_amdgcn_logf(u) * 0.69314718055994531f;  // v_log_f32 = log2
    r += (t - (u - 1.0f)) * __builtin_amdgcn_rcpf(u);
    return r;
  }
  return log1pf(t);
}
__device__ __forceinline__ double log1p_wave(double t) { return log1p(t); }

template <typename T>
__global__ __launch_bounds__(256) void k_tfidf_scale(
    int64_t n_rows, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const T* __restrict__ values, const double* __restrict__ rowsum, const T* __restrict__ idf,
    T scale, int use_scale, int flags, T* __restrict__ out, unsigned long long* zero_count) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  unsigned int zeros = 0;
  for (int64_t row = wave0; row < n_rows; row += n_waves) {
    const int64_t lo = indptr[row], hi = indptr[row + 1];
    const T inv = (T)1 / (T)rowsum[row];  // preproc.py:94  1.0 / n_peaks
    for (int64_t p0 = lo + lane; p0 < hi; p0 += 256) {
      // four independent 64-entry chunks in flight per wave (the row is streamed once: the only
      // reuse is the idf gather, which stays in L2)
      int32_t c[4];
      T x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t p = p0 + 64 * u;
        const bool ok = p < hi;
        c[u] = ok ? indices[p] : 0;
        x[u] = ok ? values[p] : (T)0;
      }
      T w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w[u] = idf[c[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t p = p0 + 64 * u;
        if (p < hi) {
          T t = inv * x[u];                              // :96  D @ counts
          if (use_scale) t = t * scale;                  // :101-102
          if (flags & MU_TFIDF_LOG_TF) t = log1p_wave(t);     // :103-104
          t = t * w[u];                                       // :110-112  tf @ diag(idf)
          if (flags & MU_TFIDF_LOG_TFIDF) t = log1p_wave(t);  // :116-117
          out[p] = t;
          zeros += (t == (T)0) ? 1u : 0u;
        }
      }
    }
  }
  if (zero_count) {
    zeros = wave_sum(zeros);
    if (lane == 0 && zeros) atomicAdd(zero_count, (unsigned long long)zeros);
  }
}

// Same arithmetic as k_tfidf_scale, walked like the sum pass: a workgroup owns a contiguous row
// range and sweeps the column slabs with the slab of idf in LDS.  k_tfidf_scale gathers idf[col]
// per lane from L2 - 64 different lines per wave instruction, which the texture addresser serves a
// few lines per clock: 2.2 TB/s against the 3.9 TB/s of the sum pass that reads the same arrays.
template <typename T>
__global__ __launch_bounds__(kSweepThreads, sizeof(T) == 4 ? 8 : 4) void k_tfidf_scale_sweep(
    int64_t n_rows, int64_t n_cols, int64_t S, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const T* __restrict__ values,
    const int64_t* __restrict__ sp, const double* __restrict__ rowsum, const T* __restrict__ idf,
    T scale, int use_scale, int flags, T* __restrict__ out, unsigned long long* zero_count) {
  __shared__ T lidf[kSlab];  // 32 / 64 KiB
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) sweep_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  __syncthreads();
  const int64_t r0 = s_r[0], r1 = s_r[1];
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  unsigned int zeros = 0;
  // (strips of 64 rows with the slab pointers and 1 / row sum in lanes, as in the sum sweep)
  const int64_t wg_base = uniform64(indptr[r0 < n_rows ? r0 : n_rows]);
  const int32_t* __restrict__ ib = indices + wg_base;
  const T* __restrict__ vb = values + wg_base;
  T* __restrict__ ob = out + wg_base;
  for (int64_t s = 0; s < S; ++s) {
    const int32_t cbase = (int32_t)(s * kSlab);
    const int64_t ncol_here = (n_cols - (int64_t)cbase) < kSlab ? (n_cols - (int64_t)cbase) : kSlab;
    for (int t = threadIdx.x; t < kSlab; t += kSweepThreads) lidf[t] = t < ncol_here ? idf[cbase + t] : (T)0;
    __syncthreads();
    for (int64_t strip = r0 + wave; strip < r1; strip += (int64_t)kSweepWaves * 64) {
      const int64_t myrow = strip + (int64_t)kSweepWaves * lane;
      int lo_l = 0, hi_l = 0;
      T inv_l = (T)0;
      if (myrow < r1) {
        lo_l = (int)(sp[myrow * (S + 1) + s] - wg_base);
        hi_l = (int)(sp[myrow * (S + 1) + s + 1] - wg_base);
        inv_l = (T)1 / (T)rowsum[myrow];  // preproc.py:94  1.0 / n_peaks
      }
      const int64_t left = (r1 - strip + kSweepWaves - 1) / kSweepWaves;
      const int nrow = left < 64 ? (int)left : 64;  // wave-uniform
      for (int l = 0; l < nrow; ++l) {
        const int lo = __builtin_amdgcn_readlane(lo_l, l), hi = __builtin_amdgcn_readlane(hi_l, l);
        if (lo >= hi) continue;
        const T inv = __shfl(inv_l, l, 64);
        for (int p0 = lo + lane; p0 < hi; p0 += 256) {
          int32_t c[4];
          T x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int p = p0 + 64 * u;
            const bool ok = p < hi;
            c[u] = ok ? ib[p] : cbase;
            x[u] = ok ? vb[p] : (T)0;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int p = p0 + 64 * u;
            if (p < hi) {
              T t = inv * x[u];                                   // :96  D @ counts
              if (use_scale) t = t * scale;                       // :101-102
              if (flags & MU_TFIDF_LOG_TF) t = log1p_wave(t);     // :103-104
              t = t * lidf[c[u] - cbase];                         // :110-112  tf @ diag(idf)
              if (flags & MU_TFIDF_LOG_TFIDF) t = log1p_wave(t);  // :116-117
              ob[p] = t;
              zeros += (t == (T)0) ? 1u : 0u;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (zero_count) {
    zeros = wave_sum(zeros);
    if (lane == 0 && zeros) atomicAdd(zero_count, (unsigned long long)zeros);
  }
}

// The same sweep with M times wider slabs (f32: M = 4 -> 32 768 columns of idf = 128 KiB of LDS, one workgroup
// per CU at up to 128 registers): a visit - the piece of one row inside one slab - is M times longer, so the
// per-visit work (pointer lanes, loop set-up, the drain of the loads at its end) is paid M times less often and
// CH chunks of 64 entries are in flight per wave.  Measured at 1e6 x 200 000 (scripts/tfidf_probe.py): M = 1 / 2 /
// 4: 18.2 / 16.9 / 14.4 ms (4.15 / 4.5 / 5.2 TB/s); 8 chunks in flight instead of 4: no change.  The SUM sweep
// does not gain from the same change (f64 bins: M = 2 = 128 KiB, 14.2 -> 16.6 ms: its LDS atomics want the waves).  The slab pointers are the ones of the 8192-column slabs,
// read with stride M.  Same arithmetic, entry by entry: bit-identical results.
template <int M, int CH>
__global__ __launch_bounds__(kSweepThreads, 4) void k_tfidf_scale_sweep_wide(
    int64_t n_rows, int64_t n_cols, int64_t S, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const float* __restrict__ values,
    const int64_t* __restrict__ sp, const double* __restrict__ rowsum, const float* __restrict__ idf,
    float scale, int use_scale, int flags, float* __restrict__ out, unsigned long long* zero_count) {
  __shared__ float lidf[kSlab * M];
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) sweep_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  __syncthreads();
  const int64_t r0 = s_r[0], r1 = s_r[1];
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  unsigned int zeros = 0;
  const int64_t wg_base = uniform64(indptr[r0 < n_rows ? r0 : n_rows]);
  const int32_t* __restrict__ ib = indices + wg_base;
  const float* __restrict__ vb = values + wg_base;
  float* __restrict__ ob = out + wg_base;
  const int64_t SW = (S + M - 1) / M;
  for (int64_t sw = 0; sw < SW; ++sw) {
    const int64_t s_lo = sw * M, s_hi = (sw + 1) * M < S ? (sw + 1) * M : S;
    const int32_t cbase = (int32_t)(s_lo * kSlab);
    const int64_t ncol_here = (n_cols - (int64_t)cbase) < (int64_t)kSlab * M ? (n_cols - (int64_t)cbase) : (int64_t)kSlab * M;
    for (int t = threadIdx.x; t < kSlab * M; t += kSweepThreads) lidf[t] = t < ncol_here ? idf[cbase + t] : 0.f;
    __syncthreads();
    for (int64_t strip = r0 + wave; strip < r1; strip += (int64_t)kSweepWaves * 64) {
      const int64_t myrow = strip + (int64_t)kSweepWaves * lane;
      int lo_l = 0, hi_l = 0;
      float inv_l = 0.f;
      if (myrow < r1) {
        lo_l = (int)(sp[myrow * (S + 1) + s_lo] - wg_base);
        hi_l = (int)(sp[myrow * (S + 1) + s_hi] - wg_base);
        inv_l = 1.0f / (float)rowsum[myrow];  // preproc.py:94  1.0 / n_peaks
      }
      const int64_t left = (r1 - strip + kSweepWaves - 1) / kSweepWaves;
      const int nrow = left < 64 ? (int)left : 64;  // wave-uniform
      for (int l = 0; l < nrow; ++l) {
        const int lo = __builtin_amdgcn_readlane(lo_l, l), hi = __builtin_amdgcn_readlane(hi_l, l);
        if (lo >= hi) continue;
        const float inv = __shfl(inv_l, l, 64);
        for (int p0 = lo + lane; p0 < hi; p0 += 64 * CH) {
          int32_t c[CH];
          float x[CH];
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const int p = p0 + 64 * u;
            const bool ok = p < hi;
            c[u] = ok ? ib[p] : cbase;
            x[u] = ok ? vb[p] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const int p = p0 + 64 * u;
            if (p < hi) {
              float t = inv * x[u];                               // :96  D @ counts
              if (use_scale) t = t * scale;                       // :101-102
              if (flags & MU_TFIDF_LOG_TF) t = log1p_wave(t);     // :103-104
              t = t * lidf[c[u] - cbase];                         // :110-112  tf @ diag(idf)
              if (flags & MU_TFIDF_LOG_TFIDF) t = log1p_wave(t);  // :116-117
              ob[p] = t;
              zeros += (t == 0.f) ? 1u : 0u;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (zero_count) {
    zeros = wave_sum(zeros);
    if (lane == 0 && zeros) atomicAdd(zero_count, (unsigned long long)zeros);
  }
}

// k_tfidf_scale_sweep_wide, software pipelined (see k_row_col_sums_pipe).  A lane past the end of a piece holds
// the piece's LAST entry and stores its value again: same address, same bits - no branch around the store either.
// STREAM (r05): the sweep also writes the ROW STREAM of the result - the (column, value) pairs of every row, 8 bytes each,
// contiguous from pair row_dst[row] of `ent`, i.e. in the launch order of the SpMM that lsi() runs next (csrc/spmm_win.hip)
// - while it has every entry in registers.  lsi's streaming copy of X (8 bytes read + 8 written per entry, next to the
// transposition's fill) disappears, and the transposition reads rows as contiguous pairs (csrc/tpack4.hip).
template <int M, int CH, int ABL = 0, bool STREAM = false>  // ABL (timing ablations, tune "tfidf_abl"): 3 no arithmetic, 4 no stores
__global__ __launch_bounds__(kSweepThreads, 4) void k_tfidf_scale_sweep_pipe(
    int64_t n_rows, int64_t n_cols, int64_t S, const int64_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const float* __restrict__ values,
    const int64_t* __restrict__ sp, const double* __restrict__ rowsum, const float* __restrict__ idf,
    float scale, int use_scale, int flags, float* __restrict__ out, unsigned long long* zero_count,
    const int64_t* __restrict__ row_dst = nullptr, unsigned long long* __restrict__ ent = nullptr) {
  __shared__ float lidf[kSlab * M];
  __shared__ int64_t s_r[2];
  const int g = blockIdx.x, G = gridDim.x;
  if (threadIdx.x == 0) sweep_row_range(indptr, n_rows, g, G, s_r[0], s_r[1]);
  __syncthreads();
  const int64_t r0 = uniform64(s_r[0]), r1 = uniform64(s_r[1]);
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  unsigned int zeros = 0;
  const int64_t wg_base = uniform64(indptr[r0 < n_rows ? r0 : n_rows]);
  const int32_t* __restrict__ ib = indices + wg_base;
  const float* __restrict__ vb = values + wg_base;
  float* __restrict__ ob = out + wg_base;
  const int64_t SW = (S + M - 1) / M;
  for (int64_t sw = 0; sw < SW; ++sw) {
    const int64_t s_lo = sw * M, s_hi = (sw + 1) * M < S ? (sw + 1) * M : S;
    const int32_t cbase = (int32_t)(s_lo * kSlab);
    const int64_t ncol_here = (n_cols - (int64_t)cbase) < (int64_t)kSlab * M ? (n_cols - (int64_t)cbase) : (int64_t)kSlab * M;
    for (int t = threadIdx.x; t < kSlab * M; t += kSweepThreads) lidf[t] = t < ncol_here ? idf[cbase + t] : 0.f;
    __syncthreads();
    for (int64_t strip = r0 + wave; strip < r1; strip += (int64_t)kSweepWaves * 64) {
      const int64_t myrow = strip + (int64_t)kSweepWaves * lane;
      int lo_l = 0, hi_l = 0;
      float inv_l = 0.f;
      int dlo_l = 0, dhi_l = 0;  // STREAM: pair index in `ent` of the entry at offset 0 from wg_base, for this lane's row
      if (myrow < r1) {
        lo_l = (int)(sp[myrow * (S + 1) + s_lo] - wg_base);
        hi_l = (int)(sp[myrow * (S + 1) + s_hi] - wg_base);
        inv_l = 1.0f / (float)rowsum[myrow];  // preproc.py:94  1.0 / n_peaks
        if constexpr (STREAM) {
          const int64_t d = row_dst[myrow] - (indptr[myrow] - wg_base);
          dlo_l = (int)(unsigned)(d & 0xffffffffll);
          dhi_l = (int)(d >> 32);
        }
      }
      const int64_t left = (r1 - strip + kSweepWaves - 1) / kSweepWaves;
      FlatWalk w{-1, 0, 0, left < 64 ? (int)left : 64, 64 * CH};
      int32_t ca[CH], cb[CH];
      float xa[CH], xb[CH];
      auto load = [&](int32_t (&c)[CH], float (&x)[CH], int pb, int hi) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          int q = pb + lane + 64 * u;
          q = q < hi ? q : hi - 1;
          pipe_load(ib, vb, (unsigned)q * 4u, c[u], x[u]);
        }
      };
      auto work = [&](int32_t (&c)[CH], float (&x)[CH], int pb, int hi, int l) {
        const float inv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv_l), l));
        unsigned long long* eb = nullptr;
        if constexpr (STREAM) {
          const unsigned dlo = (unsigned)__builtin_amdgcn_readlane(dlo_l, l);
          const int dhi = __builtin_amdgcn_readlane(dhi_l, l);
          eb = ent + (((int64_t)dhi << 32) | (int64_t)dlo);
        }
        static_for<CH>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          pipe_wait<2 * (CH - 1 - u) + 2 * CH>(c[u], x[u]);
        });
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          // (no branch around a chunk, see k_row_col_sums_pipe: a chunk with nothing in it stores the last value again)
          const int p = pb + lane + 64 * u;
          const int q = p < hi ? p : hi - 1;
          float t = inv * x[u];                               // :96  D @ counts
          if constexpr (ABL != 3) {
            if (use_scale) t = t * scale;                       // :101-102
            if (flags & MU_TFIDF_LOG_TF) t = log1p_wave(t);     // :103-104
            t = t * lidf[c[u] - cbase];                         // :110-112  tf @ diag(idf)
            if (flags & MU_TFIDF_LOG_TFIDF) t = log1p_wave(t);  // :116-117
          } else {
            t += (float)c[u];
          }
          if constexpr (ABL != 4) ob[q] = t;
          if constexpr (STREAM)  // (a lane past the piece's end stores the piece's last pair again: same address, same bits)
            eb[q] = (unsigned long long)(unsigned)c[u] | ((unsigned long long)__builtin_bit_cast(unsigned, t) << 32);
          zeros += (p < hi && t == 0.f) ? 1u : 0u;
        }
      };
      if (w.advance(lo_l, hi_l)) {
        int a_pb = w.pb, a_hi = w.hi, a_l = w.l, b_pb, b_hi, b_l;
        load(ca, xa, a_pb, a_hi);
        for (;;) {
          bool more = w.advance(lo_l, hi_l);
          b_pb = more ? w.pb : a_pb, b_hi = more ? w.hi : a_hi, b_l = more ? w.l : a_l;
          load(cb, xb, b_pb, b_hi);
          work(ca, xa, a_pb, a_hi, a_l);
          if (!more) break;
          more = w.advance(lo_l, hi_l);
          a_pb = more ? w.pb : b_pb, a_hi = more ? w.hi : b_hi, a_l = more ? w.l : b_l;
          load(ca, xa, a_pb, a_hi);
          work(cb, xb, b_pb, b_hi, b_l);
          if (!more) break;
        }
        pipe_drain(ca, xa, cb, xb);
      }
    }
    __syncthreads();
  }
  if (zero_count) {
    zeros = wave_sum(zeros);
    if (lane == 0 && zeros) atomicAdd(zero_count, (unsigned long long)zeros);
  }
}

// ---------------------------------------------------------------------------------
// explicit-zero compaction, scan, fill
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_count_nonzero(int64_t n_rows,
                                                       const int64_t* __restrict__ indptr,
                                                       const T* __restrict__ values,
                                                       int64_t* __restrict__ row_nnz) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t row = wave0; row < n_rows; row += n_waves) {
    const int64_t lo = indptr[row], hi = indptr[row + 1];
    int cnt = 0;
    for (int64_t p = lo + lane; p < hi; p += 64) cnt += (values[p] != (T)0) ? 1 : 0;  // NaN kept
    cnt = wave_sum(cnt);
    if (lane == 0) row_nnz[row] = cnt;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_compact_nonzero(
    int64_t n_rows, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const T* __restrict__ values, const int64_t* __restrict__ new_indptr,
    int32_t* __restrict__ new_indices, T* __restrict__ new_values) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t row = wave0; row < n_rows; row += n_waves) {
    const int64_t lo = indptr[row], hi = indptr[row + 1];
    int64_t dst = new_indptr[row];
    for (int64_t p0 = lo; p0 < hi; p0 += 64) {
      const int64_t p = p0 + lane;
      const bool in = p < hi;
      const T v = in ? values[p] : (T)0;
      const bool keep = in && (v != (T)0);
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        new_indices[dst + rank] = indices[p];
        new_values[dst + rank] = v;
      }
      dst += __popcll(m);
    }
  }
}

// out[0] = 0, out[i+1] = out[i] + in[i]; one workgroup, three phases
__global__ __launch_bounds__(1024) void k_exclusive_scan_i64(int64_t n, const int64_t* __restrict__ in,
                                                             int64_t* __restrict__ out) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t chunk = (n + 1023) / 1024;
  const int64_t b = (int64_t)t * chunk;
  const int64_t e = (b + chunk < n) ? (b + chunk) : n;
  int64_t s = 0;
  for (int64_t i = b; i < e; ++i) s += in[i];
  part[t] = s;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 1024 partials
  for (int off = 1; off < 1024; off <<= 1) {
    int64_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int64_t run = (t == 0) ? 0 : part[t - 1];
  if (t == 0) out[0] = 0;
  for (int64_t i = b; i < e; ++i) {
    run += in[i];
    out[i + 1] = run;
  }
}

// Large n: the same scan over G chunks with no scratch memory.  Pass 1 leaves the sum of chunk g in
// out[e_g] (the chunk's last output slot), pass 2 turns these G values into their inclusive scan -
// which is exactly what out[e_g] has to hold in the end - and pass 3 fills the rest of every chunk
// from out[e_{g-1}].  (in and out must not overlap.)
constexpr int kScanChunks = 256;
__device__ __forceinline__ void scan_chunk_range(int64_t n, int g, int64_t& b, int64_t& e) {
  const int64_t chunk = (n + kScanChunks - 1) / kScanChunks;
  b = (int64_t)g * chunk;
  if (b > n) b = n;
  e = (b + chunk < n) ? (b + chunk) : n;
}
// block-wide sum / exclusive scan helpers over 1024 per-thread values
__device__ __forceinline__ int64_t block_exclusive_scan_1024(int64_t v, int64_t* part, int64_t& total) {
  const int t = threadIdx.x;
  part[t] = v;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int64_t u = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += u;
    __syncthreads();
  }
  total = part[1023];
  return part[t] - v;
}
__global__ __launch_bounds__(1024) void k_scan_chunk_sums(int64_t n, const int64_t* __restrict__ in,
                                                          int64_t* __restrict__ out) {
  __shared__ int64_t part[1024];
  int64_t b, e;
  scan_chunk_range(n, blockIdx.x, b, e);
  int64_t s = 0;
  for (int64_t i = b + threadIdx.x; i < e; i += 1024) s += in[i];
  int64_t total;
  block_exclusive_scan_1024(s, part, total);
  if (threadIdx.x == 0 && e > b) out[e] = total;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
}
__global__ __launch_bounds__(1024) void k_scan_chunk_offsets(int64_t n, int64_t* __restrict__ out) {
  __shared__ int64_t part[1024];
  int64_t b = 0, e = 0;
  const bool has = threadIdx.x < kScanChunks;
  if (has) scan_chunk_range(n, threadIdx.x, b, e);
  const int64_t v = (has && e > b) ? out[e] : 0;
  int64_t total;
  const int64_t excl = block_exclusive_scan_1024(v, part, total);
  if (has && e > b) out[e] = excl + v;
}
__global__ __launch_bounds__(1024) void k_scan_chunk_fill(int64_t n, const int64_t* __restrict__ in,
                                                          int64_t* __restrict__ out) {
  __shared__ int64_t part[1024];
  int64_t b, e;
  scan_chunk_range(n, blockIdx.x, b, e);
  if (e <= b) return;
  const int64_t base = (b == 0) ? 0 : out[b];  // = out[e_{g-1}]
  // thread t owns a contiguous piece of the chunk
  const int64_t len = e - b, per = (len + 1023) / 1024;
  const int64_t tb = b + (int64_t)threadIdx.x * per;
  const int64_t te = (tb + per < e) ? (tb + per) : e;
  int64_t s = 0;
  for (int64_t i = tb; i < te; ++i) s += in[i];
  int64_t total;
  int64_t run = base + block_exclusive_scan_1024(s, part, total);
  for (int64_t i = tb; i < te; ++i) {
    run += in[i];
    if (i + 1 < e) out[i + 1] = run;  // out[e] is already final
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_binarize(int64_t n, T* __restrict__ v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    if (v[i] != (T)0) v[i] = (T)1;  // preproc.py:150
}

static inline unsigned grid_for_rows(int64_t n_rows) {
  // one wave per row, 4 waves per block, capped at 16 blocks per CU (grid-stride beyond)
  int64_t blocks = (n_rows + 3) / 4;
  const int64_t cap = (int64_t)mu_num_cus() * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

extern "C" {

size_t mu_csr_row_col_sums_worksize(int64_t n_rows, int64_t n_cols) {
  const int64_t S = num_slabs(n_cols);
  const int G = sweep_grid();
  return (size_t)(n_rows * (S + 1)) * sizeof(int64_t) + (size_t)G * (size_t)n_cols * sizeof(double) +
         256;
}

static int row_col_sums_impl(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                             const int32_t* d_indices, const void* d_values, double* d_rowsum, double* d_colsum,
                             void* d_work, size_t work_bytes, const int64_t* d_slab_ptr, void* stream);

int mu_csr_row_col_sums(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                        const int32_t* d_indices, const void* d_values, double* d_rowsum,
                        double* d_colsum, void* d_work, size_t work_bytes, void* stream) {
  return row_col_sums_impl(dtype, n_rows, n_cols, d_indptr, d_indices, d_values, d_rowsum, d_colsum, d_work, work_bytes,
                           nullptr, stream);
}

/* r05: the slab pointers (first entry of every row at or behind every 8192-column boundary: 26 binary searches per row
 * at 200 000 columns, 3.0 ms of a 1e6-row step) depend on the index arrays alone, which do not change between ingest,
 * binarize, tfidf and lsi: mu_csr_slab_ptr builds the table once where the device CSR is made (d_sp: int64[n_rows *
 * (ceil(n_cols / 8192) + 1)]), the _sp entries read it instead of searching (d_slab_ptr == NULL: search, as before). */
/* the same table for slabs of `width` columns (d_sp: int64[n_rows * (ceil(n_cols / width) + 1)]): the operand layouts that
 * cut rows into column slabs of their own (the sliced-ELL operand of MOFA's sparse views: 1024 / 512 columns) read the
 * entries per (row, slab) off it instead of histogramming every entry */
int mu_csr_slab_ptr_width(int64_t n_rows, int64_t n_cols, int64_t width, const int64_t* d_indptr,
                          const int32_t* d_indices, int64_t* d_sp, void* stream) {
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0 && width > 0, "bad shape");
  if (n_rows == 0 || n_cols == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_sp, "null pointer");
  const int64_t S = (n_cols + width - 1) / width;
  const int64_t total = n_rows * (S + 1);
  int64_t blocks = (total + 255) / 256;
  const int64_t cap = (int64_t)mu_num_cus() * 32;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_slab_ptr, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_rows, S, d_indptr,
                     d_indices, d_sp, width);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_csr_slab_ptr(int64_t n_rows, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices, int64_t* d_sp,
                    void* stream) {
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative shape");
  if (n_rows == 0 || n_cols == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_sp, "null pointer");
  return launch_slab_ptr(n_rows, n_cols, d_indptr, d_indices, d_sp, (hipStream_t)stream);
}

int mu_csr_row_col_sums_sp(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                           const int32_t* d_indices, const void* d_values, double* d_rowsum, double* d_colsum,
                           void* d_work, size_t work_bytes, const int64_t* d_slab_ptr, void* stream) {
  return row_col_sums_impl(dtype, n_rows, n_cols, d_indptr, d_indices, d_values, d_rowsum, d_colsum, d_work, work_bytes,
                           d_slab_ptr, stream);
}

}  // extern "C"

static int row_col_sums_impl(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                             const int32_t* d_indices, const void* d_values, double* d_rowsum, double* d_colsum,
                             void* d_work, size_t work_bytes, const int64_t* d_slab_ptr, void* stream) {
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative shape");
  MU_REQUIRE(d_indptr && d_rowsum && d_colsum, "null pointer");
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(work_bytes >= mu_csr_row_col_sums_worksize(n_rows, n_cols) && d_work,
             "work buffer too small");
  hipStream_t st = (hipStream_t)stream;
  if (n_cols == 0 || n_rows == 0) {
    if (n_cols) MU_CHECK_HIP(hipMemsetAsync(d_colsum, 0, sizeof(double) * n_cols, st));
    if (n_rows) MU_CHECK_HIP(hipMemsetAsync(d_rowsum, 0, sizeof(double) * n_rows, st));
    return MU_OK;
  }
  const int64_t S = num_slabs(n_cols);
  const int G = sweep_grid();
  const int64_t* sp = d_slab_ptr ? d_slab_ptr : (const int64_t*)d_work;
  size_t off = ((size_t)(n_rows * (S + 1)) * sizeof(int64_t) + 255) & ~(size_t)255;
  double* partial = (double*)((char*)d_work + off);
  // software-pipelined walk (r04); tune "tfidf_pipe" = 1: the kernels of before, for comparison.  f32 matrices of
  // 100 000 rows and more: 16 384-column bins, one workgroup per CU (tune "tfidf_sum_m" = 1: never, 2: whatever
  // the size (tests))
  const bool pipe = mu_tune_get("tfidf_pipe") != 1;
  const int abl = mu_tune_get("tfidf_abl"), sum_m = mu_tune_get("tfidf_sum_m");
  const bool big = dtype == MU_DTYPE_F32 && pipe && abl == 0 && sum_m != 1 && (n_rows >= 100000 || sum_m == 2);
  if (!d_slab_ptr) {
    int rc = launch_slab_ptr(n_rows, n_cols, d_indptr, d_indices, (int64_t*)d_work, st);
    if (rc) return rc;
  }
  if (big)
    hipLaunchKernelGGL((k_row_col_sums_pipe<float, 4, 0, 2>), dim3(mu_num_cus()), dim3(kSweepThreads), 0, st, n_rows,
                       n_cols, S, d_indptr, d_indices, (const float*)d_values, sp, d_rowsum, partial);
  else if (dtype == MU_DTYPE_F32 && pipe && abl == 1)
    hipLaunchKernelGGL((k_row_col_sums_pipe<float, 4, 1>), dim3(G), dim3(kSweepThreads), 0, st, n_rows, n_cols, S,
                       d_indptr, d_indices, (const float*)d_values, sp, d_rowsum, partial);
  else if (dtype == MU_DTYPE_F32 && pipe && abl == 2)
    hipLaunchKernelGGL((k_row_col_sums_pipe<float, 4, 2>), dim3(G), dim3(kSweepThreads), 0, st, n_rows, n_cols, S,
                       d_indptr, d_indices, (const float*)d_values, sp, d_rowsum, partial);
  else if (dtype == MU_DTYPE_F32 && pipe)
    hipLaunchKernelGGL((k_row_col_sums_pipe<float, 4>), dim3(G), dim3(kSweepThreads), 0, st, n_rows, n_cols, S,
                       d_indptr, d_indices, (const float*)d_values, sp, d_rowsum, partial);
  else if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_row_col_sums<float>, dim3(G), dim3(kSweepThreads), 0, st, n_rows, n_cols, S,
                       d_indptr, d_indices, (const float*)d_values, sp, d_rowsum, partial);
  else  // (f64 values: the kernel of before - the pipelined walk is asm for f32 loads)
    hipLaunchKernelGGL(k_row_col_sums<double>, dim3(G), dim3(kSweepThreads), 0, st, n_rows, n_cols,
                       S, d_indptr, d_indices, (const double*)d_values, sp, d_rowsum, partial);
  MU_CHECK_LAUNCH();
  const int G_used = big ? mu_num_cus() : G;
  hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st,
                     n_cols, G_used, partial, d_colsum);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

extern "C" {

int mu_tfidf_idf(int dtype, int64_t n_cols, double n_obs, const double* d_colsum, int flags,
                 void* d_idf, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  if (n_cols == 0) return MU_OK;
  MU_REQUIRE(d_colsum && d_idf, "null pointer");
  const unsigned blocks = (unsigned)((n_cols + 255) / 256);
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_idf<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_cols, n_obs,
                       d_colsum, flags, (float*)d_idf);
  else
    hipLaunchKernelGGL(k_idf<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_cols, n_obs,
                       d_colsum, flags, (double*)d_idf);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_tfidf_scale(int dtype, int64_t n_rows, const int64_t* d_indptr, const int32_t* d_indices,
                   const void* d_values, const double* d_rowsum, const void* d_idf, double scale,
                   int flags, void* d_out, unsigned long long* d_zero_count, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(!((flags & MU_TFIDF_LOG_TFIDF) && (flags & (MU_TFIDF_LOG_TF | MU_TFIDF_LOG_IDF))),
             "log_tfidf excludes log_tf / log_idf (preproc.py:69-73)");
  hipStream_t st = (hipStream_t)stream;
  if (d_zero_count) MU_CHECK_HIP(hipMemsetAsync(d_zero_count, 0, sizeof(unsigned long long), st));
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_rowsum && d_idf && d_out, "null pointer");
  const int use_scale = !(scale == 0.0 || scale == 1.0);  // preproc.py:101
  const unsigned blocks = grid_for_rows(n_rows);
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_tfidf_scale<float>, dim3(blocks), dim3(256), 0, st, n_rows, d_indptr,
                       d_indices, (const float*)d_values, d_rowsum, (const float*)d_idf,
                       (float)scale, use_scale, flags, (float*)d_out, d_zero_count);
  else
    hipLaunchKernelGGL(k_tfidf_scale<double>, dim3(blocks), dim3(256), 0, st, n_rows, d_indptr,
                       d_indices, (const double*)d_values, d_rowsum, (const double*)d_idf, scale,
                       use_scale, flags, (double*)d_out, d_zero_count);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

static int scale_sweep_impl(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                            const int32_t* d_indices, const void* d_values, const double* d_rowsum, const void* d_idf,
                            double scale, int flags, void* d_out, unsigned long long* d_zero_count, void* d_work,
                            size_t work_bytes, int have_slab_ptr, const int64_t* d_slab_ptr, void* stream);

int mu_tfidf_scale_sweep(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                         const int32_t* d_indices, const void* d_values, const double* d_rowsum,
                         const void* d_idf, double scale, int flags, void* d_out,
                         unsigned long long* d_zero_count, void* d_work, size_t work_bytes,
                         int have_slab_ptr, void* stream) {
  return scale_sweep_impl(dtype, n_rows, n_cols, d_indptr, d_indices, d_values, d_rowsum, d_idf, scale, flags, d_out,
                          d_zero_count, d_work, work_bytes, have_slab_ptr, nullptr, stream);
}

/* the same with the slab pointers of mu_csr_slab_ptr (no work buffer needed) */
int mu_tfidf_scale_sweep_sp(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                            const int32_t* d_indices, const void* d_values, const double* d_rowsum, const void* d_idf,
                            double scale, int flags, void* d_out, unsigned long long* d_zero_count,
                            const int64_t* d_slab_ptr, void* stream) {
  MU_REQUIRE(d_slab_ptr || n_rows == 0 || n_cols == 0, "null slab pointers");
  return scale_sweep_impl(dtype, n_rows, n_cols, d_indptr, d_indices, d_values, d_rowsum, d_idf, scale, flags, d_out,
                          d_zero_count, nullptr, 0, 1, d_slab_ptr, stream);
}

}  // extern "C"

static int scale_sweep_impl(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                            const int32_t* d_indices, const void* d_values, const double* d_rowsum, const void* d_idf,
                            double scale, int flags, void* d_out, unsigned long long* d_zero_count, void* d_work,
                            size_t work_bytes, int have_slab_ptr, const int64_t* d_slab_ptr, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  MU_REQUIRE(!((flags & MU_TFIDF_LOG_TFIDF) && (flags & (MU_TFIDF_LOG_TF | MU_TFIDF_LOG_IDF))),
             "log_tfidf excludes log_tf / log_idf (preproc.py:69-73)");
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative shape");
  hipStream_t st = (hipStream_t)stream;
  if (d_zero_count) MU_CHECK_HIP(hipMemsetAsync(d_zero_count, 0, sizeof(unsigned long long), st));
  if (n_rows == 0 || n_cols == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_rowsum && d_idf && d_out, "null pointer");
  MU_REQUIRE(d_slab_ptr || (d_work && work_bytes >= mu_csr_row_col_sums_worksize(n_rows, n_cols)), "work buffer too small");
  const int64_t S = num_slabs(n_cols);
  const int64_t* sp = d_slab_ptr ? d_slab_ptr : (const int64_t*)d_work;  // (the work buffer: same place as in mu_csr_row_col_sums)
  if (!d_slab_ptr && !have_slab_ptr) {
    int rc = launch_slab_ptr(n_rows, n_cols, d_indptr, d_indices, (int64_t*)d_work, st);
    if (rc) return rc;
  }
  const int use_scale = !(scale == 0.0 || scale == 1.0);  // preproc.py:101
  const int G = sweep_grid();
  // f32: idf slabs of 4 x 8192 columns, one workgroup per CU (r04: 18.2 -> 14.4 ms at 1e6 x 200k, bit-identical;
  // tune "tfidf_wide" = 1 keeps the 8192-column kernel for comparison)
  const int abl = mu_tune_get("tfidf_abl");
  if (dtype == MU_DTYPE_F32 && mu_tune_get("tfidf_wide") != 1 && mu_tune_get("tfidf_pipe") != 1 && abl == 3)
    hipLaunchKernelGGL((k_tfidf_scale_sweep_pipe<4, 4, 3>), dim3(mu_num_cus()), dim3(kSweepThreads), 0, st, n_rows, n_cols,
                       S, d_indptr, d_indices, (const float*)d_values, sp, d_rowsum, (const float*)d_idf,
                       (float)scale, use_scale, flags, (float*)d_out, d_zero_count);
  else if (dtype == MU_DTYPE_F32 && mu_tune_get("tfidf_wide") != 1 && mu_tune_get("tfidf_pipe") != 1 && abl == 4)
    hipLaunchKernelGGL((k_tfidf_scale_sweep_pipe<4, 4, 4>), dim3(mu_num_cus()), dim3(kSweepThreads), 0, st, n_rows, n_cols,
                       S, d_indptr, d_indices, (const float*)d_values, sp, d_rowsum, (const float*)d_idf,
                       (float)scale, use_scale, flags, (float*)d_out, d_zero_count);
  else if (dtype == MU_DTYPE_F32 && mu_tune_get("tfidf_wide") != 1 && mu_tune_get("tfidf_pipe") != 1)
    hipLaunchKernelGGL((k_tfidf_scale_sweep_pipe<4, 4>), dim3(mu_num_cus()), dim3(kSweepThreads), 0, st, n_rows, n_cols,
                       S, d_indptr, d_indices, (const float*)d_values, sp, d_rowsum, (const float*)d_idf,
                       (float)scale, use_scale, flags, (float*)d_out, d_zero_count);
  else if (dtype == MU_DTYPE_F32 && mu_tune_get("tfidf_wide") != 1)
    hipLaunchKernelGGL((k_tfidf_scale_sweep_wide<4, 4>), dim3(mu_num_cus()), dim3(kSweepThreads), 0, st, n_rows, n_cols,
                       S, d_indptr, d_indices, (const float*)d_values, sp, d_rowsum, (const float*)d_idf,
                       (float)scale, use_scale, flags, (float*)d_out, d_zero_count);
  else if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_tfidf_scale_sweep<float>, dim3(G), dim3(kSweepThreads), 0, st, n_rows, n_cols, S,
                       d_indptr, d_indices, (const float*)d_values, sp, d_rowsum, (const float*)d_idf,
                       (float)scale, use_scale, flags, (float*)d_out, d_zero_count);
  else
    hipLaunchKernelGGL(k_tfidf_scale_sweep<double>, dim3(G), dim3(kSweepThreads), 0, st, n_rows, n_cols,
                       S, d_indptr, d_indices, (const double*)d_values, sp, d_rowsum,
                       (const double*)d_idf, scale, use_scale, flags, (double*)d_out, d_zero_count);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

extern "C" {

/* mu_tfidf_scale_sweep for f32 that ALSO writes the row stream of the result (see k_tfidf_scale_sweep_pipe): pair i
 * of row r goes to d_ent[d_row_dst[r] + i].  Same values out, bit for bit. */
int mu_tfidf_scale_sweep_stream(int64_t n_rows, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices,
                                const float* d_values, const double* d_rowsum, const float* d_idf, double scale,
                                int flags, float* d_out, unsigned long long* d_zero_count, void* d_work,
                                size_t work_bytes, int have_slab_ptr, const int64_t* d_slab_ptr,
                                const int64_t* d_row_dst, void* d_ent, void* stream) {
  MU_REQUIRE(!((flags & MU_TFIDF_LOG_TFIDF) && (flags & (MU_TFIDF_LOG_TF | MU_TFIDF_LOG_IDF))),
             "log_tfidf excludes log_tf / log_idf (preproc.py:69-73)");
  MU_REQUIRE(n_rows >= 0 && n_cols >= 0, "negative shape");
  hipStream_t st = (hipStream_t)stream;
  if (d_zero_count) MU_CHECK_HIP(hipMemsetAsync(d_zero_count, 0, sizeof(unsigned long long), st));
  if (n_rows == 0 || n_cols == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_rowsum && d_idf && d_out && d_row_dst && d_ent, "null pointer");
  MU_REQUIRE(d_slab_ptr || (d_work && work_bytes >= mu_csr_row_col_sums_worksize(n_rows, n_cols)), "work buffer too small");
  const int64_t S = num_slabs(n_cols);
  const int64_t* sp = d_slab_ptr ? d_slab_ptr : (const int64_t*)d_work;  // (else: same place as in mu_csr_row_col_sums)
  if (!d_slab_ptr && !have_slab_ptr) {
    int rc = launch_slab_ptr(n_rows, n_cols, d_indptr, d_indices, (int64_t*)d_work, st);
    if (rc) return rc;
  }
  const int use_scale = !(scale == 0.0 || scale == 1.0);  // preproc.py:101
  hipLaunchKernelGGL((k_tfidf_scale_sweep_pipe<4, 4, 0, true>), dim3(mu_num_cus()), dim3(kSweepThreads), 0, st, n_rows,
                     n_cols, S, d_indptr, d_indices, d_values, sp, d_rowsum, d_idf, (float)scale, use_scale, flags,
                     d_out, d_zero_count, d_row_dst, (unsigned long long*)d_ent);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_csr_count_nonzero(int dtype, int64_t n_rows, const int64_t* d_indptr, const void* d_values,
                         int64_t* d_row_nnz, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_row_nnz, "null pointer");
  const unsigned blocks = grid_for_rows(n_rows);
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_count_nonzero<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       n_rows, d_indptr, (const float*)d_values, d_row_nnz);
  else
    hipLaunchKernelGGL(k_count_nonzero<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       n_rows, d_indptr, (const double*)d_values, d_row_nnz);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_csr_compact_nonzero(int dtype, int64_t n_rows, const int64_t* d_indptr,
                           const int32_t* d_indices, const void* d_values,
                           const int64_t* d_new_indptr, int32_t* d_new_indices, void* d_new_values,
                           void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_new_indptr, "null pointer");
  const unsigned blocks = grid_for_rows(n_rows);
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_compact_nonzero<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       n_rows, d_indptr, d_indices, (const float*)d_values, d_new_indptr,
                       d_new_indices, (float*)d_new_values);
  else
    hipLaunchKernelGGL(k_compact_nonzero<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       n_rows, d_indptr, d_indices, (const double*)d_values, d_new_indptr,
                       d_new_indices, (double*)d_new_values);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_exclusive_scan_i64(int64_t n, const int64_t* d_in, int64_t* d_out, void* stream) {
  MU_REQUIRE(n >= 0 && d_out, "bad arguments");
  MU_REQUIRE(n == 0 || d_in, "null input");
  hipStream_t st = (hipStream_t)stream;
  if (n < 65536) {
    hipLaunchKernelGGL(k_exclusive_scan_i64, dim3(1), dim3(1024), 0, st, n, d_in, d_out);
    MU_CHECK_LAUNCH();
    return MU_OK;
  }
  hipLaunchKernelGGL(k_scan_chunk_sums, dim3(kScanChunks), dim3(1024), 0, st, n, d_in, d_out);
  MU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_scan_chunk_offsets, dim3(1), dim3(1024), 0, st, n, d_out);
  MU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_scan_chunk_fill, dim3(kScanChunks), dim3(1024), 0, st, n, d_in, d_out);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_binarize_values(int dtype, int64_t nnz, void* d_values, void* stream) {
  MU_REQUIRE(dtype == MU_DTYPE_F32 || dtype == MU_DTYPE_F64, "dtype must be f32 or f64");
  if (nnz == 0) return MU_OK;
  MU_REQUIRE(d_values, "null pointer");
  int64_t blocks = (nnz + 255) / 256;
  const int64_t cap = (int64_t)mu_num_cus() * 16;
  if (blocks > cap) blocks = cap;
  if (dtype == MU_DTYPE_F32)
    hipLaunchKernelGGL(k_binarize<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       nnz, (float*)d_values);
  else
    hipLaunchKernelGGL(k_binarize<double>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       nnz, (double*)d_values);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
