// Runtime plumbing of the C-ABI: error string, device queries, memory helpers.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "common.hpp"

static thread_local char g_err[512] = "";

void mu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mu_num_cus() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n <= 0)
      n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

// tuning / ablation knobs (tests and bench only)
static const char* const kTuneKeys[] = {"spmm_k", "spmm_mode", "gram_wg", "pack_wg", "tpack_dbg", "spmm_narrow_off", "ell_mode", "tfidf_wide", "tfidf_pipe", "nn_interleave", "tfidf_abl", "tfidf_sum_m", "tn_pipe", "nn_fast_off", "stream_pipe", "tpack4_m", "tpack4_c", "scale_stream_off", "tpack4_plain", "tpack4_abl", "tpack4_late", "tpack4_circ", "tpack4_off", "pois_valu", "pois_lane"};
constexpr int kTuneN = sizeof(kTuneKeys) / sizeof(kTuneKeys[0]);
static int g_tune[kTuneN] = {};

extern "C" {

int mu_version(void) { return 601; }  // r06 (601): mu_mofa_poisson_blocks_for, mu_mofa_poisson_dense_ld / _sparse_ld (row stride of the factor blocks);  // r06: mu_spmm_stream_ranges_f32, mu_csr_slice_stream, mu_tpack4_cnt_offset / _err_offset added; the matrix-core SpMM experiment (mu_cells_*, mu_dense_to_f16, mu_spmm_cells_f32, mu_probe_*) and the third-generation transposition (mu_csr_tpack_*) removed (archived: scripts/probes/spmm_mfma.hip, tpack_v3.hip);  // r05: mu_tfidf_scale_sweep_stream, mu_tpack4_* (the transposition on the row stream);  // r04: matrix-core SpMM (mu_cells_*, mu_dense_to_f16, mu_spmm_cells_f32, probes);  // r03: mu_mofa_rowstats, mu_mofa_gs_update, mu_mofa_poisson_pseudo, mu_mofa_jaakkola, mu_csr_densify_rows, mu_knn_filter_f64, mu_wnn_bandwidth_f64, mu_umap_strengths_f64 added, mu_mofa_update_z takes d_corr, mu_spmm_ws_* removed

int mu_tune_set(const char* key, int value) {
  MU_REQUIRE(key, "null key");
  for (int i = 0; i < kTuneN; ++i)
    if (strcmp(key, kTuneKeys[i]) == 0) {
      MU_REQUIRE(value >= 0, "negative value");
      MU_REQUIRE(i != 0 || value <= 16, "spmm_k must be 0..16");
      g_tune[i] = value;
      return MU_OK;
    }
  mu_set_error("mu_tune_set: unknown key %s", key);
  return MU_ERR_ARG;
}

int mu_tune_get(const char* key) {
  if (key)
    for (int i = 0; i < kTuneN; ++i)
      if (strcmp(key, kTuneKeys[i]) == 0) return g_tune[i];
  return -1;
}

const char* mu_last_error(void) { return g_err; }

int mu_device_count(int* count) {
  MU_REQUIRE(count, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    mu_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return MU_ERR_NO_DEVICE;
  }
  *count = n;
  return MU_OK;
}

int mu_set_device(int device) {
  MU_CHECK_HIP(hipSetDevice(device));
  return MU_OK;
}

int mu_device_info(int device, char* name, int len, int* n_cu, size_t* total_mem) {
  hipDeviceProp_t p;
  MU_CHECK_HIP(hipGetDeviceProperties(&p, device));
  if (name && len > 0) {
    snprintf(name, (size_t)len, "%s (%s)", p.name, p.gcnArchName);
  }
  if (n_cu) *n_cu = p.multiProcessorCount;
  if (total_mem) *total_mem = p.totalGlobalMem;
  return MU_OK;
}

int mu_malloc(void** d_ptr, size_t bytes) {
  MU_REQUIRE(d_ptr, "null out pointer");
  MU_CHECK_HIP(hipMalloc(d_ptr, bytes));
  return MU_OK;
}

int mu_free(void* d_ptr) {
  MU_CHECK_HIP(hipFree(d_ptr));
  return MU_OK;
}

int mu_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream) {
  MU_CHECK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return MU_OK;
}

int mu_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream) {
  MU_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return MU_OK;
}

int mu_memset(void* d_dst, int value, size_t bytes, void* stream) {
  MU_CHECK_HIP(hipMemsetAsync(d_dst, value, bytes, (hipStream_t)stream));
  return MU_OK;
}

/* ---- host-side fingerprint of a byte range (r05) ------------------------------------------------------------
 * The resident-copy check of the API path (muon_amd/_atac/preproc.py `_fingerprint`) hashes every byte of a host
 * matrix twice per tfidf -> lsi pair; the Python xxhash binding keeps the GIL, i.e. runs on ONE core whatever the
 * number of threads (35-39 GB/s measured: 0.78 s at 250 000 x 200 000).  This is the same job on n_threads cores: the
 * range is cut into 16 MiB chunks, every chunk digested with four 64-bit multiply-rotate lanes (the XXH64 round
 * structure) and the chunk digests folded in order.  Not a cryptographic hash: a change detector. */
static inline uint64_t mu_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static const uint64_t kP1 = 0x9E3779B185EBCA87ull, kP2 = 0xC2B2AE3D27D4EB4Full, kP3 = 0x165667B19E3779F9ull,
                      kP4 = 0x85EBCA77C2B2AE63ull, kP5 = 0x27D4EB2F165667C5ull;
static inline uint64_t mu_round64(uint64_t acc, uint64_t in) { return mu_rotl64(acc + in * kP2, 31) * kP1; }
static inline uint64_t mu_avalanche64(uint64_t h) {
  h ^= h >> 33; h *= kP2; h ^= h >> 29; h *= kP3; h ^= h >> 32;
  return h;
}
static uint64_t mu_chunk_digest(const unsigned char* p, size_t n, uint64_t seed) {
  uint64_t v1 = seed + kP1 + kP2, v2 = seed + kP2, v3 = seed, v4 = seed - kP1;
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    uint64_t w[4];
    memcpy(w, p + i, 32);
    v1 = mu_round64(v1, w[0]); v2 = mu_round64(v2, w[1]); v3 = mu_round64(v3, w[2]); v4 = mu_round64(v4, w[3]);
  }
  uint64_t h = mu_rotl64(v1, 1) + mu_rotl64(v2, 7) + mu_rotl64(v3, 12) + mu_rotl64(v4, 18);
  h = (h ^ mu_round64(0, v1)) * kP1 + kP4; h = (h ^ mu_round64(0, v2)) * kP1 + kP4;
  h = (h ^ mu_round64(0, v3)) * kP1 + kP4; h = (h ^ mu_round64(0, v4)) * kP1 + kP4;
  h += (uint64_t)n;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = mu_rotl64(h ^ mu_round64(0, w), 27) * kP1 + kP4;
  }
  for (; i < n; ++i) h = mu_rotl64(h ^ (p[i] * kP5), 11) * kP1;
  return mu_avalanche64(h);
}

int mu_host_hash64(const void* h_ptr, size_t n_bytes, int n_threads, uint64_t seed, uint64_t* h_out) {
  MU_REQUIRE(h_out && (h_ptr || n_bytes == 0), "null pointer");
  const size_t chunk = (size_t)16 << 20;
  const size_t n_chunks = n_bytes ? (n_bytes + chunk - 1) / chunk : 1;
  std::vector<uint64_t> dig(n_chunks, 0);
  const unsigned char* p = (const unsigned char*)h_ptr;
  std::atomic<size_t> next(0);
  auto work = [&]() {
    for (;;) {
      const size_t c = next.fetch_add(1);
      if (c >= n_chunks) break;
      const size_t lo = c * chunk, hi = (lo + chunk < n_bytes) ? lo + chunk : n_bytes;
      dig[c] = mu_chunk_digest(p + lo, hi - lo, seed + c);
    }
  };
  int T = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  if ((size_t)T > n_chunks) T = (int)n_chunks;
  // extern "C": nothing may throw past this frame.  std::thread's constructor throws std::system_error when the
  // process cannot start another thread (a cgroup pid limit, RLIMIT_NPROC) and std::vector may throw bad_alloc:
  // whatever threads did start share the chunk counter with the calling thread, which finishes the rest alone.
  std::vector<std::thread> th;
  try {
    if (T > 1) {
      th.reserve(T - 1);
      for (int t = 1; t < T; ++t) th.emplace_back(work);
    }
  } catch (...) {
  }
  work();
  for (auto& x : th) x.join();
  uint64_t h = seed ^ kP5 ^ (uint64_t)n_bytes;
  for (size_t c = 0; c < n_chunks; ++c) h = mu_rotl64(h ^ mu_round64(0, dig[c]), 27) * kP1 + kP4;
  *h_out = mu_avalanche64(h);
  return MU_OK;
}

int mu_stream_sync(void* stream) {
  MU_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MU_OK;
}

}  // extern "C"
