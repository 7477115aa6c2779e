// Runtime plumbing of the C-ABI: error string, device queries, memory helpers.
#include <cstdarg>
#include <cstdio>
#include <cstring>
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
static const char* const kTuneKeys[] = {"spmm_k", "spmm_mode", "spmm_waves", "spmm_pipe",
                                        "tpack_abl", "tpack_c", "gram_wg", "tpack_v2", "pack_wg",
                                        "tpack_dbg", "tpack_rows", "tpack_narrow", "spmm_narrow_off",
                                        "mfma_mode", "ell_mode", "tfidf_wide", "tfidf_pipe", "nn_interleave", "tfidf_abl", "tfidf_sum_m", "tn_pipe", "nn_fast_off", "tpack_asm", "tpack_split", "stream_pipe", "tcount_pipe", "tpack4_m", "tpack4_c", "tpack_v3", "scale_stream_off", "tpack4_plain", "tpack4_abl"};
constexpr int kTuneN = sizeof(kTuneKeys) / sizeof(kTuneKeys[0]);
static int g_tune[kTuneN] = {};

extern "C" {

int mu_version(void) { return 500; }  // r05: mu_tfidf_scale_sweep_stream, mu_tpack4_* (the transposition on the row stream);  // r04: matrix-core SpMM (mu_cells_*, mu_dense_to_f16, mu_spmm_cells_f32, probes);  // r03: mu_mofa_rowstats, mu_mofa_gs_update, mu_mofa_poisson_pseudo, mu_mofa_jaakkola, mu_csr_densify_rows, mu_knn_filter_f64, mu_wnn_bandwidth_f64, mu_umap_strengths_f64 added, mu_mofa_update_z takes d_corr, mu_spmm_ws_* removed

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

int mu_stream_sync(void* stream) {
  MU_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MU_OK;
}

}  // extern "C"
