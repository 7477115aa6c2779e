/*
 * muon_amd.h - C-ABI of libmuon_amd.so: the MI355X (gfx950) kernels behind
 *   muon.atac.pp.tfidf   (/root/reference/muon/_atac/preproc.py:16-129)
 *   muon.atac.tl.lsi     (/root/reference/muon/_atac/tools.py:29-71)
 *   muon.tl.mofa         (/root/reference/muon/_core/tools.py:290-708)
 *
 * The reference is pure Python and has no FFI of its own; the boundary a
 * maintainer binds is therefore "the scipy / mofapy2 calls those three functions
 * make".  Every entry point below names the reference statement it replaces.
 * INTEGRATION.md shows the ctypes stub that would sit in the reference.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success and a negative mu_status otherwise;
 *     mu_last_error() returns a thread-local description of the last failure.
 *   - pointers named d_* are DEVICE pointers (hipMalloc'd by the caller - e.g.
 *     torch tensors' data_ptr() - or obtained from mu_malloc); h_* are host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *     kernels are launched asynchronously on it; nothing synchronises unless the
 *     comment says so.  The library keeps no global state except the error string.
 *   - CSR layout: indptr int64[n_rows+1], indices int32[nnz], values f32|f64[nnz];
 *     column indices sorted ascending inside each row where stated.
 *   - dense blocks are row-major with a fixed leading dimension ld == B columns,
 *     B in {16, 32, 64}.
 */
#ifndef MUON_AMD_H
#define MUON_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MU_OK = 0,
  MU_ERR_ARG = -1,     /* bad argument (null pointer, unsupported B, ...) */
  MU_ERR_HIP = -2,     /* a HIP runtime call or kernel launch failed      */
  MU_ERR_NO_DEVICE = -3,
  MU_ERR_UNSUPPORTED = -4
} mu_status;

#define MU_DTYPE_F32 0
#define MU_DTYPE_F64 1

/* tfidf flag bits: the booleans of preproc.py:18-20 */
#define MU_TFIDF_LOG_TF 1
#define MU_TFIDF_LOG_IDF 2
#define MU_TFIDF_LOG_TFIDF 4

/* ---- runtime plumbing ------------------------------------------------------ */
int mu_version(void);
const char* mu_last_error(void);
int mu_device_count(int* count);
int mu_set_device(int device);
/* name (<= len-1 chars), #CUs and bytes of device memory of `device` */
int mu_device_info(int device, char* name, int len, int* n_cu, size_t* total_mem);
int mu_malloc(void** d_ptr, size_t bytes);
int mu_free(void* d_ptr);
int mu_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
int mu_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
int mu_memset(void* d_dst, int value, size_t bytes, void* stream);
int mu_stream_sync(void* stream);
/* 64-bit change-detecting digest of a HOST byte range on n_threads cores (16 MiB chunks, XXH64-style lanes, chunk
 * digests folded in order): the fingerprint of the API path's resident-copy check (the matrices `adata.X` that
 * /root/reference/muon/_atac/preproc.py:81-129 and tools.py:50-53 read), which a Python-level hash computes on one core. */
int mu_host_hash64(const void* h_ptr, size_t n_bytes, int n_threads, uint64_t seed, uint64_t* h_out); /* blocks the host */

/* ---- TF-IDF (preproc.py:92-119) ------------------------------------------- */
/* Bytes of scratch mu_csr_row_col_sums needs for this shape (d_work). */
size_t mu_csr_row_col_sums_worksize(int64_t n_rows, int64_t n_cols);

/* rowsum[i] = sum_j c_ij  (preproc.py:93  counts.sum(axis=1))
 * colsum[j] = sum_i c_ij  (preproc.py:106 counts.sum(axis=0))
 * One pass over (indices, values): LDS-staged per-column partial sums swept slab by
 * slab over sorted rows, wave shuffle reductions for the row sums, fixed-order final
 * reduction (bit-reproducible).  Requires sorted column indices.  Both outputs f64. */
int mu_csr_row_col_sums(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                        const int32_t* d_indices, const void* d_values, double* d_rowsum,
                        double* d_colsum, void* d_work, size_t work_bytes, void* stream);

/* idf[j] = n_obs / colsum[j]; log1p if MU_TFIDF_LOG_IDF  (preproc.py:106-108), computed
 * and stored in `dtype` (f32 counts give an f32 idf in the reference).
 * n_obs is adata.shape[0] - the GLOBAL cell count when rows are sharded. */
int mu_tfidf_idf(int dtype, int64_t n_cols, double n_obs, const double* d_colsum, int flags,
                 void* d_idf, void* stream);

/* out_ij = f(c_ij) following preproc.py:93-117 in the reference's operation order:
 *   t = (1/rowsum_i)*c_ij ; t *= scale (skipped when scale is 0 or 1) ; log1p if LOG_TF ;
 *   t *= idf_j ; log1p if LOG_TFIDF.
 * Replaces the two diag x CSR SpGEMMs (scipy csr_matmat) and the sparse log1p with one
 * fused pass; arithmetic is done in the dtype of the values like the reference does.
 * d_out may alias d_values.  d_zero_count (may be NULL) receives the number of outputs
 * that are exactly 0 - entries scipy's SpGEMM would have dropped. */
int mu_tfidf_scale(int dtype, int64_t n_rows, const int64_t* d_indptr, const int32_t* d_indices,
                   const void* d_values, const double* d_rowsum, const void* d_idf, double scale,
                   int flags, void* d_out, unsigned long long* d_zero_count, void* stream);

/* The same pass walked slab by slab with the slab of idf in LDS (the per-lane idf gather of
 * mu_tfidf_scale is what bounds it); results are bit-identical.  d_work: a buffer of
 * mu_csr_row_col_sums_worksize bytes; have_slab_ptr != 0 says it is the one a preceding
 * mu_csr_row_col_sums filled for this very (d_indptr, d_indices), whose row/slab pointers are
 * then reused instead of being searched again. */
int mu_tfidf_scale_sweep(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                         const int32_t* d_indices, const void* d_values, const double* d_rowsum,
                         const void* d_idf, double scale, int flags, void* d_out,
                         unsigned long long* d_zero_count, void* d_work, size_t work_bytes,
                         int have_slab_ptr, void* stream);

/* Drop stored entries whose value is exactly 0 (what csr_matmat does to explicit zeros).
 * Step 1 writes the surviving count of every row; the caller scans it into the new
 * indptr (mu_exclusive_scan_i64); step 2 compacts. */
int mu_csr_count_nonzero(int dtype, int64_t n_rows, const int64_t* d_indptr, const void* d_values,
                         int64_t* d_row_nnz, void* stream);
int mu_csr_compact_nonzero(int dtype, int64_t n_rows, const int64_t* d_indptr,
                           const int32_t* d_indices, const void* d_values,
                           const int64_t* d_new_indptr, int32_t* d_new_indices, void* d_new_values,
                           void* stream);
/* out[0]=0, out[i+1]=out[i]+in[i], i<n.  d_in and d_out must not overlap (n >= 65536 runs as a
 * three-pass chunked scan that parks the chunk sums in d_out). */
int mu_exclusive_scan_i64(int64_t n, const int64_t* d_in, int64_t* d_out, void* stream);

/* binarize (preproc.py:148-150): X.data[X.data != 0] = 1, in place (NaN != 0, so NaN -> 1;
 * explicit stored zeros stay 0). */
int mu_binarize_values(int dtype, int64_t nnz, void* d_values, void* stream);

/* ---- CSR transpose (device CSC copy used for X^T * Y) ----------------------- */
size_t mu_csr_transpose_worksize(int64_t n_rows, int64_t n_cols, int64_t nnz);
/* Builds the CSR of X^T: t_indptr int64[n_cols+1], t_indices int32[nnz] (row ids of X,
 * ascending inside each output row), t_values.  Stable => bit-reproducible. */
int mu_csr_transpose(int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz,
                     const int64_t* d_indptr, const int32_t* d_indices, const void* d_values,
                     int64_t* d_t_indptr, int32_t* d_t_indices, void* d_t_values, void* d_work,
                     size_t work_bytes, void* stream);

/* ---- LSI building blocks (tools.py:53: the ARPACK operator of scipy svds) ---- */
/* Y[n_rows x B] = X * Q, X CSR f32, Q dense [n_cols x B] f32 row-major (ld=B).
 * The SpMM that replaces the csr_matvec / csr_matvecs calls inside svds.
 * accumulate != 0 adds into Y instead of overwriting. */
int mu_spmm_f32(int64_t n_rows, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices,
                const float* d_values, const float* d_Q, int B, float* d_Y, int accumulate,
                void* stream);

/* X^T straight from X, cell ids ascending inside every output row - what Z = X^T Y of the iteration streams; no
 * intermediate copy: mu_tpack4_* below (csrc/tpack4.hip; r06 removed the third generation, mu_csr_tpack_*):
 *   1. mu_tpack4_count: col_nnz[c] = stored entries of column c (and keeps its per-row-block column prefixes in d_work)
 *   2. caller lays the output rows out and scans their lengths into row pointers
 *   3. mu_tpack4_fill_stream (row stream) or mu_tpack4_fill_csr with the SAME d_work
 * nnz = stored entries of X (it sizes the row blocks and tiles; pass the same value everywhere).
 * d_slab_ptr (may be NULL): the slab pointers of the SAME index arrays (mu_csr_slab_ptr) if the caller holds them.
 * Stable and free of global atomics => bit-reproducible.  Shapes mu_tpack4_supported refuses (2^31 rows, a row block
 * of 2^29 entries) take mu_csr_transpose + mu_csr_stream_fill (stable as well, slower). */

/* ---- the SpMM of the LSI iteration and its operand, the ROW STREAM (r02) --------------------------
 * Y[perm[p]][0..B-1] = row perm[p] of X times Q for every position p < n_pos (B = 16, 32 or 64): the
 * operator of scipy svds (tools.py:53; _svds.py:441-466 matvec / rmatvec) for a block of B vectors.
 *
 * Row stream: the (column int32, value f32) pairs of the matrix, 8 bytes each, row after row in
 * LAUNCH ORDER without padding - position p holds row perm[p] (perm[p] < 0: no row; perm == NULL: the
 * identity, n_pos = n_rows) at ent[sptr[p] .. sptr[p+1]), sptr int64[n_pos + 1].  The kernel gives 4
 * consecutive positions to the four 16-lane groups of a wave, K such row-sets to a wave, 16 waves
 * to a workgroup (K = mu_spmm_stream_k(n_rows)); the host sorts the rows by length and deals the
 * row-sets round robin to workgroups and waves (muon_amd/_backend.py spmm_layout), so that rows that
 * advance in lock step have similar lengths and every workgroup gets the same mix.  The 64 K rows of
 * one workgroup must span less than 4 GiB of the stream (cursors are 32-bit byte offsets).
 *   1. mu_csr_stream_len: len[p] = stored entries of row perm[p] (0 for an empty position)
 *   2. caller scans len into sptr (mu_exclusive_scan_i64) and allocates ent: 8 bytes x sptr[n_pos]
 *   3. mu_csr_stream_fill (a streaming copy), or - for X^T straight from X -
 *      mu_tpack4_count, the layout of the output rows from col_nnz, and mu_tpack4_fill_stream
 *      (inv int32[n_cols]: output row -> position) with the SAME d_work
 * Requires canonical CSR (sorted column indices, no duplicates).  The entries of a row are
 * accumulated in column order with one fmaf chain per dense column, whatever the layout =>
 * bit-reproducible and layout-independent. */
int mu_spmm_stream_k(int64_t n_rows);

/* ---- r05: the row stream of the TF-IDF result from the scale sweep, and the transposition that reads it ------------
 * mu_tfidf_scale_sweep_stream = mu_tfidf_scale_sweep for f32 values that ALSO writes the row stream of its result:
 * pair i of row r - (column int32, value f32), 8 bytes - goes to d_ent[d_row_dst[r] + i], i.e. the caller lays the
 * rows out in the launch order of mu_spmm_stream_f32 beforehand (row lengths are known before the values are) and
 * the streaming copy mu_csr_stream_fill is not needed (replaces, with the values array it still writes,
 * /root/reference/muon/_atac/preproc.py:96-117; the stream is the matvec operand of scipy svds, _svds.py:441-466).
 *
 * mu_tpack4_*: X^T as a row stream (or CSR) straight from X, fourth generation (csrc/tpack4.hip) - the bytes of
 * scipy's csr.T.tocsr() with sorted indices.  Source: the row stream of X when d_x_ent != NULL (d_x_row_dst[r] = pair index of
 * row r's first pair; d_indptr gives the row lengths) - rows are read as contiguous pairs - else the CSR arrays.
 * Row blocks are fixed row ranges (mu_tpack4_geometry: 16 waves x <= 32 rows), d_work is shared by _count and _fill,
 * d_slab_ptr: see above.  mu_tpack4_supported: 0 for shapes that take the general transposition (2^31 rows, a
 * row block of 2^29 entries).  mu_tpack4_status reads the fill's error word (0 = fine; synchronises: tests). */
int mu_tfidf_scale_sweep_stream(int64_t n_rows, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices,
                                const float* d_values, const double* d_rowsum, const float* d_idf, double scale,
                                int flags, float* d_out, unsigned long long* d_zero_count, void* d_work,
                                size_t work_bytes, int have_slab_ptr, const int64_t* d_slab_ptr,
                                const int64_t* d_row_dst, void* d_ent, void* stream);
/* The slab pointers of a CSR (first entry of every row at or behind every 8192-column boundary; int64[n_rows *
 * (ceil(n_cols / 8192) + 1)]) depend on the index arrays alone, which do not change between ingest, binarize, tfidf
 * and lsi: built ONCE where the device CSR is made (upload, 10x ingest) instead of searched by every tfidf call (26
 * binary searches per row at 200 000 columns: 3.0 ms of a 1e6-cell step).  The _sp entries take the table
 * (mu_tfidf_scale_sweep_stream and mu_tpack4_count have a d_slab_ptr argument of their own);
 * NULL = search as before.  Replaces nothing in the reference by itself: bookkeeping of preproc.py:92-117's kernels. */
int mu_csr_slab_ptr(int64_t n_rows, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices, int64_t* d_sp,
                    void* stream);
/* ... for slabs of `width` columns (int64[n_rows * (ceil(n_cols / width) + 1)]): entries per (row, slab) of the sliced-ELL
 * operand of MOFA's sparse views (mu_spmm_ell16_*; 1024 / 512 columns) without a histogram over every entry */
int mu_csr_slab_ptr_width(int64_t n_rows, int64_t n_cols, int64_t width, const int64_t* d_indptr,
                          const int32_t* d_indices, int64_t* d_sp, void* stream);
int mu_csr_row_col_sums_sp(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                           const int32_t* d_indices, const void* d_values, double* d_rowsum, double* d_colsum,
                           void* d_work, size_t work_bytes, const int64_t* d_slab_ptr, void* stream);
int mu_tfidf_scale_sweep_sp(int dtype, int64_t n_rows, int64_t n_cols, const int64_t* d_indptr,
                            const int32_t* d_indices, const void* d_values, const double* d_rowsum, const void* d_idf,
                            double scale, int flags, void* d_out, unsigned long long* d_zero_count,
                            const int64_t* d_slab_ptr, void* stream);
int mu_tpack4_supported(int64_t n_rows, int64_t n_cols, int64_t nnz);
int mu_tpack4_geometry(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t* rows_per_block, int* n_blocks,
                       int* tile_cols);
size_t mu_tpack4_worksize(int64_t n_rows, int64_t n_cols, int64_t nnz);
int mu_tpack4_count(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr, const int32_t* d_indices,
                    int64_t* d_col_nnz, void* d_work, size_t work_bytes, const int64_t* d_slab_ptr, void* stream);
int mu_tpack4_fill_stream(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr,
                          const int32_t* d_indices, const float* d_values, const int64_t* d_x_row_dst,
                          const void* d_x_ent, const int64_t* d_sptr, const int32_t* d_inv, void* d_ent, void* d_work,
                          size_t work_bytes, void* stream);
int mu_tpack4_fill_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr, const int32_t* d_indices,
                       const float* d_values, const int64_t* d_x_row_dst, const void* d_x_ent,
                       const int64_t* d_t_indptr, int32_t* d_t_indices, float* d_t_values, void* d_work,
                       size_t work_bytes, void* stream);
int mu_tpack4_status(const void* d_work, int64_t n_rows, int64_t n_cols, int64_t nnz, int* h_err);
/* byte offset of that error word (an int32) inside d_work, for callers that read it with a fetch of their own instead of
 * synchronising (lsi reads it with its first Gram fetch and raises if the transposition tripped an invariant) */
size_t mu_tpack4_err_offset(int64_t n_rows, int64_t n_cols, int64_t nnz);
/* byte offset inside d_work of the count pass' prefix table uint32 cnt[n_blocks + 1][n_cols] (cnt[g][c] = entries of column
 * c in the row blocks before g; mu_tpack4_geometry gives the rows per block): where the cells of a row-block range begin
 * inside every row of X^T - the table of mu_spmm_stream_ranges_f32 for the warm start's X_S^T Y_S */
size_t mu_tpack4_cnt_offset(int64_t n_rows, int64_t n_cols, int64_t nnz);
int mu_tpack4_phase_cycles(unsigned long long* h_out6, int reset);
int mu_csr_stream_len(int64_t n_pos, const int32_t* d_perm, const int64_t* d_indptr, int64_t* d_len,
                      void* stream);
int mu_csr_stream_fill(int64_t n_pos, const int32_t* d_perm, const int64_t* d_indptr,
                       const int32_t* d_indices, const float* d_values, const int64_t* d_sptr,
                       void* d_ent, void* stream);
/* B = 64 / 32: k_spmm_win (csrc/spmm_win.hip).  B = 16: k_spmm_narrow (csrc/spmm_narrow.hip: 16 stored entries x 4
 * columns per wave step) - the DEFAULT kernel of every 16-column product on a row stream: lsi with n_comps <= 2, the
 * neighbourhood means of mu.pp.neighbors on embeddings of <= 16 dimensions, MOFA f32 views whose operand is not laid
 * out as sliced ELL (tests/test_gpu_kernels.py runs it against scipy at B = 16).  MOFA's default for <= 16 stacked
 * factor columns is mu_spmm_ell16_* below. */
int mu_spmm_stream_f32(int64_t n_pos, int64_t n_cols, const int64_t* d_sptr, const void* d_ent,
                       const int32_t* d_perm, int k_layout, const float* d_Q, int B, float* d_Y,
                       void* stream);
/* The same row stream (f32 stored values) against an f64 dense block, f64 accumulation and product,
 * B = 16 / 32: mofapy2's default precision (tools.py:308 use_float32=False).  accumulate != 0 adds to
 * d_Y - an f64-valued matrix is the sum of two f32-valued ones (v = fl32(v) + fl32(v - fl32(v)), exact
 * to 2^-48), i.e. two streams and two launches; data that is exact in f32 (counts) needs one. */
/* r06 - the products of lsi's SUBSAMPLED WARM START (muon_amd/_atac/tools.py; the reference has no counterpart: its
 * ARPACK run, /root/reference/muon/_atac/tools.py:53, starts from a random vector) without operands of their own:
 * mu_spmm_stream_ranges_f32 is mu_spmm_stream_f32 (B = 64) restricted to a list of <= 32 column ranges.  h_ranges5 holds
 * {col0, col1, q_off, ta, tb} per range: the entries of position p's row in columns [col0, col1)
 * are pairs d_sptr[p] + d_tbl[ta * tbl_stride + row] .. d_sptr[p] + d_tbl[tb * tbl_stride + row] (row = d_perm[p], or p),
 * column c multiplies row c - col0 + q_off of d_Q (q_rows x 64).  Workgroup (x, y) walks ranges [y per_wg, (y + 1) per_wg)
 * and writes its partial product to d_Y + y * y_stride (elements): the caller sums the ceil(n_ranges / per_wg) partials in
 * fixed order.  X_S^T Y_S: the row stream of X^T as it is, d_tbl = the transposition's count prefixes
 * (mu_tpack4_cnt_offset), one range per run of row blocks, per_wg = n_ranges.  X_S Q: mu_csr_slice_stream's compact
 * stream of the slice's rows, ranges = 8192-column super-slabs, per_wg = 1 .. (the column slabs split over the chip).
 * mu_csr_slice_stream: the (column, value) pairs of <= 32 ranges of consecutive rows of a CSR, range after range, rows in
 * their own order (h_row0 / h_rows: the ranges; h_lo / h_hi: d_indptr at their ends, known to the host), d_sptr
 * int64[n_s + 1], d_rel uint32[(ceil(n_cols / 8192) + 1) * n_s] from the CSR's slab pointers (mu_csr_slab_ptr). */
int mu_csr_slice_stream(int n_ranges, const int64_t* h_row0, const int64_t* h_rows, const int64_t* h_lo,
                        const int64_t* h_hi, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices,
                        const float* d_values, const int64_t* d_slab_ptr, int64_t* d_sptr, void* d_ent, uint32_t* d_rel,
                        void* stream);
int mu_spmm_stream_ranges_f32(int64_t n_pos, const int64_t* d_sptr, const void* d_ent, const int32_t* d_perm,
                              int k_layout, const float* d_Q, int64_t q_rows, float* d_Y, int64_t y_stride,
                              const uint32_t* d_tbl, int64_t tbl_stride, int n_ranges, const int32_t* h_ranges5,
                              int per_wg, void* stream);
int mu_spmm_stream_f64(int64_t n_pos, int64_t n_cols, const int64_t* d_sptr, const void* d_ent,
                       const int32_t* d_perm, int k_layout, const double* d_Q, int B, double* d_Y,
                       int accumulate, void* stream);


/* ---- narrow-block SpMM on a sliced-ELL operand (r04; csrc/spmm_ell.hip) -------------------------------------
 * Y[n, 16] = X Q for the <= 16-column factor blocks of MOFA's sparse views (A = Y (tau o W), B = Y^T Z: the Z / W node
 * updates of mofapy2's ent.run(), /root/reference/muon/_core/tools.py:583-585), on an operand laid out once per fit:
 * positions in launch order (d_perm[p] = row at position p, -1 none), 16 positions = a group = the rows of one wave,
 * columns in slabs of 1024.  The entries of (group, slab) are steps - one entry of each of the 16 rows, padded to the
 * longest row - stored as windows of 4 steps (384 bytes: value[64] f32, then offset[64] u16; slot 4 r + j = row r's
 * step 4 w + j, offset = byte offset of the Q row inside the slab = column % 1024 * 64; padding = (0.0, 0)).
 * A group's windows are contiguous from d_wave_base[group], slab after slab; d_hdr[group][slab] is the number of
 * windows.  d_ent needs eight windows of slack (the kernel reads that far ahead).  A workgroup = `waves` (1 .. 15)
 * consecutive groups plus one wave that copies the Q slabs; mu_spmm_ell16_waves picks it for a row count so that
 * full rounds of workgroups cover the chip - it is a launch parameter, not part of the layout. */
int mu_spmm_ell16_waves(int64_t n_rows);
int mu_spmm_ell16_f32(int waves, int64_t n_pos, int64_t n_cols, const int32_t* d_hdr, const int64_t* d_wave_base,
                      const void* d_ent, const int32_t* d_perm, const float* d_Q, float* d_Y, void* stream);
/* The same operand format against an f64 block (f64 products and sums; mofapy2's default precision): Q rows are 128
 * bytes, so the slabs are 512 columns and offset = column % 512 * 128.  Stored values stay f32 - an f64-valued matrix
 * is two operands, hi = fl32(v) and lo = fl32(v - hi), and two launches, the second with accumulate != 0. */
int mu_spmm_ell16_f64(int waves, int64_t n_pos, int64_t n_cols, const int32_t* d_hdr, const int64_t* d_wave_base,
                      const void* d_ent, const int32_t* d_perm, const double* d_Q, double* d_Y, int accumulate,
                      void* stream);
/* r06: the same two products with the column slabs split into `parts` over blockIdx.y - part y writes its partial product
 * to d_Y + y * y_stride (elements), the caller sums the partials in order - for operands of a few thousand rows (one rank's
 * shard of a sharded mu.tl.mofa fit, /root/reference/muon/_core/tools.py:583-585): every workgroup otherwise pulls all of Q
 * through its LDS.  mu_spmm_ell16_parts: the (waves, parts) to launch with (parts = 1: the plain entries). */
int mu_spmm_ell16_parts(int64_t n_rows, int64_t n_cols, int wide, int* waves, int* parts);
int mu_spmm_ell16_parts_f32(int waves, int parts, int64_t n_pos, int64_t n_cols, const int32_t* d_hdr,
                            const int64_t* d_wave_base, const void* d_ent, const int32_t* d_perm, const float* d_Q,
                            float* d_Y, int64_t y_stride, void* stream);
int mu_spmm_ell16_parts_f64(int waves, int parts, int64_t n_pos, int64_t n_cols, const int32_t* d_hdr,
                            const int64_t* d_wave_base, const void* d_ent, const int32_t* d_perm, const double* d_Q,
                            double* d_Y, int64_t y_stride, void* stream);

/* The layout itself: the windows of a canonical f32 CSR from its slab pointers of the operand's width
 * (mu_csr_slab_ptr_width(slab_cols): d_slab_ptr[row][0 .. S]).  d_perm / d_hdr / d_win_base[group][slab] (first window
 * of every (group, slab): the exclusive scan of d_hdr) as the product reads them, made by the caller from the slab
 * pointers (rows by descending length, the longest row of a group per slab).  Writes every slot of every window
 * (padding included); the eight windows of slack behind them are the caller's to zero. */
int mu_ell16_fill(int64_t n_groups, int64_t n_cols, int64_t nnz, int slab_cols, const int32_t* d_indices, const float* d_values,
                  const int64_t* d_slab_ptr, const int32_t* d_perm, const int32_t* d_hdr, const int64_t* d_win_base,
                  void* d_ent, void* stream);


/* Tuning / ablation knobs (tests and bench only; all default to 0 = what ships):
 *   "spmm_k"     row-sets per wave of the packed SpMM (0 = automatic)
 *   "spmm_waves" waves per workgroup of the packed SpMM: 16 (default), 12, 8
 *   "spmm_pipe"  software pipelining level of the packed SpMM
 *   "spmm_mode"  timing ablations of the packed SpMM (bit mask; results are then WRONG)
 *   "mfma_mode"  timing ablations of the matrix-core SpMM (1 no MFMA, 2 no gathers, 4 no masks; WRONG results)
 *   "ell_mode"   timing ablations of the sliced-ELL SpMM (1 no gathers / FMAs, 2 no Q slab copies; WRONG results)
 *   "tfidf_wide" 1: the f32 TF-IDF scale sweep with 8192-column idf slabs (r03) instead of 32 768 (same results) */
int mu_tune_set(const char* key, int value);
int mu_tune_get(const char* key);

/* Same product in f64 (MOFA runs in float64 unless use_float32=True, tools.py:308). */
int mu_spmm_f64(int64_t n_rows, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices,
                const double* d_values, const double* d_Q, int B, double* d_Y, int accumulate,
                void* stream);

size_t mu_gram_worksize(int64_t n_rows, int B);
/* G[B x B] (f64) = A^T A and colsum[B] (f64) = 1^T A for A [n_rows x B] f32 (ld=B).
 * f64 MFMA (v_mfma_f64_16x16x4_f64), fixed-order reduction. Replaces the dense QR of
 * svds (_svds.py:513) together with mu_dense_apply_f32 (CholeskyQR). */
int mu_gram_f32(int64_t n_rows, int B, const float* d_A, double* d_G, double* d_colsum,
                void* d_work, size_t work_bytes, void* stream);

/* C[B x B] (f64) = A^T Bm for A, Bm [n_rows x B] f32 (ld = B); same work buffer size as mu_gram_f32.
 * Feeds the stopping rule of lsi (angle between the Ritz subspaces of consecutive iterations). */
int mu_gram_cross_f32(int64_t n_rows, int B, const float* d_A, const float* d_Bm, double* d_C,
                      void* d_work, size_t work_bytes, void* stream);

/* Out[n_rows x B] = A[n_rows x B] * M[B x B] + bias[B] (bias may be NULL); f32 MFMA
 * (v_mfma_f32_16x16x4_f32).  Out may alias A. */
int mu_dense_apply_f32(int64_t n_rows, int B, const float* d_A, const float* d_M,
                       const float* d_bias, float* d_Out, void* stream);
/* Z[n_rows x B] -= A[n_rows x B] * C[B x B] with C in f64 as mu_gram_cross_f32 leaves it (rounded to
 * f32 for the MFMA): one block of the Gram-Schmidt projection that keeps a new Krylov block
 * orthogonal to the basis (the reorthogonalisation inside ARPACK's dsaupd, tools.py:53). */
int mu_dense_project_out_f32(int64_t n_rows, int B, const float* d_A, const double* d_C, float* d_Z,
                             void* stream);

/* M[B x B] (f32) = R^-1, upper triangular, zero outside the leading w x w block, for G = R^T R (f64,
 * B x B, B <= 64): the small step of CholeskyQR on the device (one wave, LDS resident) instead of two
 * host round trips.  A pivot that is not safely positive sets *d_flag to 1 (never cleared here) and is
 * clamped: the output stays finite, the caller redoes the step on the host. */
int mu_chol_rinv_f64(int B, int w, const double* d_G, float* d_M, int* d_flag, void* stream);

/* standard normal fill, counter based (seed, element index) -> reproducible */
int mu_randn_f32(int64_t count, uint64_t seed, float* d_out, void* stream);

/* ---- MOFA+: the two passes over a DENSE view per iteration (tall-skinny products on the matrix cores)
 * Y [n_rows x D] row-major with leading dimension ldY (a row range of the view); the factor blocks are
 * zero padded to 16 columns.  Replace the library GEMMs of the dense views (tools.py:583-585 ->
 * mofapy2's Y W / Y^T Z products); both stream Y exactly once.
 *   nn: out[n_rows x 16] = Y * T,  T [D x 16]            (Z update:  A = Y (tau o <W>))
 *   tn: C[D x 16] = Y^T * Z,       Z [n_rows x 16]       (W / tau / ELBO:  B = Y^T <Z>) */
size_t mu_skinny_tn_worksize(int dtype, int64_t n_rows, int64_t D);
int mu_skinny_nn(int dtype, int64_t n_rows, int64_t D, int64_t ldY, const void* d_Y, const void* d_T,
                 void* d_out, void* stream);
int mu_skinny_tn(int dtype, int64_t n_rows, int64_t D, int64_t ldY, const void* d_Y, const void* d_Z,
                 void* d_C, void* d_work, size_t work_bytes, void* stream);
/* The same two products in f64 with Y STORED in f32 (r04): a dense view whose values are exact in f32 - AnnData's
 * default dtype, tools.py:308 converts to float64 for the default use_float32=False - streams half the bytes and is
 * widened in registers; every product and sum is the f64 one.  (work: mu_skinny_tn_worksize(MU_DTYPE_F64, ..).) */
int mu_skinny_nn_f64_f32(int64_t n_rows, int64_t D, int64_t ldY, const float* d_Y, const double* d_T, double* d_out,
                         void* stream);
int mu_skinny_tn_f64_f32(int64_t n_rows, int64_t D, int64_t ldY, const float* d_Y, const double* d_Z, double* d_C,
                         void* d_work, size_t work_bytes, void* stream);

/* ---- MOFA+ coordinate updates (tools.py:585 ent.run(): mofapy2's W and Z node updates) ---- */
/* Spike-and-slab + ARD update of the weights of ONE view, one thread per feature, Gauss-Seidel
 * over the K factors.  Inputs are the sufficient statistics of the current factors:
 *   B[G][D][K] = Y_g^T <Z_g>, tau[G][D], Gz[G][K][K] = <Z_g>^T <Z_g>, Z2[G][K] = sum_n <z_nk^2>,
 *   alpha[K] = <alpha_k>, lth[K] = <ln theta_k>, l1mth[K] = <ln(1-theta_k)>.
 * In/out EW[D][K]; out EW2 = <w^2>, gamma = q(s=1), EWh2 = <w_hat^2>, sig2 = slab variance. */
int mu_mofa_update_w(int dtype, int64_t D, int K, int G, const void* d_B, const void* d_tau,
                     const void* d_Gz, const void* d_Z2, const void* d_alpha, const void* d_lth,
                     const void* d_l1mth, int spikeslab, void* d_EW, void* d_EW2, void* d_gamma,
                     void* d_EWh2, void* d_sig2, void* stream);
/* Update of the factors, one thread per sample.  A[M][N][K] = Y_m (tau_g o <W_m>) (row n uses its
 * own group's tau), pres[M][N] in {0,1} (sample observed in view m), grp[N] group id,
 * Gw[M][G][K][K] = <W>^T diag(tau_g) <W>, dw2[M][G][K] = sum_d tau_gd <w_dk^2>, alphaz[G][K],
 * corr[M][G][K] (nullable): subtracted from A per (view, group) - mu_g^T (tau_g o <W>), the implicit
 * centring of a sparse view.  In/out EZ[N][K]; out EZ2 = <z^2>, sig2 = posterior variance. */
int mu_mofa_update_z(int dtype, int64_t N, int K, int M, int G, const void* d_A, const void* d_pres,
                     const int32_t* d_grp, const void* d_Gw, const void* d_dw2, const void* d_alphaz,
                     const void* d_corr, void* d_EZ, void* d_EZ2, void* d_sig2, void* stream);

/* The K x K statistics of a factor / weight block E[R][K] (second moments E2) in one pass over rows
 * r0 .. r1-1 (csrc/mofa_stats.hip; mofapy2 recomputes the same moments inside its node updates,
 * tools.py:585): with a row weight w (nullable = 1) and an auxiliary row weight a (nullable = 1), both
 * indexed by the absolute row,
 *   pad[r][col0 + k] = (scale_out ? w_r : 1) E[r][k]   (leading dimension ld; nullable; the other columns
 *                      are the caller's - zero padding is written once by the caller)
 *   out_t[k][r]      = the same value, transposed with leading dimension ld_t (nullable)
 *   gram[i][j] = sum_r w_r E[r][i] E[r][j],  s2[k] = sum_r w_r E2[r][k],  s1[k] = sum_r a_r w_r E[r][k]
 * (each nullable).  W side: w = tau_g, a = the feature means of a sparse view -> tau o <W>, Gw, dw2 and the
 * centring correction; Z side: w = the presence mask -> <Z>, Gz, sum <z^2>, sum <z>.  f64 accumulation,
 * fixed-order reduction.  d_work: mu_mofa_rowstats_work_doubles(K) doubles. */
size_t mu_mofa_rowstats_work_doubles(int K);
int mu_mofa_rowstats(int dtype, int64_t r0, int64_t r1, int K, const void* d_E, const void* d_E2,
                     const void* d_wgt, const void* d_aux, int scale_out, void* d_out_pad, int ld, int col0,
                     void* d_out_t, int64_t ld_t, void* d_gram, void* d_s2, void* d_s1, double* d_work,
                     void* stream);

/* Per-feature moments of rows r0 .. r1-1 of a dense row-major view Y[.][D] (f32 / f64 storage, f64 sums) in one pass:
 * partial[chunk][0][j] = sum y_rj, partial[chunk][1][j] = sum y_rj^2 over the chunk's rows, `chunks`
 * (mu_dense_col_moments_chunks(rows, D); <= 65535) chunks of consecutive rows, folded by the caller in a fixed order.
 * The group means mofapy2 centres a view by before training (kept as the intercepts,
 * /root/reference/muon/_core/tools.py:283-286) and the sum of squares behind the noise update. */
int mu_dense_col_moments_chunks(int64_t n_rows, int64_t D);
int mu_dense_col_moments(int dtype, int64_t r0, int64_t r1, int64_t D, const void* d_Y, int chunks,
                         double* d_partial, void* stream);

/* ---- exhaustive nearest-neighbour search, the filter pass (replaces the n x n distance panels + radix top-k
 * of the tensor formulation behind muon.pp.neighbors, /root/reference/muon/_core/preproc.py:366-373,453-461,
 * 525-533, where the reference calls UMAP's approximate NN-descent).  For every query q < n_q and every
 * candidate position c in [c_lo, c_hi) with |x_q - y_c|^2 = sqq[q] + sqc[c] - 2 x_q . y_c < thr[q] and
 * c != self_pos[q]: append (c, that squared distance) to the query's buffer, rows of `cap` entries; cnt[q] =
 * number of candidates that passed (may exceed cap: the caller then redoes the query).  Xq [n_q x p_pad],
 * Xc [>= c_hi x p_pad] row-major f64, zero-padded to p_pad (a multiple of 4) columns. */
int mu_knn_filter_f64(int64_t n_q, int64_t c_lo, int64_t c_hi, int p_pad, const double* d_Xq, const double* d_Xc,
                      const double* d_sqq, const double* d_sqc, const double* d_thr, const int32_t* d_self_pos,
                      int cap, int32_t* d_buf_pos, double* d_buf_d, int32_t* d_cnt, void* stream);
/* The merge after a filter pass (r04): for every query the kc smallest (distance, position) pairs of its list [kc] and
 * the first min(cnt, cap) buffer entries, ascending, ties by position, and d_thr = the kc-th distance - what the
 * reference's NN-descent keeps per point as its heap (pynndescent behind /root/reference/muon/_core/preproc.py:366-373,
 * 517-523).  kc + cap <= 1024; not in place.  A row with cnt > cap is the caller's to redo. */
int mu_knn_merge_f64(int64_t n_q, int kc, int cap, const double* d_cur_d, const int64_t* d_cur_p, const double* d_buf_d,
                     const int32_t* d_buf_pos, const int32_t* d_cnt, double* d_out_d, int64_t* d_out_p, double* d_thr,
                     void* stream);

/* ---- kernel bandwidths of muon.pp.neighbors (/root/reference/muon/_core/preproc.py:400-472): csigma[i] = mean
 * Euclidean distance from cell i to the n_bw cells whose neighbour sets overlap its own least (but do), ties
 * towards the larger distance, i.e. the n_bw smallest keys N (1 - jaccard distance) + (bbox - euclid) / bbox
 * (:53-77), ties by cell number.  g_*: the kNN graph (CSR pattern, int64 row pointers, int32 columns), r_*: its
 * transpose; X [n x p] f64 row-major, p <= 256.  NaN for a cell without candidates; *d_overflow = 1 when a
 * cell's candidate list exceeds 8192 entries (repetitions included) - the caller then uses its tensor formulation;
 * n_bw <= 64. */
int mu_wnn_bandwidth_f64(int64_t n, int p, const double* d_X, const int64_t* d_g_indptr, const int32_t* d_g_indices,
                         const int64_t* d_r_indptr, const int32_t* d_r_indices, int n_bw, double bbox,
                         double* d_csigma, int32_t* d_overflow, void* stream);

/* ---- MOFA+ with element-wise precisions (non-gaussian likelihoods, NaN entries): one Gauss-Seidel sweep over
 * the K factors of every row r with the row's own statistics T[r] (K x K) and b[r] (K), tools.py:585 -> mofapy2's
 * W / Z node updates:  t = b_k - sum_{j != k} E_j T_kj,  prec = T_kk + prior_k,  sigma2 = 1 / prec,  mu = t sigma2,
 * gamma = sigmoid(lth_k - l1mth_k + (ln prior_k - ln prec + t^2 sigma2) / 2) (1 without spike-and-slab),
 * E_k = gamma mu,  E2_k = gamma (mu^2 + sigma2),  Eh2_k = E2_k + (1 - gamma) / prior_k.  d_gamma / d_Eh2 nullable
 * (sample rows).  f64 arithmetic, storage type `dtype`. */
int mu_mofa_gs_update(int dtype, int64_t n, int K, const void* d_T, const void* d_b, const double* d_prior,
                      const double* d_lth, const double* d_l1mth, int spikeslab, void* d_E, void* d_E2,
                      void* d_gamma, void* d_Eh2, void* d_sig2, void* stream);

/* ---- UMAP connectivities of a fixed-degree neighbour table (scanpy's `umap` method behind sc.pp.neighbors /
 * mu.pp.neighbors, preproc.py:615-622): per row rho (first positive distance), sigma by 64 bisection steps towards
 * sum_{j >= 1} exp(-max(d_j - rho, 0) / sigma) = target (= log2 n_neighbors), floors, then val[r][j] = 0 for the row
 * itself, 1 where d <= rho, exp(-(d - rho) / sigma) elsewhere.  dist [n x k] f64 (rounded to f32 inside, as umap),
 * idx [n x k] int64, mean_all = the table's mean distance. */
int mu_umap_strengths_f64(int64_t n, int k, const double* d_dist, const int64_t* d_idx, double target, double mean_all,
                          double* d_val, void* stream);

/* Rows [r0, r1) of a CSR (int64 row pointers, int32 columns, values of `dtype`) as a dense row-major chunk
 * d_out [(r1 - r0) x D]: zero fill and scatter in one pass (the chunk walks of the element-wise-precision MOFA
 * engine: zeros are data for a count likelihood, tools.py:117-141 densifies whole modalities on the host). */
int mu_csr_densify_rows(int dtype, int64_t r0, int64_t r1, int64_t D, const int64_t* d_indptr,
                        const int32_t* d_indices, const void* d_values, void* d_out, void* stream);

/* Bernoulli pseudo-data precision of a dense chunk (mofapy2's Bernoulli node, Jaakkola bound): out = 2 lambda(xi) =
 * tanh(xi / 2) / (2 xi) with xi^2 = max(zeta^2 + a - b, 0), xi >= 1e-8; n elements, arithmetic in the storage type. */
int mu_mofa_jaakkola(int dtype, int64_t n, const void* d_zeta, const void* d_a, const void* d_b, void* d_out,
                     void* stream);

/* Poisson pseudo-data of a dense chunk of predictions zeta [n_rows x D] (mofapy2's Poisson node, Seeger bound):
 * rate = softplus(zeta) clamped away from 0;  mode 0: out = kappa_d zeta - sigmoid(zeta) (1 - y / rate);
 * mode 1: out = y ln(rate) - rate.  Element-wise, arithmetic in the storage type `dtype`; d_out may alias d_zeta. */
int mu_mofa_poisson_pseudo(int dtype, int64_t n_rows, int64_t D, int mode, const void* d_zeta, const void* d_Y,
                           const void* d_kappa, void* d_out, void* stream);

/* A poisson view WITHOUT anything of size N x D (r04, csrc/mofa_poisson.hip): for y = 0 an element of the pseudo-data
 * depends on (z_n, w_d) only, so a pass = a dense sweep over all (n, d) that reads just the two K-column blocks, plus a
 * correction over the stored entries.  mode 0: own = samples, other = features, out[n][k] = sum_d R[n, d] w_dk (the Z
 * update's a = R <W>);  mode 1: own = features, other = samples, out[d][k] = sum_n R[n, d] z_nk (the W update's b = R^T
 * <Z>), kappa indexed by own;  mode 2: out[n] = sum_d y ln(rate) - rate (the likelihood term);  mode 3 (r05): mode 1 and
 * the likelihood term in one sweep, rows of K + 1 values: out[d][0 .. K-1] as mode 1, out[d][K] = sum_n y ln(rate) - rate
 * (d_part [..][n_own][K + 1]) - the ELBO pass of an iteration and the W update of the next read the same factors.  R, rate as in
 * mu_mofa_poisson_pseudo.  E_own [n_own x KP], E_other [n_other x KP] row-major with the K <= 32 columns PADDED with zeros
 * to KP = 4 / 8 / 12 / 16 / 32 (the smallest of these >= K; 16-byte aligned rows); outputs have K columns.
 * mu_mofa_poisson_dense writes PARTIAL results for column blocks of `other_block` rows of the other block
 * (mu_mofa_poisson_blocks_for(dtype, mode, K, n_own, n_other) picks it - r06: from the occupancy of the kernel those
 * arguments select, so that the workgroups fill whole rounds; mu_mofa_poisson_blocks is r04's kernel-blind rule, kept
 * for callers of that release; any multiple of 128 is valid): d_part [ceil(n_other / other_block)][n_own][K] (mode 2: [..][n_own]),
 * to be added in block order; mu_mofa_poisson_sparse ADDS the stored entries' terms to the summed result: (indptr,
 * indices, values) = CSR of the view for modes 0 / 2, of its transpose for mode 1. */
int64_t mu_mofa_poisson_blocks(int64_t n_own, int64_t n_other);
int64_t mu_mofa_poisson_blocks_for(int dtype, int mode, int K, int64_t n_own, int64_t n_other);
int mu_mofa_poisson_dense(int dtype, int mode, int64_t n_own, int64_t n_other, int K, int64_t other_block,
                          const void* d_E_own, const void* d_E_other, const void* d_kappa, void* d_part, void* stream);
int mu_mofa_poisson_sparse(int dtype, int mode, int64_t n_own, int K, const int64_t* d_indptr, const int32_t* d_indices,
                           const void* d_values, const void* d_E_own, const void* d_E_other, void* d_out, void* stream);
/* r06: the same two with an explicit row stride `ld` (in elements) of BOTH factor blocks: a multiple of 4, at least the
 * padded width above; columns K .. ld - 1 zero.  With 9 <= K <= 16 and ld = 16 (64-byte rows of f32) the stored-entry pass
 * reads a row with ONE cache-line look-up - four lanes per entry, k_pois_sparse_quad - instead of three (the look-up rate
 * of the gathers is what bounds it), while the dense sweep still runs its KP = 12 instance for K <= 12. */
int mu_mofa_poisson_dense_ld(int dtype, int mode, int64_t n_own, int64_t n_other, int K, int ld, int64_t other_block,
                             const void* d_E_own, const void* d_E_other, const void* d_kappa, void* d_part,
                             void* stream);
int mu_mofa_poisson_sparse_ld(int dtype, int mode, int64_t n_own, int K, int ld, const int64_t* d_indptr,
                              const int32_t* d_indices, const void* d_values, const void* d_E_own, const void* d_E_other,
                              void* d_out, void* stream);

/* A bernoulli view WITHOUT anything of size N x D (r06, csrc/mofa_bernoulli.hip; mofapy2's Bernoulli node with the Jaakkola
 * bound, reached from /root/reference/muon/_core/tools.py:583-585): the data enter through R = y - 1/2 and the likelihood
 * only (sparse products and the poisson likelihood sweep, mode 2 above), the precision Omega_nd = tanh(xi / 2) / (2 xi),
 * xi^2 = zeta^2 + sum_k (<z_k^2><w_k^2> - <z_k>^2 <w_k>^2), depends on the two factor blocks alone.  The sweep:
 *   out[own][c] = sum_other Omega(own, other) M_other[c],   c over the pc = K (K + 1) / 2 distinct entries of <m m^T>
 * (own = features, other = samples, M = packed <z z^T>: the W update's T;  own = samples, other = features, M = packed
 * <w w^T>: the Z update's S).  E / E2 [rows x K] row-major (first / second moments, stride K), M_other [n_other x ldm] from
 * mu_mofa_pack_moments (k <= l row-major, the diagonal = second moments, zero padded; ldm = mu_mofa_jaakkola_cols(K)),
 * 1 <= K <= 16.  Partial results for column blocks of `other_block` rows (mu_mofa_jaakkola_blocks picks it):
 * d_part [ceil(n_other / other_block)][n_own][pc], to be added in block order.  f32 and f64 on the matrix cores. */
int mu_mofa_jaakkola_cols(int K);
int mu_mofa_pack_moments(int dtype, int64_t n, int K, int ldm, const void* d_E, const void* d_E2, void* d_M,
                         void* stream);
int64_t mu_mofa_jaakkola_blocks(int dtype, int K, int64_t n_own, int64_t n_other);
int mu_mofa_jaakkola_sweep(int dtype, int64_t n_own, int64_t n_other, int K, int64_t other_block, const void* d_E_own,
                           const void* d_E2_own, const void* d_E_other, const void* d_E2_other, const void* d_M_other,
                           int ldm, void* d_part, void* stream);

/* ---- MOFA+ small nodes and the ELBO, fused (tools.py:585 ent.run(): mofapy2's Tau, AlphaW, ThetaW,
 * AlphaZ node updates and calculateELBO()).  Arithmetic in f64 for both storage types; every entry
 * ADDS its ELBO terms to the device scalar *d_elbo.  d_work: mu_mofa_elbo_work_doubles(K) doubles. */
size_t mu_mofa_elbo_work_doubles(int K);
/* tau node of one view and the likelihood term.  yy[G][D] = sum_n (y - mean)^2 over the observed
 * samples of the group, Ngm[G] their number, EW / EW2[D][K], B[G][D][K], Gz[G][K][K], Z2[G][K] as
 * in mu_mofa_update_w.  Out tau[G][D] = <tau>, ltau[G][D] = <ln tau>. */
int mu_mofa_tau_elbo(int dtype, int64_t D, int K, int G, const void* d_yy, const void* d_Ngm,
                     const void* d_EW, const void* d_EW2, const void* d_B, const void* d_Gz,
                     const void* d_Z2, double a0, double b0, void* d_tau, void* d_ltau, double* d_elbo,
                     double* d_work, void* stream);
/* r06, for views whose statistics are per FEATURE (a dense gaussian view with missing entries in the general engine): the
 * expected squared residual S[d] = yy[d] - 2 <w_d> . B[d] + sum_kl Q[d][k, l] <w w^T>_d[k, l] (Q [q_rows x K^2], q_rows = D
 * or 1 = the same block for every feature; <w w^T>[k, k] = <w_k^2>) in f64, and the node's finish for n = G x D (group,
 * feature) pairs with their own f64 counts: a = a0 + N / 2, b = b0 + S / 2, tau = a / b, <ln tau> = psi(a) - ln b, the
 * likelihood and tau-node terms ADDED to *d_elbo (work: mu_mofa_elbo_work_doubles).  Between the two the caller sums S and
 * N over the ranks. */
int mu_mofa_stats_resid(int dtype, int64_t D, int K, int64_t q_rows, const double* d_yy, const void* d_EW,
                        const void* d_EW2, const void* d_B, const void* d_Q, double* d_S, void* stream);
int mu_mofa_tau_finish(int dtype, int64_t n, const double* d_S, const double* d_Ngd, double a0, double b0, void* d_tau,
                       void* d_ltau, double* d_elbo, double* d_work, void* stream);
/* ARD precision (ard != 0: alpha[K], lalpha[K] = <ln alpha>) and sparsity level (spikeslab != 0:
 * lth[K] = <ln theta>, l1mth[K] = <ln(1 - theta)>) of one view's weights from EWh2, gamma, sig2
 * [D][K] (outputs of mu_mofa_update_w), and the ELBO terms of the W, alpha_w and theta nodes.
 * a_alpha = a0 + D / 2; (th_a0, th_b0): the Beta prior of theta. */
int mu_mofa_w_elbo(int dtype, int64_t D, int K, int ard, int spikeslab, const void* d_EWh2,
                   const void* d_gamma, const void* d_sig2, double a_alpha, double a0, double b0,
                   double th_a0, double th_b0, void* d_alpha, void* d_lalpha, void* d_lth, void* d_l1mth,
                   double* d_elbo, double* d_work, void* stream);
/* d_out[0][K] = sum_n <z_nk^2>, d_out[1][K] = sum_n ln sig2_nk over the rows [n0, n1) (one group's
 * samples on this rank; the caller adds the ranks up), then with zs[G][2][K] and the group sizes
 * Ng[G] (f64): the ARD precision of the factors (ard != 0: alpha_z, lalpha_z [G][K]) and the ELBO
 * terms of the Z and alpha_z nodes. */
int mu_mofa_z_sums(int dtype, int64_t n0, int64_t n1, int K, const void* d_EZ2, const void* d_sig2,
                   double* d_out, double* d_work, void* stream);
int mu_mofa_z_elbo(int dtype, int K, int G, int ard, const double* d_zs, const double* d_Ng, double a0,
                   double b0, void* d_alpha_z, void* d_lalpha_z, double* d_elbo, void* stream);

/* ---- synthetic planted-topic counts (bench / tests only) ------ */
/* Pass 1: nnz of every row for rows [row0, row0+n_rows) of the global matrix.
 * Pass 2 (after scanning the counts into indptr): fills indices / values (f32 counts).*/
int mu_synth_row_nnz(int64_t row0, int64_t n_rows, int64_t n_cols, int n_topics, double density,
                     uint64_t seed, int64_t* d_row_nnz, void* stream);
int mu_synth_fill(int64_t row0, int64_t n_rows, int64_t n_cols, int n_topics, double density,
                  uint64_t seed, const int64_t* d_indptr, int32_t* d_indices, float* d_values,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MUON_AMD_H */
