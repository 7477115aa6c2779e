// SpMM on the row stream of a CSR (f32):  Y[n x B] = X[n x d] * Q[d x B],  B = 64 / 32 / 16.
//
// This is the kernel the block Lanczos iteration of muon_amd.atac.tl.lsi spends its time in; it
// stands where ARPACK's reverse-communication loop calls csr_matvec / csr_matvecs through
// scipy.sparse.linalg.svds (/root/reference/muon/_atac/tools.py:53, scipy _svds.py:441-466,516).
// The transposed product runs through the same kernel on the row stream of X^T (tpack4.hip).
//
// Operand ("row stream", built once per lsi() call by mu_csr_stream_fill / mu_tpack4_fill_stream):
// the (column, value) pairs of the matrix, 8 bytes each, row after row in LAUNCH ORDER - position p
// of the launch holds row perm[p] at ent[sptr[p] .. sptr[p+1]) - without any padding.  The rows of
// one workgroup are contiguous, so a cursor is a 32-bit byte offset from the workgroup's base.
//
// r02 (was: 128-byte chunks "PCR16", a window cut out of current ++ next chunk with two
// ds_bpermute and ~30 VALU per pass): a 16-lane group reads the 16 pairs behind its row's cursor
// with ONE unaligned 128-byte request, so the window IS the load result; what is left of stage A
// is a compare, a ballot, four scalar popcounts and the cursor update (13 VALU).  The unconsumed
// tail of a window is requested again one slab later and is served by the L2 (measured: HBM
// fetch 1.15x the algorithmic bytes at K = 4, 2.5x at K = 8 where the XCD's live lines exceed
// its 4 MiB).
//
// Kernel structure (one 1024-thread workgroup = 64*K rows, K <= 8, per CU):
//   * the columns of X are swept in slabs of 256; the slab's 256 Q rows (64 KiB at B = 64) are
//     copied to LDS by LDS-DMA, double buffered; a wave issues its 1 KiB pieces of the NEXT slab
//     one per pass, not as one burst (the burst kept every wave ~200 cycles per pass in the issue
//     queue of the texture path and delayed the window requests queued behind it);
//   * a wave is four 16-lane groups, group g walks the row at position 4k+g of row-set k and keeps
//     its K accumulators in registers; lane `sub` owns dense columns NB*sub .. NB*sub+NB-1;
//   * rows are sorted, so the entries of this slab are a prefix of the window: count them per
//     group (ballot + s_bcnt1), advance the cursor, and request the next window right away - it
//     arrives a full slab sweep later (EXEC-masked global_load_dwordx2 from inline asm: exactly
//     one VMEM instruction per row-set and slab, so completion is tracked with an exact counted
//     s_waitcnt vmcnt instead of the vmcnt(0) hipcc falls back to for conditional loads);
//   * entry e's (LDS address, value) is broadcast inside the group with DPP row_newbcast, one
//     ds_read_b128 serves four rows, FMAs in f32; the LDS reads of window slots 0-7 are issued
//     together (one LDS round trip instead of two; a pass is a chain of dependent round trips).
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.hpp"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"  // 32-bit LDS addresses made from integers

namespace {

constexpr int kSlabCols = 256;   // Q rows per slab: 256 x 256 B = 64 KiB, double buffered
constexpr int kWaves = 16;
constexpr int kKMax = 8;
constexpr int kNX = 110;         // hipcc owns v[0 .. kNX-1] (amdgpu_num_vgpr(55)), the asm v[kNX .. kNX+2K-1]
constexpr int kPadCol = 0x7fffffff;

template <int E>
__device__ __forceinline__ int bcast_i(int x) {
  return __builtin_amdgcn_update_dpp(0, x, 0x150 + E, 0xf, 0xf, true);  // row_newbcast:E
}
template <int E>
__device__ __forceinline__ float bcast_f(float x) {
  return __builtin_bit_cast(float, bcast_i<E>(__builtin_bit_cast(int, x)));
}

// NB = dense columns a lane owns (B = 16 NB): 4 (ds_read_b128), 2 (b64), 1 (b32) in f32.
// DT = type of the dense operand and the product: float, or double (MOFA's default precision: the
// stored values stay f32 - the window machinery is untouched - and every FMA widens its value, a
// lane's columns are NB doubles: B = 16 reads ds_read_b64, B = 32 ds_read_b128)
template <int NB, typename DT = float> struct Vec;
template <> struct Vec<4, float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec<2, float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<1, float> { typedef float type __attribute__((ext_vector_type(1))); };
template <> struct Vec<2, double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct Vec<1, double> { typedef double type __attribute__((ext_vector_type(1))); };
__device__ __forceinline__ float mu_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double mu_fma(double a, double b, double c) { return fma(a, b, c); }

template <int NB, typename DT = float> struct Quad { typename Vec<NB, DT>::type q0, q1, q2, q3; };

// Entries E..E+3 of every group's window: four independent LDS reads.  `base` is the LDS byte
// address of the slab buffer plus this lane's column offset; slots a group does not use carry
// v = 0 (a read of some row of the slab and FMAs with zero).
template <int E, int NB, typename DT = float>
__device__ __forceinline__ Quad<NB, DT> quad_read(unsigned base, int a) {
  typedef __attribute__((address_space(3))) const typename Vec<NB, DT>::type* lds_p;
  const unsigned a0 = (unsigned)bcast_i<E>(a) + base, a1 = (unsigned)bcast_i<E + 1>(a) + base;
  const unsigned a2 = (unsigned)bcast_i<E + 2>(a) + base, a3 = (unsigned)bcast_i<E + 3>(a) + base;
  Quad<NB, DT> r;
  r.q0 = *(lds_p)(a0);
  r.q1 = *(lds_p)(a1);
  r.q2 = *(lds_p)(a2);
  r.q3 = *(lds_p)(a3);
  return r;
}
template <int E, int NB, typename DT = float>
__device__ __forceinline__ void quad_fma(const Quad<NB, DT>& r, float v, typename Vec<NB, DT>::type& acc) {
  const DT v0 = (DT)bcast_f<E>(v), v1 = (DT)bcast_f<E + 1>(v);
  const DT v2 = (DT)bcast_f<E + 2>(v), v3 = (DT)bcast_f<E + 3>(v);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = mu_fma(v0, r.q0[c], acc[c]);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = mu_fma(v1, r.q1[c], acc[c]);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = mu_fma(v2, r.q2[c], acc[c]);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = mu_fma(v3, r.q3[c], acc[c]);
}

// Entries E, E+1 only: the upper half of the window is used by few groups (8 entries per row and
// slab on the bench matrices), so it is gated pair by pair instead of quad by quad.
template <int NB, typename DT = float> struct Pair { typename Vec<NB, DT>::type q0, q1; };
template <int E, int NB, typename DT = float>
__device__ __forceinline__ Pair<NB, DT> pair_read(unsigned base, int a) {
  typedef __attribute__((address_space(3))) const typename Vec<NB, DT>::type* lds_p;
  const unsigned a0 = (unsigned)bcast_i<E>(a) + base, a1 = (unsigned)bcast_i<E + 1>(a) + base;
  Pair<NB, DT> r;
  r.q0 = *(lds_p)(a0);
  r.q1 = *(lds_p)(a1);
  return r;
}
template <int E, int NB, typename DT = float>
__device__ __forceinline__ void pair_fma(const Pair<NB, DT>& r, float v, typename Vec<NB, DT>::type& acc) {
  const DT v0 = (DT)bcast_f<E>(v), v1 = (DT)bcast_f<E + 1>(v);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = mu_fma(v0, r.q0[c], acc[c]);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = mu_fma(v1, r.q1[c], acc[c]);
}

// one LDS-DMA piece: 64 lanes x 16 B land contiguously at the wave-uniform LDS byte address; the source
// is base (scalar) + a 32-bit byte offset per lane (r03: the 64-bit per-lane addresses of r02 cost four
// registers and a 64-bit clamp per piece; the dense operand is < 4 GiB, checked on the host)
__device__ __forceinline__ void dma_piece(const void* base, unsigned byte_off, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(byte_off), "s"(base), "s"(lds_dst)
      : "memory");
}

// The window of every (row-set k, group) that is in flight lives in v[kNX + 2k] (column) and
// v[kNX + 2k + 1] (value bits).  These registers are written by the asm loads while the wave keeps
// running, so they must never be visible to hipcc as values: a compiler-made copy of a register
// whose load is still in flight reads stale data.  The kernel is compiled with
// amdgpu_num_vgpr(55) - on the unified gfx950 register file that caps hipcc's own allocation at
// v[0 .. 109] - and the asm statements name the registers above literally; the clobber lists make
// the kernel descriptor allocate them.  tests/test_layout.py audits the generated ISA.
#pragma clang diagnostic ignored "-Winline-asm"
#define MU_WIN_CLOB                                                                               \
  "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", \
      "v122", "v123", "v124", "v125"

// Request the window at the group's cursor: lanes of `mask` (those still inside their row) load
// their (column, value) pair from base + off; the others get the padding column.  Always exactly
// one VMEM instruction, also when the mask is empty: gfx950 counts a VMEM instruction issued with
// EXEC = 0 in order (scripts/probes/exec0_vmcnt.hip), which is what the counted waits rely on.
template <int k>
__device__ __forceinline__ void request_window(unsigned off, const void* base, unsigned long long mask) {
  unsigned long long save;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "v_mov_b32 v%c4, 0x7fffffff\n\t"
      "s_and_b64 exec, exec, %3\n\t"
      "global_load_dwordx2 v[%c4:%c5], %1, %2\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(save)
      : "v"(off), "s"(base), "s"(mask), "i"(kNX + 2 * k), "i"(kNX + 2 * k + 1)
      : MU_WIN_CLOB, "scc");  // s_and_b64 writes SCC: hipcc does keep compares alive across the asm
}

// wait until at most N VMEM operations are outstanding, then read the window of row-set k
template <int k, int N>
__device__ __forceinline__ void wait_window(int& col, int& valbits) {
  asm volatile(
      "s_waitcnt vmcnt(%c2)\n\t"
      "v_mov_b32 %0, v%c3\n\t"
      "v_mov_b32 %1, v%c4"
      : "=v"(col), "=v"(valbits)
      : "i"(N), "i"(kNX + 2 * k), "i"(kNX + 2 * k + 1)
      : MU_WIN_CLOB);
}

template <int k>
__device__ __forceinline__ void set_window(int col, int valbits) {
  asm volatile(
      "v_mov_b32 v%c2, %0\n\t"
      "v_mov_b32 v%c3, %1"
      :
      : "v"(col), "v"(valbits), "i"(kNX + 2 * k), "i"(kNX + 2 * k + 1)
      : MU_WIN_CLOB);
}

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// r06 - RANGED launches (the warm start's products on a cell slice): a workgroup walks a LIST of column
// ranges instead of all columns, and where a row's entries of a range begin and end comes from a table of 32-bit
// offsets relative to the row's first pair, indexed by the matrix row:
//     lo = sptr[p] + tbl[ta * stride + row],   hi = sptr[p] + tbl[tb * stride + row],   row = perm[p]
// Two users (muon_amd/_backend.py):
//   * X_S^T Y_S on the row stream of X^T AS IT IS: the cells of a row block range [g0, g1) of the transposition are a
//     contiguous piece of every row of X^T (cells ascend), and where it begins is the count pass' prefix table
//     (csrc/tpack4.hip: cnt[g][peak] = entries of that peak in row blocks before g) - no operand of the slice is built;
//     the dense operand is the compact Y_S, range r's cells at rows q_off ..;
//   * X_S Q on a compact stream of the slice's rows with the COLUMN SLABS SPLIT over blockIdx.y (a slice of a few
//     thousand rows is a handful of workgroups: each would sweep all 782 slabs on 1/20 of the chip): 8192-column
//     super-slabs from the slab pointers, partial products per blockIdx.y, summed in fixed order by the caller.
// A range may start at any column: its slabs are the 256 columns from there on (LDS row = (column - start) mod 256).
constexpr int kMaxRanges = 32;
struct WinRange {
  int col0, col1;  // columns [col0, col1) of the operand's index space
  int q_off;       // row of the dense operand that holds column col0
  int ta, tb;      // table rows of the begin / end offsets
};
struct WinRanges {
  const uint32_t* tbl;
  int64_t stride;    // table row length
  int64_t q_rows;    // rows of the dense operand (the slab copies clamp there)
  int64_t y_stride;  // elements between the partial products of consecutive blockIdx.y
  int n, per_wg;     // ranges in total; workgroup (x, y) walks ranges [y per_wg, (y + 1) per_wg)
  WinRange r[kMaxRanges];
};

struct Win {      // what stage A of a pass hands to stage B
  int a;          // lane e of a group: LDS byte offset (inside the slab) of window entry e
  float vv;       // lane e: value of window entry e, 0 if the entry is not of this slab
  unsigned any16; // bit e: some group of the wave uses window entry e
};

// MODE is 0 in production; the other bits switch parts of the kernel off for timing ablations
// (results are then wrong on purpose): 1 no LDS gathers / FMAs, 8 no window requests (and no
// overflow passes), 32 window slots 0-7 as two batches of four LDS reads (r01), 64 per-wave cycle
// accounting instead of the product.
template <int K, int MODE, int NB, typename DT, bool RNG = false>
__device__ __forceinline__ void spmm_win_body(int64_t n_pos, int64_t n_cols,
                                              const int64_t* __restrict__ sptr,
                                              const unsigned long long* __restrict__ ent,
                                              const int32_t* __restrict__ perm,
                                              const DT* __restrict__ Q, DT* __restrict__ Y,
                                              int accumulate, const WinRanges* rgp = nullptr) {
  static_assert(K >= 1 && K <= kKMax, "K out of range");
  static_assert(sizeof(DT) == 4 || MODE == 0, "the timing ablations exist for f32 only");
  constexpr int W = kWaves;
  typedef typename Vec<NB, DT>::type acc_t;
  constexpr int kRowBytes = 16 * NB * (int)sizeof(DT);       // one Q row: 16 NB elements (64 .. 256 B)
  constexpr int kSlabBytes = kSlabCols * kRowBytes;          // 64 / 32 / 16 KiB
  constexpr int kRowShift = kRowBytes == 256 ? 8 : (kRowBytes == 128 ? 7 : 6);
  static_assert(kRowBytes <= 256, "a Q row is at most 256 bytes");
  constexpr int kPieces = kSlabBytes / 1024;                 // 1 KiB LDS-DMA pieces per slab
  constexpr int kMyPieces = kPieces / W;                     // per wave and slab: 4 / 2 / 1
  static_assert(kPieces % W == 0, "every wave issues the same number of DMA pieces");
  constexpr bool kDeep = !(MODE & 32);
  // VMEM order of a wave in one slab: D0 R0 D1 R1 ... (DMA piece u of the NEXT slab goes out right
  // before pass u, the pieces a short K leaves over after the last pass; R = the window request of a
  // pass).  Between the request of (row-set k, slab s-1) and pass (k, s) that is always K - 1
  // requests and kMyPieces DMA pieces, whatever k - the last slab issues its (unused) pieces too -
  // so the wait for a window is exact: nothing younger is waited for.
  constexpr int kWaitMain = (K - 1) + kMyPieces;
  // ... and after the last DMA piece come the requests of passes kMyPieces-1 .. K-1
  constexpr int kWaitSlab = K >= kMyPieces ? (K - kMyPieces + 1) : 0;
  // r03: overflow passes without drains.  A row with 16 or more entries in one slab is revisited after
  // the main passes of the slab; its continuation window is the request its main pass issued, so the
  // revisit of row-set k may leave outstanding what was issued AFTER that request - statically the
  // requests of passes k+1 .. K-1 and the DMA pieces after pass k (ovf_wait(k); fewer than the truth
  // when other row-sets were revisited first, i.e. stricter, i.e. safe).  The revisit issues the
  // row-set's next request out of the counted order: instead of draining everything behind it (r02: a
  // full memory round trip, exposed, in nearly every slab of every workgroup - 512 rows x P(16+ of
  // Poisson(8)) = 4 such rows per slab - and the other 15 waves wait for it at the slab barrier), the
  // wave remembers the row-set (ovf_prev) and its main pass of the NEXT slab waits until everything
  // issued before that slab has retired: strict_wait(k) = what this slab itself issued before the wait.
  auto constexpr ovf_wait = [](int k) { return (K - 1 - k) + (kMyPieces > k + 1 ? kMyPieces - (k + 1) : 0); };
  auto constexpr strict_wait = [](int k) { return k + (k + 1 < kMyPieces ? k + 1 : kMyPieces); };
  // ... and a revisit at the END of the slab issues the row-set's next request only k + 1 passes before
  // the next slab needs it (measured: the drain's round trip moved from the barrier into that wait).
  // Row-sets 0 .. kMid-2 are therefore revisited after pass kMid = K - 3 already (their continuation was
  // requested two passes earlier and more: landed), the others at the end of the slab (their next
  // request then has K - 3 passes and more before it is needed).  At the mid point the wave has issued
  // the requests of passes 0 .. kMid+1 and every DMA piece.
  constexpr int kMid = (K >= 5 && !(MODE & (64 | 4096 | 16384))) ? K - 3 : -1;
  constexpr unsigned kEarlyMask = kMid >= 2 ? ((1u << (kMid - 1)) - 1u) : 0u;
  static_assert(kMid < 0 || kMyPieces <= kMid + 2, "every DMA piece is out at the mid point");
  auto constexpr mid_wait = [](int k) { return (kMid + 1 - k) + (kMyPieces > k + 1 ? kMyPieces - (k + 1) : 0); };
  __shared__ float4 qs[2][kSlabBytes / 16];  // double buffer; Q row c of a slab at byte c * kRowBytes
  const int lane = threadIdx.x & 63;
  const int wave = uniform32(threadIdx.x >> 6);
  const int sub = lane & 15, g = lane >> 4;
  const unsigned gmask = (g & 1) ? 0xffff0000u : 0x0000ffffu;  // this group's lanes inside its ballot word
  const int sub_off = sub * (NB * (int)sizeof(DT));  // this lane's byte offset inside a Q row
  const int64_t rb0 = (int64_t)blockIdx.x * (4 * W * K);
  const int64_t rb1 = (rb0 + 4 * W * K) < n_pos ? (rb0 + 4 * W * K) : n_pos;
  const float4* __restrict__ Q4 = reinterpret_cast<const float4*>(Q);
  const int64_t q4_total = (RNG ? rgp->q_rows : n_cols) * (kRowBytes / 16);
  const unsigned qs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&qs[0][0]);
  // the rows of this workgroup are contiguous in the stream: cursors are byte offsets from here
  const int64_t wg0 = uniform64(sptr[rb0]);
  const char* __restrict__ entb = reinterpret_cast<const char*>(ent + wg0);  // wave-uniform

  acc_t acc[K];
  unsigned off[K];  // byte offset of pair (cursor + sub) of (row-set k, this lane's group)
  unsigned endv;    // lane 16 g + k: byte offset of the end of the row of (row-set k, group g)
  static_for<K>([&](auto kc) {
#pragma unroll
    for (int c = 0; c < NB; ++c) acc[decltype(kc)::value][c] = (DT)0;
  });
  const unsigned q4_last = (unsigned)(q4_total - 1);
  // MODE & 64: per-wave cycle accounting (s_memtime) of the places a pass can spend time in;
  // the passes run unpipelined (A(k) then B(k)) and the sums replace the product in Y.
  unsigned t_wait = 0, t_a = 0, t_b = 0, t_bar = 0, t_dma = 0;
  auto now = [&]() -> unsigned { return (unsigned)__builtin_amdgcn_s_memtime(); };

  // RNG: the ranges of this workgroup one after the other - cursors, windows and the first slab are set up again per
  // range (one exposed round trip each), the accumulators run through.  Otherwise: one "range" = all columns.
  const int r_first = RNG ? (int)blockIdx.y * rgp->per_wg : 0;
  const int r_last = RNG ? ((r_first + rgp->per_wg) < rgp->n ? (r_first + rgp->per_wg) : rgp->n) : 1;
  for (int ri = r_first; ri < r_last; ++ri) {
  const int col0 = RNG ? rgp->r[ri].col0 : 0;
  const int ncols32 = RNG ? rgp->r[ri].col1 : (int)n_cols;
  const int q_shift = RNG ? (rgp->r[ri].q_off - col0) : 0;  // dense operand row of column c: c + q_shift
  {
    // lane 16 g + k looks the row of (row-set k, group g) up; the others idle
    // (RNG: this block runs once per range - its per-lane values are made from an opaque copy of the lane id so that
    //  the compiler recomputes them per range instead of keeping them alive across the slab loop, where the K = 7 / 8
    //  instances have no register to spare: a spill is a VMEM instruction and would break the counted waits)
    int lane_o = lane;
    if constexpr (RNG) asm volatile("" : "+v"(lane_o));
    const int sub_o = RNG ? (lane_o & 15) : sub, g_o = RNG ? (lane_o >> 4) : g;
    const int64_t p = rb0 + ((int64_t)wave * K + sub_o) * 4 + g_o;
    bool ok = (sub_o < K) && (p < rb1);
    unsigned lo;
    if constexpr (RNG) {
      const int64_t row = ok ? (perm ? (int64_t)perm[p] : p) : -1;
      ok = ok && row >= 0;
      const int64_t b = ok ? (sptr[p] - wg0) : 0;
      const uint32_t t0 = ok ? rgp->tbl[(int64_t)rgp->r[ri].ta * rgp->stride + row] : 0u;
      const uint32_t t1 = ok ? rgp->tbl[(int64_t)rgp->r[ri].tb * rgp->stride + row] : 0u;
      lo = (unsigned)((b + (int64_t)t0) << 3);
      endv = (unsigned)((b + (int64_t)t1) << 3);
    } else {
      lo = ok ? (unsigned)((sptr[p] - wg0) << 3) : 0u;
      endv = ok ? (unsigned)((sptr[p + 1] - wg0) << 3) : 0u;
    }
    static_for<K>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      off[k] = (unsigned)bcast_i<k>((int)lo) + (unsigned)sub * 8u;
      const bool in = off[k] < (unsigned)bcast_i<k>((int)endv);
      const unsigned long long e = in ? *reinterpret_cast<const unsigned long long*>(entb + off[k]) : 0ull;
      set_window<k>(in ? (int)(unsigned)e : kPadCol, (int)(unsigned)(e >> 32));
    });
  }

  auto dma_one = [&](int64_t s0, int buf, int u) {           // 1 KiB piece u of this wave
    const int piece = wave + u * W;
    unsigned i = (unsigned)((int)s0 + q_shift) * (unsigned)(kRowBytes / 16) + (unsigned)(piece * 64 + lane);  // float4 index into Q
    i = i < q4_last ? i : q4_last;                           // tail / past the end: clamp (never consumed)
    dma_piece(Q4, i << 4, qs_lds + (unsigned)buf * (unsigned)kSlabBytes + (unsigned)piece * 1024u);
  };
#pragma unroll
  for (int u = 0; u < kMyPieces; ++u) dma_one(col0, 0, u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int buf = 0;
  unsigned ovf_prev = 0;  // row-sets whose window in flight was requested by a revisit of the previous slab
  for (int64_t s0 = col0; s0 < ncols32; s0 += kSlabCols, buf ^= 1) {
    // piece u of the next slab, issued right before pass u: the 64 KiB do not hit the texture
    // path as one burst behind which every wave's window requests would queue
    auto next_piece = [&](int u) {
      unsigned td0 = 0;
      if constexpr (MODE & 64) td0 = now();
      dma_one(s0 + kSlabCols, buf ^ 1, u);
      if constexpr (MODE & 64) t_dma += now() - td0;
    };
    const int s_hi = ((int)s0 + kSlabCols) < ncols32 ? ((int)s0 + kSlabCols) : ncols32;
    const unsigned qbase = qs_lds + (unsigned)buf * (unsigned)kSlabBytes + (unsigned)sub_off;
    unsigned again = 0;
    int rv_kind = 1;           // wait of a revisit: 1 at the mid point, 2 first round at the slab end, 0 drain
    unsigned revisited = 0;    // row-sets revisited in this slab

    // Stage A of a pass: the window of (row-set k, every group) has arrived; the entries of this
    // slab are a prefix of it.  Count them per group, advance the cursors, request the next window.
    // SLOW = overflow pass (a row had more than 16 entries in this slab): its request was issued
    // just now, so drain everything; a main pass waits for exactly its request of the previous slab.
    auto stage_a = [&](auto kc, auto slowc) -> Win {
      constexpr int k = decltype(kc)::value;
      constexpr bool SLOW = decltype(slowc)::value;
      if constexpr (MODE & 128) {  // stage B alone: a synthetic 12-entry window for every group
        Win ws;
        ws.a = ((lane * 37 + k * 13 + (int)s0) & (kSlabCols - 1)) << kRowShift;
        ws.vv = 1.0f;
        // (+ 65536 / 131072: the windows of waves 12-15 / 8-15 are empty - stage B alone on 12 / 8 of the
        //  16 waves, what that many "gather" waves of a wave-specialised kernel could deliver)
        ws.any16 = (((MODE & 65536) && wave >= 12) || ((MODE & 131072) && wave >= 8)) ? 0u : 0x0fffu;
        if constexpr (MODE & 64) t_a -= now();
        return ws;
      }
      int col, valbits;
      if constexpr (MODE & 262144) __builtin_amdgcn_s_setprio(2);  // (timing: stage A above the gathers)
      unsigned tw0 = 0;
      if constexpr (MODE & 64) tw0 = now();
      if constexpr (SLOW) {
        // (a row-set that was revisited before in this slab waits for a request out of the counted order)
        if (rv_kind == 1 && kMid >= 0 && !(revisited & (1u << k))) wait_window<k, (kMid >= 0 ? mid_wait(k) : 0)>(col, valbits);
        else if (rv_kind == 2 && !(revisited & (1u << k))) wait_window<k, ovf_wait(k)>(col, valbits);
        else wait_window<k, 0>(col, valbits);
      } else {
        if (ovf_prev & (1u << k)) wait_window<k, (strict_wait(k) < kWaitMain ? strict_wait(k) : kWaitMain)>(col, valbits);
        else wait_window<k, kWaitMain>(col, valbits);
      }
      if constexpr (MODE & 64) {
        const unsigned tw1 = now();
        t_wait += tw1 - tw0;
        t_a -= tw1;  // closed by the caller after the request
      }
      const bool valid = col < s_hi;  // sorted rows: a prefix; padding lanes hold INT_MAX
      const unsigned long long m = __ballot(valid);
      const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
      const unsigned mm = mlo | mhi;
      Win w;
      w.any16 = (mm | (mm >> 16)) & 0xffffu;  // bit e: some group has entry e
      // (r03: the kernel runs at the chip's power limit - the same launch on an all-zero Q takes 16 % less
      //  time, scripts/probes/spmm_power_probe.py -, and a third of the window slots a pass gathers are
      //  padding: they all read row 0 of the slab instead of whatever row their stale column names, so
      //  the padded gathers return the same bytes again and again)
      w.a = valid ? (((col - col0) & (kSlabCols - 1)) << kRowShift) : 0;  // (col0 = 0 unless ranged: a range may start anywhere)
      w.vv = valid ? __builtin_bit_cast(float, valbits) : 0.f;
      if constexpr (!(MODE & 8)) {
        // entries consumed by this lane's group: the bits of its 16 lanes in the ballot, counted on
        // the vector side (a wave issues one scalar instruction per ~5 cycles and nothing else
        // meanwhile: the r02k version - four s_bcnt1 and a packed byte per group - was 14 SALU)
        const unsigned mine = (lane & 32) ? mhi : mlo;
        const unsigned cnt = (unsigned)__popc(mine & gmask);
        off[k] += cnt << 3;  // 8 bytes per consumed pair
        request_window<k>(off[k], entb, __ballot(off[k] < (unsigned)bcast_i<k>((int)endv)));
        // a window used up (some group has all 16 bits set): maybe more in this slab
        if (__ballot(cnt == 16u)) again |= 1u << k;
      }
      if constexpr (MODE & 1024) {  // real stage A, synthetic 12-entry window for stage B
        w.a = ((lane * 37 + k * 13 + (int)s0) & (kSlabCols - 1)) << kRowShift;
        w.vv = 1.0f;
        w.any16 = 0x0fffu;
      }
      if constexpr (MODE & 2048) w.any16 = (w.any16 & 0xf000u) ? 0xffffu : (w.any16 & 0x0f00u) ? 0x0fffu : w.any16;
      return w;
    };
    // Stage B: the LDS gathers and FMAs of the window.
    auto stage_b = [&](auto kc, const Win& w) {
      constexpr int k = decltype(kc)::value;
      // the gathers are the part that is bound by a shared pipe (LDS): a wave that is in them goes
      // first (measured on X Q / X^T Y at 125k x 200k: -6.5 % / -2.8 %)
      __builtin_amdgcn_s_setprio(1);
      if constexpr (MODE & 1) {
        acc[k][0] += (DT)(w.vv + (float)w.a);
      } else if constexpr (kDeep) {
        // (MODE 8192 / 8192 + 32768, timing only: the reads with lanes 32-63 / the odd groups
        // switched off in EXEC - does the LDS pipe charge for inactive quarter-waves?)
        auto qrd = [&](auto ec) {
          Quad<NB, DT> r;
          if constexpr (MODE & 8192) {
            asm volatile("s_mov_b64 exec, %0" ::"s"((MODE & 32768) ? 0x0000ffff0000ffffull : 0x00000000ffffffffull) : "memory");
            r = quad_read<decltype(ec)::value, NB, DT>(qbase, w.a);
            asm volatile("s_mov_b64 exec, -1" ::: "memory");
          } else {
            r = quad_read<decltype(ec)::value, NB, DT>(qbase, w.a);
          }
          return r;
        };
        // A pass is a chain of LDS round trips; sorted rows fill the window from slot 0 (bit e of
        // any16 set => every lower bit set), 8 entries per row and slab on the bench matrices.
        // Slots 0-7 go out as one batch of eight reads, the upper half as one more batch sized by
        // the highest slot in use: two round trips for almost every pass (r01: three to six).
        // (twelve reads in flight for the passes that use slots 8-11: hipcc spills at K >= 7)
        if (w.any16 & 0x00f0u) {
          const Quad<NB, DT> r0 = qrd(std::integral_constant<int, 0>{});
          const Quad<NB, DT> r1 = qrd(std::integral_constant<int, 4>{});
          quad_fma<0, NB, DT>(r0, w.vv, acc[k]);
          quad_fma<4, NB, DT>(r1, w.vv, acc[k]);
          // (hipcc otherwise sinks the second quad into a block shared with the branch below - one
          //  quad of reads, its FMAs, then the next quad: two LDS round trips instead of one)
          asm volatile("; eight reads in flight" ::: "memory");
        } else if (w.any16 & 0x000fu) {
          const Quad<NB, DT> r = qrd(std::integral_constant<int, 0>{});
          quad_fma<0, NB, DT>(r, w.vv, acc[k]);
        }
        if (w.any16 & 0xf000u) {
          const Quad<NB, DT> r0 = qrd(std::integral_constant<int, 8>{});
          const Quad<NB, DT> r1 = qrd(std::integral_constant<int, 12>{});
          quad_fma<8, NB, DT>(r0, w.vv, acc[k]);
          quad_fma<12, NB, DT>(r1, w.vv, acc[k]);
        } else if (w.any16 & 0x0c00u) {
          const Quad<NB, DT> r = qrd(std::integral_constant<int, 8>{});
          quad_fma<8, NB, DT>(r, w.vv, acc[k]);
        } else if (w.any16 & 0x0300u) {
          const Pair<NB, DT> r = pair_read<8, NB, DT>(qbase, w.a);
          pair_fma<8, NB, DT>(r, w.vv, acc[k]);
        }
      } else {
        if (w.any16 & 0x000fu) { const Quad<NB, DT> r = quad_read<0, NB, DT>(qbase, w.a); quad_fma<0, NB, DT>(r, w.vv, acc[k]); }
        if (w.any16 & 0x00f0u) { const Quad<NB, DT> r = quad_read<4, NB, DT>(qbase, w.a); quad_fma<4, NB, DT>(r, w.vv, acc[k]); }
        if (w.any16 & 0xff00u) {
          { const Pair<NB, DT> r = pair_read<8, NB, DT>(qbase, w.a); pair_fma<8, NB, DT>(r, w.vv, acc[k]); }
          if (w.any16 & 0x0c00u) { const Pair<NB, DT> r = pair_read<10, NB, DT>(qbase, w.a); pair_fma<10, NB, DT>(r, w.vv, acc[k]); }
          if (w.any16 & 0x3000u) { const Pair<NB, DT> r = pair_read<12, NB, DT>(qbase, w.a); pair_fma<12, NB, DT>(r, w.vv, acc[k]); }
          if (w.any16 & 0xc000u) { const Pair<NB, DT> r = pair_read<14, NB, DT>(qbase, w.a); pair_fma<14, NB, DT>(r, w.vv, acc[k]); }
        }
      }
      __builtin_amdgcn_s_setprio(0);
    };

    // Slots 0-7 of a main pass in two halves: the eight LDS reads go out BEFORE stage A of the next
    // pass, whose scalar work then runs under their latency; the FMAs and the upper half follow.
    struct Low8 { Quad<NB, DT> r0, r1; };
    auto b_issue = [&](const Win& w) -> Low8 {
      Low8 r;
      r.r0 = quad_read<0, NB, DT>(qbase, w.a);
      r.r1 = quad_read<4, NB, DT>(qbase, w.a);
      return r;
    };
    auto b_finish = [&](auto kc, const Win& w, const Low8& r) {
      constexpr int k = decltype(kc)::value;
      __builtin_amdgcn_s_setprio(1);
      quad_fma<0, NB, DT>(r.r0, w.vv, acc[k]);
      quad_fma<4, NB, DT>(r.r1, w.vv, acc[k]);
      if (w.any16 & 0xf000u) {
        const Quad<NB, DT> q0 = quad_read<8, NB, DT>(qbase, w.a);
        const Quad<NB, DT> q1 = quad_read<12, NB, DT>(qbase, w.a);
        quad_fma<8, NB, DT>(q0, w.vv, acc[k]);
        quad_fma<12, NB, DT>(q1, w.vv, acc[k]);
      } else if (w.any16 & 0x0c00u) {
        const Quad<NB, DT> q = quad_read<8, NB, DT>(qbase, w.a);
        quad_fma<8, NB, DT>(q, w.vv, acc[k]);
      } else if (w.any16 & 0x0300u) {
        const Pair<NB, DT> q = pair_read<8, NB, DT>(qbase, w.a);
        pair_fma<8, NB, DT>(q, w.vv, acc[k]);
      }
      __builtin_amdgcn_s_setprio(0);
    };
    // (measured r02p: 4.80 / 4.89 ms against 4.54 / 4.75 unsplit on X Q / X^T Y at 125k x 200k: the
    //  eight results held across stage A cost more than the latency they hide; kept as ablation 4096)
    constexpr bool kSplit = kDeep && (MODE & 4096) && !(MODE & (1 | 64));
    constexpr bool kPipeB = (MODE & 16384) && !(MODE & (1 | 64 | 4096));

    if constexpr (MODE & 64) {
      static_for<K>([&](auto kc) {
        if constexpr (decltype(kc)::value < kMyPieces) next_piece(decltype(kc)::value);
        const Win wt = stage_a(kc, std::false_type{});
        const unsigned t2 = now();
        t_a += t2;
        stage_b(kc, wt);
        t_b += now() - t2;
      });
    } else {
      if constexpr (kPipeB) {
        // Stage B as a software pipeline ACROSS passes: the gathers of a pass are three batches of
        // four window slots (0-3, 4-7, 8-11; 12-15 on demand), and a batch's LDS reads are issued two
        // batches before its FMAs - the reads of pass k+1's first two batches go out between the
        // FMAs of pass k, behind stage A(k+1).  A pass no longer is a chain of three LDS round trips.
        next_piece(0);
        Win wk = stage_a(std::integral_constant<int, 0>{}, std::false_type{});
        Quad<NB, DT> b0 = quad_read<0, NB, DT>(qbase, wk.a);
        Quad<NB, DT> b1 = quad_read<4, NB, DT>(qbase, wk.a);
        static_for<K>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          Win wn = wk;
          if constexpr (k + 1 < K) {
            if constexpr (k + 1 < kMyPieces) next_piece(k + 1);
            wn = stage_a(std::integral_constant<int, k + 1>{}, std::false_type{});
          }
          const bool n2 = (wk.any16 & 0x0f00u) != 0;
          __builtin_amdgcn_s_setprio(1);
          Quad<NB, DT> b2;
          quad_fma<0, NB, DT>(b0, wk.vv, acc[k]);
          if (n2) b2 = quad_read<8, NB, DT>(qbase, wk.a);
          quad_fma<4, NB, DT>(b1, wk.vv, acc[k]);
          if constexpr (k + 1 < K) b0 = quad_read<0, NB, DT>(qbase, wn.a);
          if (n2) quad_fma<8, NB, DT>(b2, wk.vv, acc[k]);
          if constexpr (k + 1 < K) b1 = quad_read<4, NB, DT>(qbase, wn.a);
          if (wk.any16 & 0xf000u) {
            const Quad<NB, DT> r = quad_read<12, NB, DT>(qbase, wk.a);
            quad_fma<12, NB, DT>(r, wk.vv, acc[k]);
          }
          __builtin_amdgcn_s_setprio(0);
          wk = wn;
        });
      } else {
      // A(k+1) is issued before B(k): the scalar part of the next pass runs under this pass' gathers
      next_piece(0);
      Win w = stage_a(std::integral_constant<int, 0>{}, std::false_type{});
      static_for<K>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        Win wn = w;
        Low8 low;
        if constexpr (kSplit) low = b_issue(w);
        if constexpr (k + 1 < K) {
          if constexpr (k + 1 < kMyPieces) next_piece(k + 1);
          wn = stage_a(std::integral_constant<int, k + 1>{}, std::false_type{});
        }
        if constexpr (kSplit) b_finish(kc, w, low);
        else stage_b(kc, w);
        w = wn;
        if constexpr (k == kMid && kEarlyMask != 0) {
          // mid point: revisit the early row-sets now (one round; a second overflow waits for the slab end)
          const unsigned pend = again & kEarlyMask;
          if (pend) {
            again &= ~kEarlyMask;
            static_for<(kMid >= 2 ? kMid - 1 : 0)>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              if (pend & (1u << j)) {
                const Win wo = stage_a(jc, std::true_type{});
                stage_b(jc, wo);
              }
            });
            revisited |= pend;
          }
        }
      });
      }
    }
#pragma unroll
    for (int u = K; u < kMyPieces; ++u) next_piece(u);  // K < kMyPieces: the pieces left over
    rv_kind = 2;
    if (again) {
      do {
        const unsigned pend = again;
        again = 0;
        static_for<K>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if (pend & (1u << k)) {
            const Win wo = stage_a(kc, std::true_type{});
            unsigned t2 = 0;
            if constexpr (MODE & 64) { t2 = now(); t_a += t2; }
            stage_b(kc, wo);
            if constexpr (MODE & 64) t_b += now() - t2;
          }
        });
        revisited |= pend;
        rv_kind = 0;
      } while (again);
    }
    ovf_prev = revisited;  // their requests are out of the counted order: strict waits in the next slab
    // The last DMA piece of the next slab went out before the requests of passes kMyPieces-1 ..
    // K-1: allowing that many outstanding VMEM operations proves the slab landed.
    unsigned tb0 = 0;
    if constexpr (MODE & 64) tb0 = now();
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kWaitSlab) : "memory");
    __syncthreads();  // next slab visible; everyone finished reading this one
    if constexpr (MODE & 64) t_bar += now() - tb0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requests still in flight target v[kNX ..]
  }  // (ranges)
  if constexpr (MODE & 64) {
    float dep = 0.f;  // (keeps the FMAs of stage B alive)
    static_for<K>([&](auto kc) {
#pragma unroll
      for (int c = 0; c < NB; ++c) dep += (float)acc[decltype(kc)::value][c];
    });
    if (__ballot(dep == 1.2345e-30f)) t_b += 1;
    if (lane < 5) {
      const unsigned t = lane == 0 ? t_wait : lane == 1 ? t_a : lane == 2 ? t_b : lane == 3 ? t_bar : t_dma;
      Y[((int64_t)blockIdx.x * W + wave) * (16 * NB) + lane] = (DT)t;
    }
    return;
  }
  static_for<K>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int64_t p = rb0 + ((int64_t)wave * K + k) * 4 + g;
    if (p < rb1) {
      const int64_t out = perm ? (int64_t)perm[p] : p;  // position -> row of the product (-1: none)
      if (out >= 0) {
        acc_t* y = reinterpret_cast<acc_t*>(Y + (RNG ? (int64_t)blockIdx.y * rgp->y_stride : 0) + out * (16 * NB) + sub * NB);
        *y = accumulate ? (*y + acc[k]) : acc[k];
      }
    }
  });
}

template <int K, int MODE, int NB, typename DT = float>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(55))) void k_spmm_win(
    int64_t n_pos, int64_t n_cols, const int64_t* __restrict__ sptr,
    const unsigned long long* __restrict__ ent, const int32_t* __restrict__ perm,
    const DT* __restrict__ Q, DT* __restrict__ Y, int accumulate) {
  spmm_win_body<K, MODE, NB, DT>(n_pos, n_cols, sptr, ent, perm, Q, Y, accumulate);
}

// the ranged instance (B = 64, f32): the descriptors travel as a by-value kernel argument (scalar loads)
template <int K>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(55))) void k_spmm_win_rng(
    int64_t n_pos, const int64_t* __restrict__ sptr, const unsigned long long* __restrict__ ent,
    const int32_t* __restrict__ perm, const float* __restrict__ Q, float* __restrict__ Y, const WinRanges rg) {
  spmm_win_body<K, 0, 4, float, true>(n_pos, 0, sptr, ent, perm, Q, Y, 0, &rg);
}

// ---- the row stream ------------------------------------------------------------------------
// Position p holds row perm[p] of the matrix (perm == nullptr: the identity; perm[p] < 0: no row).
// The host deals the rows, sorted by length, round robin to workgroups and waves
// (muon_amd/_backend.py: spmm_layout) - see DESIGN.md 4.1.
__global__ __launch_bounds__(256) void k_stream_len(int64_t n_pos, const int32_t* __restrict__ perm,
                                                    const int64_t* __restrict__ indptr,
                                                    int64_t* __restrict__ len) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pos) return;
  const int64_t r = perm ? (int64_t)perm[p] : p;
  len[p] = (r < 0) ? 0 : (indptr[r + 1] - indptr[r]);
}

// a wave per row: streaming copy (8 B in, 8 B out per entry)
__global__ __launch_bounds__(256) void k_stream_fill(int64_t n_pos, const int32_t* __restrict__ perm,
                                                     const int64_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices,
                                                     const float* __restrict__ values,
                                                     const int64_t* __restrict__ sptr,
                                                     unsigned long long* __restrict__ ent) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = uniform64(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t pos = wave0; pos < n_pos; pos += n_waves) {
    const int64_t row = perm ? (int64_t)uniform32(perm[pos]) : pos;
    if (row < 0) continue;
    const int64_t lo = uniform64(indptr[row]), hi = uniform64(indptr[row + 1]);
    const int64_t o0 = uniform64(sptr[pos]);
    // four independent chunks per iteration: the copy runs on a few waves per CU next to the
    // transposition's fill (backend.stream_both) and lives on loads in flight
    const int64_t len = hi - lo;
    for (int64_t j = lane; j < len; j += 256) {
      int32_t c[4];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = j + 64 * u < len;
        c[u] = ok ? indices[lo + j + 64 * u] : 0;
        v[u] = ok ? values[lo + j + 64 * u] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j + 64 * u < len)
          ent[o0 + j + 64 * u] = (unsigned long long)(unsigned)c[u] |
                                 ((unsigned long long)__builtin_bit_cast(unsigned, v[u]) << 32);
    }
  }
}

// The same copy, software pipelined (r04).  What the compiler made of the loop above: the predicated loads became
// branches with an `s_waitcnt vmcnt(0)` behind every second one - the "four chunks in flight" were four round trips per
// 256 entries.  Here the loads are unconditional (a lane past the row's end re-reads the row's last entry and stores it
// again: same address, same bits), issued from asm, two iterations in flight per wave; the wait for a set leaves
// the next set's eight loads outstanding.  Same bytes out.
__device__ __forceinline__ void sf_load(const int32_t* ib, const float* vb, unsigned off, int32_t& c, float& v) {
  asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %4"
               : "=&v"(c), "=&v"(v)
               : "v"(off), "s"(ib), "s"(vb)
               : "memory");
}
template <int N>
__device__ __forceinline__ void sf_wait(int32_t (&c)[4], float (&v)[4]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])
               : "n"(N)
               : "memory");
}
__global__ __launch_bounds__(256) void k_stream_fill_pipe(int64_t n_pos, const int32_t* __restrict__ perm,
                                                          const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices,
                                                          const float* __restrict__ values,
                                                          const int64_t* __restrict__ sptr,
                                                          unsigned long long* __restrict__ ent) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = uniform64(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t pos = wave0; pos < n_pos; pos += n_waves) {
    const int64_t row = perm ? (int64_t)uniform32(perm[pos]) : pos;
    if (row < 0) continue;
    const int64_t lo = uniform64(indptr[row]), hi = uniform64(indptr[row + 1]);
    const int64_t len64 = hi - lo;
    if (len64 <= 0) continue;
    if (len64 >= (1ll << 29)) {  // (32-bit byte offsets below: a row of 5e8 entries takes the plain loop)
      const int64_t o0 = uniform64(sptr[pos]);
      for (int64_t j = lane; j < len64; j += 64)
        ent[o0 + j] = (unsigned long long)(unsigned)indices[lo + j] |
                      ((unsigned long long)__builtin_bit_cast(unsigned, values[lo + j]) << 32);
      continue;
    }
    const int len = (int)len64;
    const int32_t* ib = indices + lo;
    const float* vb = values + lo;
    unsigned long long* ob = ent + uniform64(sptr[pos]);
    int32_t ca[4], cb[4];
    float va[4], vb_[4];
    auto load = [&](int32_t (&c)[4], float (&v)[4], int j0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int q = j0 + lane + 64 * u;
        q = q < len ? q : len - 1;
        sf_load(ib, vb, (unsigned)q * 4u, c[u], v[u]);
      }
    };
    auto store = [&](int32_t (&c)[4], float (&v)[4], int j0) {
      sf_wait<8>(c, v);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int q = j0 + lane + 64 * u;
        q = q < len ? q : len - 1;
        ob[q] = (unsigned long long)(unsigned)c[u] | ((unsigned long long)__builtin_bit_cast(unsigned, v[u]) << 32);
      }
    };
    load(ca, va, 0);
    for (int j0 = 0;; j0 += 512) {
      load(cb, vb_, j0 + 256);  // (past the end: clamped, not stored)
      store(ca, va, j0);
      if (j0 + 256 >= len) break;
      load(ca, va, j0 + 512);
      store(cb, vb_, j0 + 256);
      if (j0 + 512 >= len) break;
    }
    sf_wait<0>(ca, va);  // what is still in flight targets these registers
    sf_wait<0>(cb, vb_);
  }
}

// ---- compact row stream of a cell slice (r06: the operand of the warm start's X_S Q) ---------------------------------
// R ranges of consecutive rows of a CSR, one after the other, rows in their own order (no sort, no deal: the slice is
// 1/32 of the cells and multiplied four times), plus what a ranged launch needs: sptr and, per 8192-column super-slab
// boundary t, the offset of the row's first entry at or behind it relative to the row's first entry (from the CSR's
// slab pointers, mu_csr_slab_ptr) - rel[t * n_s + i].
struct SliceRanges {
  int n;
  int64_t row0[kMaxRanges], lo[kMaxRanges];                // first row of range r, its first entry in the CSR
  int64_t dst_row[kMaxRanges + 1], dst_ent[kMaxRanges + 1];  // where the range begins in the slice (rows, pairs)
};
__global__ __launch_bounds__(256) void k_slice_pairs(const SliceRanges sr, const int32_t* __restrict__ indices,
                                                     const float* __restrict__ values,
                                                     unsigned long long* __restrict__ ent) {
  const int64_t total = sr.dst_ent[sr.n];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = 0;
#pragma unroll
    for (int i = 1; i < kMaxRanges; ++i) r += (i < sr.n && e >= sr.dst_ent[i]) ? 1 : 0;
    const int64_t src = sr.lo[r] + (e - sr.dst_ent[r]);
    ent[e] = (unsigned long long)(unsigned)indices[src] |
             ((unsigned long long)__builtin_bit_cast(unsigned, values[src]) << 32);
  }
}
__global__ __launch_bounds__(256) void k_slice_rows(const SliceRanges sr, int64_t S1, const int64_t* __restrict__ indptr,
                                                    const int64_t* __restrict__ slab_ptr, int64_t* __restrict__ sptr,
                                                    uint32_t* __restrict__ rel) {
  const int64_t n_s = sr.dst_row[sr.n];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_s) return;
  if (i == n_s) {
    sptr[i] = sr.dst_ent[sr.n];
    return;
  }
  int r = 0;
#pragma unroll
  for (int j = 1; j < kMaxRanges; ++j) r += (j < sr.n && i >= sr.dst_row[j]) ? 1 : 0;
  const int64_t row = sr.row0[r] + (i - sr.dst_row[r]);
  const int64_t first = indptr[row];
  sptr[i] = first - sr.lo[r] + sr.dst_ent[r];
  for (int64_t t = 0; t < S1; ++t) rel[t * n_s + i] = (uint32_t)(slab_ptr[row * S1 + t] - first);
}

// K row-sets per wave: the smallest number of full-chip rounds R whose 64*K-row blocks fit the
// register budget (K <= kKMax); one workgroup per CU (128 KiB of LDS at B = 64).
int pick_k(int64_t n_rows) {
  const int64_t cus = mu_num_cus();
  // Many rounds of workgroups anyway (>= 4 at K = 6): six row-sets per wave.  The windows a workgroup
  // keeps alive are what the XCD's L2 has to hold between two visits of a row (32 CUs x 64 K rows x 2
  // lines): at K = 8 they exceed its 4 MiB and the re-requested window tails miss (fabric traffic 2.9x
  // the algorithmic bytes at 1e6 rows); at K = 6 it is 1.7x for +0.5 % time (r02 K sweep).
  if (n_rows >= 4 * 64 * 6 * cus) return 6;
  for (int64_t R = 1; R <= 4096; ++R) {
    const int64_t k = (n_rows + 64 * cus * R - 1) / (64 * cus * R);
    if (k <= kKMax) return (int)(k < 1 ? 1 : k);
  }
  return kKMax;
}

template <int K, int MODE>
int launch(int B, hipStream_t st, int64_t n_pos, int64_t n_cols, const int64_t* sptr,
           const unsigned long long* ent, const int32_t* perm, const float* Q, float* Y) {
  const int64_t wgs = (n_pos + 64 * K - 1) / (64 * K);
  if (B == 64)
    hipLaunchKernelGGL((k_spmm_win<K, MODE, 4>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols,
                       sptr, ent, perm, Q, Y, 0);
  else if (B == 32)
    hipLaunchKernelGGL((k_spmm_win<K, MODE, 2>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols,
                       sptr, ent, perm, Q, Y, 0);
  else
    hipLaunchKernelGGL((k_spmm_win<K, MODE, 1>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols,
                       sptr, ent, perm, Q, Y, 0);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

// f32 stored values, f64 dense operand and product (B = 16 / 32)
template <int K>
int launch_f64(int B, hipStream_t st, int64_t n_pos, int64_t n_cols, const int64_t* sptr,
               const unsigned long long* ent, const int32_t* perm, const double* Q, double* Y,
               int accumulate) {
  const int64_t wgs = (n_pos + 64 * K - 1) / (64 * K);
  if (B == 32)
    hipLaunchKernelGGL((k_spmm_win<K, 0, 2, double>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos,
                       n_cols, sptr, ent, perm, Q, Y, accumulate);
  else
    hipLaunchKernelGGL((k_spmm_win<K, 0, 1, double>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos,
                       n_cols, sptr, ent, perm, Q, Y, accumulate);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // namespace

// csrc/spmm_narrow.hip: the B = 16 product on the same operand with 16 entries x 4 columns per wave step
int mu_spmm_narrow_f32_launch(hipStream_t st, int64_t n_pos, int64_t n_cols, int K, const int64_t* sptr,
                              const unsigned long long* ent, const int32_t* perm, const float* Q, float* Y);

extern "C" {

int mu_spmm_stream_k(int64_t n_rows) { return pick_k(n_rows); }

int mu_csr_stream_len(int64_t n_pos, const int32_t* d_perm, const int64_t* d_indptr, int64_t* d_len,
                      void* stream) {
  MU_REQUIRE(n_pos >= 0, "negative size");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_len, "null pointer");
  hipLaunchKernelGGL(k_stream_len, dim3((unsigned)((n_pos + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, n_pos, d_perm, d_indptr, d_len);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_csr_stream_fill(int64_t n_pos, const int32_t* d_perm, const int64_t* d_indptr,
                       const int32_t* d_indices, const float* d_values, const int64_t* d_sptr,
                       void* d_ent, void* stream) {
  MU_REQUIRE(n_pos >= 0, "negative size");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_sptr && d_ent, "null pointer");
  int64_t blocks = (n_pos + 3) / 4;
  // workgroups per CU: 32 alone; a caller that runs the copy next to a kernel that needs whole CUs
  // (the transpose on another stream) lowers it so that both stay resident (tune pack_wg)
  const int per_cu = mu_tune_get("pack_wg") > 0 ? mu_tune_get("pack_wg") : 32;
  const int64_t cap = (int64_t)mu_num_cus() * per_cu;
  if (blocks > cap) blocks = cap;
  if (mu_tune_get("stream_pipe") == 1)  // (the loop of before, for comparison)
    hipLaunchKernelGGL(k_stream_fill, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_pos,
                       d_perm, d_indptr, d_indices, d_values, d_sptr, (unsigned long long*)d_ent);
  else
    hipLaunchKernelGGL(k_stream_fill_pipe, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_pos,
                       d_perm, d_indptr, d_indices, d_values, d_sptr, (unsigned long long*)d_ent);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_spmm_stream_f32(int64_t n_pos, int64_t n_cols, const int64_t* d_sptr, const void* d_ent,
                       const int32_t* d_perm, int k_layout, const float* d_Q, int B, float* d_Y,
                       void* stream) {
  MU_REQUIRE(B == 16 || B == 32 || B == 64, "B must be 16, 32 or 64");
  MU_REQUIRE(n_pos >= 0 && n_cols > 0 && n_cols < ((int64_t)1 << 31), "shape out of range");
  // (the Q slabs are addressed with 32-bit byte offsets from d_Q)
  MU_REQUIRE((n_cols + 512) * (int64_t)B * 4 < ((int64_t)1 << 32), "dense operand of 4 GiB or more");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_sptr && d_ent && d_Q && d_Y, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const unsigned long long* ent = (const unsigned long long*)d_ent;
  int K = (k_layout >= 1 && k_layout <= kKMax) ? k_layout : pick_k(n_pos);
  const int force_k = mu_tune_get("spmm_k");  // tests / tuning only (mu_tune_set); 0 in production
  if (force_k >= 1 && force_k <= kKMax) K = force_k;
  const int mode = mu_tune_get("spmm_mode");
  // B = 16: the narrow-block kernel (tune spmm_narrow_off = 1: this file's NB = 1 instance, kept for A/B runs)
  if (B == 16 && (mode == 0 || (mode >= 2 && mode <= 6)) && mu_tune_get("spmm_narrow_off") == 0)
    return mu_spmm_narrow_f32_launch(st, n_pos, n_cols, K, d_sptr, ent, d_perm, d_Q, d_Y);
#define MU_ARGS B, st, n_pos, n_cols, d_sptr, ent, d_perm, d_Q, d_Y
  if (mode != 0) {
    // timing ablations that are still compiled (the others - 32, 4096, 8192, 16384, 65536, 131072,
    // 262144: DESIGN.md 4.2 has their r02 measurements - need an entry here to be instantiated again)
    if (mode == 1 && K == 8) return launch<8, 1>(MU_ARGS);
    if (mode == 9 && K == 8) return launch<8, 9>(MU_ARGS);
    if (mode == 64 && K == 8) return launch<8, 64>(MU_ARGS);
    if (mode == 128 && K == 8) return launch<8, 128>(MU_ARGS);
    if (mode == 128 + 64 && K == 8) return launch<8, 128 + 64>(MU_ARGS);
    mu_set_error("spmm_mode %d has no compiled instance for K = %d", mode, K);
    return MU_ERR_ARG;
  }
  switch (K) {
    case 1: return launch<1, 0>(MU_ARGS);
    case 2: return launch<2, 0>(MU_ARGS);
    case 3: return launch<3, 0>(MU_ARGS);
    case 4: return launch<4, 0>(MU_ARGS);
    case 5: return launch<5, 0>(MU_ARGS);
    case 6: return launch<6, 0>(MU_ARGS);
    case 7: return launch<7, 0>(MU_ARGS);
    default: return launch<8, 0>(MU_ARGS);
  }
#undef MU_ARGS
}

int mu_csr_slice_stream(int n_ranges, const int64_t* h_row0, const int64_t* h_rows, const int64_t* h_lo,
                        const int64_t* h_hi, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices,
                        const float* d_values, const int64_t* d_slab_ptr, int64_t* d_sptr, void* d_ent, uint32_t* d_rel,
                        void* stream) {
  MU_REQUIRE(n_ranges >= 1 && n_ranges <= kMaxRanges && n_cols > 0, "1 .. 32 ranges");
  MU_REQUIRE(h_row0 && h_rows && h_lo && h_hi && d_indptr && d_indices && d_values && d_slab_ptr && d_sptr && d_ent && d_rel,
             "null pointer");
  SliceRanges sr;
  sr.n = n_ranges;
  sr.dst_row[0] = sr.dst_ent[0] = 0;
  for (int i = 0; i < n_ranges; ++i) {
    MU_REQUIRE(h_rows[i] >= 0 && h_hi[i] >= h_lo[i] && h_hi[i] - h_lo[i] < ((int64_t)1 << 32), "range out of bounds");
    sr.row0[i] = h_row0[i];
    sr.lo[i] = h_lo[i];
    sr.dst_row[i + 1] = sr.dst_row[i] + h_rows[i];
    sr.dst_ent[i + 1] = sr.dst_ent[i] + (h_hi[i] - h_lo[i]);
  }
  for (int i = n_ranges; i < kMaxRanges; ++i) {
    sr.row0[i] = sr.lo[i] = 0;
    sr.dst_row[i + 1] = sr.dst_row[n_ranges];
    sr.dst_ent[i + 1] = sr.dst_ent[n_ranges];
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t n_s = sr.dst_row[n_ranges], nnz_s = sr.dst_ent[n_ranges];
  const int64_t S1 = (n_cols + 8191) / 8192 + 1;
  if (nnz_s > 0) {
    int64_t blocks = (nnz_s + 1023) / 1024;
    const int64_t cap = (int64_t)mu_num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_slice_pairs, dim3((unsigned)blocks), dim3(256), 0, st, sr, d_indices, d_values,
                       (unsigned long long*)d_ent);
    MU_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_slice_rows, dim3((unsigned)((n_s + 1 + 255) / 256)), dim3(256), 0, st, sr, S1, d_indptr, d_slab_ptr,
                     d_sptr, d_rel);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_spmm_stream_ranges_f32(int64_t n_pos, const int64_t* d_sptr, const void* d_ent, const int32_t* d_perm,
                              int k_layout, const float* d_Q, int64_t q_rows, float* d_Y, int64_t y_stride,
                              const uint32_t* d_tbl, int64_t tbl_stride, int n_ranges, const int32_t* h_ranges5,
                              int per_wg, void* stream) {
  MU_REQUIRE(n_pos >= 0 && q_rows > 0 && n_ranges >= 1 && n_ranges <= kMaxRanges && per_wg >= 1, "shape out of range");
  MU_REQUIRE(k_layout >= 1 && k_layout <= kKMax, "the layout's K is needed");
  MU_REQUIRE((q_rows + 512) * (int64_t)64 * 4 < ((int64_t)1 << 32), "dense operand of 4 GiB or more");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_sptr && d_ent && d_Q && d_Y && d_tbl && h_ranges5, "null pointer");
  WinRanges rg;
  rg.tbl = d_tbl;
  rg.stride = tbl_stride;
  rg.q_rows = q_rows;
  rg.y_stride = y_stride;
  rg.n = n_ranges;
  rg.per_wg = per_wg;
  for (int i = 0; i < n_ranges; ++i) {
    const int32_t* h = h_ranges5 + 5 * i;
    MU_REQUIRE(h[0] >= 0 && h[1] >= h[0] && h[2] >= 0 && h[3] >= 0 && h[4] >= 0, "range out of bounds");
    rg.r[i] = WinRange{h[0], h[1], h[2], h[3], h[4]};
  }
  for (int i = n_ranges; i < kMaxRanges; ++i) rg.r[i] = WinRange{0, 0, 0, 0, 0};
  const int K = k_layout;
  const unsigned wgs = (unsigned)((n_pos + 64 * K - 1) / (64 * K));
  const unsigned ny = (unsigned)((n_ranges + per_wg - 1) / per_wg);
  hipStream_t st = (hipStream_t)stream;
  const unsigned long long* ent = (const unsigned long long*)d_ent;
#define MU_RNG(K_)                                                                                                   \
  hipLaunchKernelGGL((k_spmm_win_rng<K_>), dim3(wgs, ny), dim3(1024), 0, st, n_pos, d_sptr, ent, d_perm, d_Q, d_Y, rg)
  switch (K) {
    case 1: MU_RNG(1); break;
    case 2: MU_RNG(2); break;
    case 3: MU_RNG(3); break;
    case 4: MU_RNG(4); break;
    case 5: MU_RNG(5); break;
    case 6: MU_RNG(6); break;
    case 7: MU_RNG(7); break;
    default: MU_RNG(8); break;
  }
#undef MU_RNG
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_spmm_stream_f64(int64_t n_pos, int64_t n_cols, const int64_t* d_sptr, const void* d_ent,
                       const int32_t* d_perm, int k_layout, const double* d_Q, int B, double* d_Y,
                       int accumulate, void* stream) {
  MU_REQUIRE(B == 16 || B == 32, "B must be 16 or 32");
  MU_REQUIRE(n_pos >= 0 && n_cols > 0 && n_cols < ((int64_t)1 << 31), "shape out of range");
  MU_REQUIRE((n_cols + 512) * (int64_t)B * 8 < ((int64_t)1 << 32), "dense operand of 4 GiB or more");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_sptr && d_ent && d_Q && d_Y, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const unsigned long long* ent = (const unsigned long long*)d_ent;
  const int K = (k_layout >= 1 && k_layout <= kKMax) ? k_layout : pick_k(n_pos);
#define MU_ARGS B, st, n_pos, n_cols, d_sptr, ent, d_perm, d_Q, d_Y, accumulate
  switch (K) {
    case 1: return launch_f64<1>(MU_ARGS);
    case 2: return launch_f64<2>(MU_ARGS);
    case 3: return launch_f64<3>(MU_ARGS);
    case 4: return launch_f64<4>(MU_ARGS);
    case 5: return launch_f64<5>(MU_ARGS);
    case 6: return launch_f64<6>(MU_ARGS);
    case 7: return launch_f64<7>(MU_ARGS);
    default: return launch_f64<8>(MU_ARGS);
  }
#undef MU_ARGS
}

}  // extern "C"
