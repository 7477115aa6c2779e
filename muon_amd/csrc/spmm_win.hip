// SpMM straight from the CSR arrays (f32):  Y[n x B] = X[n x d] * Q[d x B],  B = 64 / 32 / 16.
//
// This is the kernel the block Lanczos iteration of muon_amd.atac.tl.lsi spends its time in; it
// stands where ARPACK's reverse-communication loop calls csr_matvec / csr_matvecs through
// scipy.sparse.linalg.svds (/root/reference/muon/_atac/tools.py:53, scipy _svds.py:441-466,516).
// The transposed product runs through the same kernel on the CSR of X^T (tpack.hip).
//
// r02: no packed copy any more.  r01 re-laid every operand out as 128-byte chunks ("PCR16") so
// that the chunk stream could be fetched line by line; cutting a 16-slot window out of (current
// chunk ++ next chunk) cost two ds_bpermute and ~30 VALU per (row-set, slab) - 30 % of the launch
// - and building the copies cost a quarter of an lsi() call.  Here a 16-lane group reads the 16
// entries behind its row's cursor directly (two 64-byte pieces of indices[] and values[],
// unaligned), so the window IS the load result; what is left of stage A is a compare, a ballot,
// four scalar popcounts and the cursor update.  A line is requested about twice (the unconsumed
// tail of a window is loaded again one slab later, from L2 / Infinity Cache).
//
// Kernel structure (one 1024-thread workgroup = 64*K rows, K <= 8, per CU):
//   * the columns of X are swept in slabs of 256; the slab's 256 Q rows (64 KiB at B = 64) are
//     copied to LDS by LDS-DMA, double buffered;
//   * a wave is four 16-lane groups, group g walks the row at position 4k+g of row-set k and keeps
//     its K accumulators in registers; lane `sub` owns dense columns NB*sub .. NB*sub+NB-1;
//   * rows are sorted, so the entries of this slab are a prefix of the window: count them per
//     group (ballot + s_bcnt1), advance the cursor, and request the next window right away - it
//     arrives a full slab sweep later (EXEC-masked global_load_dword x 2 from inline asm: exactly
//     two VMEM instructions per row-set and slab, so completion is tracked with counted
//     s_waitcnt vmcnt instead of the vmcnt(0) hipcc falls back to for conditional loads);
//   * entry e's (LDS address, value) is broadcast inside the group with DPP row_newbcast, one
//     ds_read_b128 serves four rows, FMAs in f32.
//
// Row order: position p of the launch handles row perm[p] (-1: none); the host sorts the rows by
// length and deals them round robin (muon_amd/_backend.py: spmm_layout) - nothing is moved in
// memory, the positions only decide which rows share a wave.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.hpp"

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

// NB = dense columns a lane owns (B = 16 NB): 4 (ds_read_b128), 2 (b64), 1 (b32)
template <int NB> struct Vec;
template <> struct Vec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<1> { typedef float type __attribute__((ext_vector_type(1))); };

template <int NB> struct Quad { typename Vec<NB>::type q0, q1, q2, q3; };

// Entries E..E+3 of every group's window: four independent LDS reads.  `base` is the LDS byte
// address of the slab buffer plus this lane's column offset; slots a group does not use carry
// v = 0 (a read of some row of the slab and FMAs with zero).
template <int E, int NB>
__device__ __forceinline__ Quad<NB> quad_read(unsigned base, int a) {
  typedef __attribute__((address_space(3))) const typename Vec<NB>::type* lds_p;
  const unsigned a0 = (unsigned)bcast_i<E>(a) + base, a1 = (unsigned)bcast_i<E + 1>(a) + base;
  const unsigned a2 = (unsigned)bcast_i<E + 2>(a) + base, a3 = (unsigned)bcast_i<E + 3>(a) + base;
  Quad<NB> r;
  r.q0 = *(lds_p)(a0);
  r.q1 = *(lds_p)(a1);
  r.q2 = *(lds_p)(a2);
  r.q3 = *(lds_p)(a3);
  return r;
}
template <int E, int NB>
__device__ __forceinline__ void quad_fma(const Quad<NB>& r, float v, typename Vec<NB>::type& acc) {
  const float v0 = bcast_f<E>(v), v1 = bcast_f<E + 1>(v);
  const float v2 = bcast_f<E + 2>(v), v3 = bcast_f<E + 3>(v);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = fmaf(v0, r.q0[c], acc[c]);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = fmaf(v1, r.q1[c], acc[c]);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = fmaf(v2, r.q2[c], acc[c]);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = fmaf(v3, r.q3[c], acc[c]);
}

// Entries E, E+1 only: the upper half of the window is used by few groups (8 entries per row and
// slab on the bench matrices), so it is gated pair by pair instead of quad by quad.
template <int NB> struct Pair { typename Vec<NB>::type q0, q1; };
template <int E, int NB>
__device__ __forceinline__ Pair<NB> pair_read(unsigned base, int a) {
  typedef __attribute__((address_space(3))) const typename Vec<NB>::type* lds_p;
  const unsigned a0 = (unsigned)bcast_i<E>(a) + base, a1 = (unsigned)bcast_i<E + 1>(a) + base;
  Pair<NB> r;
  r.q0 = *(lds_p)(a0);
  r.q1 = *(lds_p)(a1);
  return r;
}
template <int E, int NB>
__device__ __forceinline__ void pair_fma(const Pair<NB>& r, float v, typename Vec<NB>::type& acc) {
  const float v0 = bcast_f<E>(v), v1 = bcast_f<E + 1>(v);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = fmaf(v0, r.q0[c], acc[c]);
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = fmaf(v1, r.q1[c], acc[c]);
}

// one LDS-DMA piece: 64 lanes x 16 B land contiguously at the wave-uniform LDS byte address
__device__ __forceinline__ void dma_piece(const float4* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
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
// (column, value); the others get the padding column.  Always exactly two VMEM instructions, also
// when the mask is empty: gfx950 counts a VMEM instruction issued with EXEC = 0 in order
// (scripts/probes/exec0_vmcnt.hip), which is what the counted waits rely on.
template <int k>
__device__ __forceinline__ void request_window(const void* pidx, const void* pval,
                                               unsigned long long mask) {
  unsigned long long save;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "v_mov_b32 v%c4, 0x7fffffff\n\t"
      "s_and_b64 exec, exec, %3\n\t"
      "global_load_dword v%c4, %1, off\n\t"
      "global_load_dword v%c5, %2, off\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(save)
      : "v"(pidx), "v"(pval), "s"(mask), "i"(kNX + 2 * k), "i"(kNX + 2 * k + 1)
      : MU_WIN_CLOB, "scc");  // s_and_b64 writes SCC: hipcc does keep compares alive across the asm
}

// Same for the pair stream (8 bytes per entry: column, value bits): one global_load_dwordx2.
template <int k>
__device__ __forceinline__ void request_window_pairs(const void* pent, unsigned long long mask) {
  unsigned long long save;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "v_mov_b32 v%c3, 0x7fffffff\n\t"
      "s_and_b64 exec, exec, %2\n\t"
      "global_load_dwordx2 v[%c3:%c4], %1, off\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(save)
      : "v"(pent), "s"(mask), "i"(kNX + 2 * k), "i"(kNX + 2 * k + 1)
      : MU_WIN_CLOB, "scc");
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

struct Win {      // what stage A of a pass hands to stage B
  int a;          // lane e of a group: LDS byte offset (inside the slab) of window entry e
  float vv;       // lane e: value of window entry e, 0 if the entry is not of this slab
  unsigned any16; // bit e: some group of the wave uses window entry e
};

// MODE is 0 in production; the other bits switch parts of the kernel off for timing ablations
// (results are then wrong on purpose): 1 no LDS gathers / FMAs, 2 no slab DMA, 8 no window
// requests (and no overflow passes).
// PAIRS: `indices` points at the pair stream (ent[p] = column | value bits << 32, same row pointers as
// the CSR), `values` is unused; one 8-byte load per lane and window instead of two 4-byte ones.
template <int K, int MODE, int NB, bool PAIRS>
__device__ __forceinline__ void spmm_win_body(int64_t n_pos, int64_t n_cols,
                                              const int64_t* __restrict__ indptr,
                                              const int32_t* __restrict__ indices,
                                              const float* __restrict__ values,
                                              const int32_t* __restrict__ perm,
                                              const float* __restrict__ Q, float* __restrict__ Y) {
  static_assert(K >= 1 && K <= kKMax, "K out of range");
  constexpr int W = kWaves;
  typedef typename Vec<NB>::type acc_t;
  constexpr int kRowBytes = 64 * NB;                         // one Q row: 16 NB floats
  constexpr int kSlabBytes = kSlabCols * kRowBytes;          // 64 / 32 / 16 KiB
  constexpr int kRowShift = NB == 4 ? 8 : (NB == 2 ? 7 : 6);
  constexpr int kPieces = kSlabBytes / 1024;                 // 1 KiB LDS-DMA pieces per slab
  constexpr int kMyPieces = kPieces / W;                     // per wave and slab: 4 / 2 / 1
  static_assert(kPieces % W == 0, "every wave issues the same number of DMA pieces");
  // VMEM order of a wave in one slab: D0 R0 D1 R1 ... (DMA piece u of the NEXT slab goes out right
  // before pass u, the pieces a short K leaves over after the last pass; R = the window request of a
  // pass).  Between the request of (row-set k, slab s-1) and pass (k, s) that is always K - 1
  // requests and kMyPieces DMA pieces, whatever k - the last slab issues its (unused) pieces too -
  // so the wait for a window is exact: nothing younger is waited for.
  // eight LDS reads in flight need 32 result registers: they fit next to K <= 6 accumulator sets
  // (hipcc spills beyond - and scratch traffic would share vmcnt with the hand-counted requests)
  constexpr bool kDeep = (K * NB <= 24) && !(MODE & 32);
  constexpr int kPerPass = PAIRS ? 1 : 2;  // VMEM instructions of one window request
  constexpr int kWaitMain = kPerPass * (K - 1) + kMyPieces;
  // ... and after the last DMA piece come the requests of passes kMyPieces-1 .. K-1
  constexpr int kWaitSlab = K >= kMyPieces ? kPerPass * (K - kMyPieces + 1) : 0;
  __shared__ float4 qs[2][kSlabBytes / 16];  // double buffer; Q row c of a slab at byte c * kRowBytes
  const int lane = threadIdx.x & 63;
  const int wave = uniform32(threadIdx.x >> 6);
  const int sub = lane & 15, g = lane >> 4;
  const int g8 = g * 8;
  const int sub_off = sub * (4 * NB);  // this lane's byte offset inside a Q row
  const int64_t rb0 = (int64_t)blockIdx.x * (4 * W * K);
  const int64_t rb1 = (rb0 + 4 * W * K) < n_pos ? (rb0 + 4 * W * K) : n_pos;
  const float4* __restrict__ Q4 = reinterpret_cast<const float4*>(Q);
  const int64_t q4_total = n_cols * (4 * NB);
  const int ncols32 = (int)n_cols;
  const unsigned qs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&qs[0][0]);
  const int64_t vdelta = PAIRS ? 4 : reinterpret_cast<const char*>(values) - reinterpret_cast<const char*>(indices);
  constexpr int kEntShift = PAIRS ? 3 : 2;  // log2 of the stride of the stream `ptr` walks

  acc_t acc[K];
  const char* ptr[K];  // address of indices[cursor + sub] of (row-set k, this lane's group)
  int rem[K];          // entries of that row from the cursor on
  {
    // lane 16 g + k looks the row of (row-set k, group g) up; the others idle
    const int64_t p = rb0 + ((int64_t)wave * K + sub) * 4 + g;
    const bool ok = (sub < K) && (p < rb1);
    const int64_t row = ok ? (perm ? (int64_t)perm[p] : p) : -1;
    const int64_t lo = row >= 0 ? indptr[row] : 0;
    const int64_t hi = row >= 0 ? indptr[row + 1] : 0;
    const int lo_l = (int)(lo & 0xffffffffll), lo_h = (int)(lo >> 32);
    const int len = (int)(hi - lo);
    static_for<K>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
#pragma unroll
      for (int c = 0; c < NB; ++c) acc[k][c] = 0.f;
      const int64_t l = ((int64_t)bcast_i<k>(lo_h) << 32) | (int64_t)(unsigned)bcast_i<k>(lo_l);
      rem[k] = bcast_i<k>(len);
      ptr[k] = reinterpret_cast<const char*>(indices) + ((l + sub) << kEntShift);
      const bool in = sub < rem[k];
      const int c0 = in ? *reinterpret_cast<const int*>(ptr[k]) : kPadCol;
      const int v0 = in ? *reinterpret_cast<const int*>(ptr[k] + vdelta) : 0;
      set_window<k>(c0, v0);
    });
  }

  auto dma_one = [&](int64_t s0, int buf, int u) {           // 1 KiB piece u of this wave
    const int piece = wave + u * W;
    int64_t i = s0 * (4 * NB) + piece * 64 + lane;           // float4 index into Q
    if (i >= q4_total) i = q4_total - 1;                     // tail / past the end: clamp (never consumed)
    dma_piece(Q4 + i, qs_lds + (unsigned)buf * (unsigned)kSlabBytes + (unsigned)piece * 1024u);
  };
#pragma unroll
  for (int u = 0; u < kMyPieces; ++u) dma_one(0, 0, u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // MODE & 64: per-wave cycle accounting (s_memtime) of the four places a pass can spend time in;
  // the passes run unpipelined (A(k) then B(k)) and the sums replace the product in Y.
  unsigned t_wait = 0, t_a = 0, t_b = 0, t_bar = 0, t_dma = 0;
  auto now = [&]() -> unsigned { return (unsigned)__builtin_amdgcn_s_memtime(); };

  int buf = 0;
  for (int64_t s0 = 0; s0 < n_cols; s0 += kSlabCols, buf ^= 1) {
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

    // Stage A of a pass: the window of (row-set k, every group) has arrived; the entries of this
    // slab are a prefix of it.  Count them per group, advance the cursors, request the next window.
    // SLOW = overflow pass (a row had more than 16 entries in this slab): its request was issued
    // just now, so drain everything; a main pass only needs the request of the previous slab.
    auto stage_a = [&](auto kc, auto slowc) -> Win {
      constexpr int k = decltype(kc)::value;
      constexpr bool SLOW = decltype(slowc)::value;
      int col, valbits;
      unsigned tw0 = 0;
      if constexpr (MODE & 64) tw0 = now();
      if constexpr (SLOW) wait_window<k, 0>(col, valbits);
      else wait_window<k, kWaitMain>(col, valbits);
      if constexpr (MODE & 64) {
        const unsigned tw1 = now();
        t_wait += tw1 - tw0;
        t_a -= tw1;  // closed by the caller after the request
      }
      const bool valid = col < s_hi;  // sorted rows: a prefix; padding lanes hold INT_MAX
      const unsigned long long m = __ballot(valid);
      const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
      const unsigned c0 = __popc(mlo & 0xffffu), c1 = __popc(mlo >> 16);
      const unsigned c2 = __popc(mhi & 0xffffu), c3 = __popc(mhi >> 16);
      const unsigned packed = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);  // wave-uniform
      const unsigned mm = mlo | mhi;
      Win w;
      w.any16 = (mm | (mm >> 16)) & 0xffffu;  // bit e: some group has entry e
      w.a = (col & (kSlabCols - 1)) << kRowShift;  // always inside the slab buffer, valid or not
      w.vv = valid ? __builtin_bit_cast(float, valbits) : 0.f;
      if constexpr (!(MODE & 8)) {
        const int cnt = (int)((packed >> g8) & 0xffu);
        rem[k] -= cnt;
        ptr[k] += cnt << kEntShift;
        if constexpr (PAIRS) request_window_pairs<k>(ptr[k], __ballot(sub < rem[k]));
        else request_window<k>(ptr[k], ptr[k] + vdelta, __ballot(sub < rem[k]));
        if ((c0 | c1 | c2 | c3) & 16u) again |= 1u << k;  // a window used up: maybe more in this slab
      }
      return w;
    };
    // Stage B: the LDS gathers and FMAs of the window.
    auto stage_b = [&](auto kc, const Win& w) {
      constexpr int k = decltype(kc)::value;
      if constexpr (MODE & 1) {
        acc[k][0] += w.vv + (float)w.a;
      } else {
        if constexpr (kDeep) {
          if (w.any16 & 0x00f0u) {
            // the usual case (8 entries per row and slab): eight LDS reads in flight, one latency
            const Quad<NB> r0 = quad_read<0, NB>(qbase, w.a);
            const Quad<NB> r1 = quad_read<4, NB>(qbase, w.a);
            quad_fma<0, NB>(r0, w.vv, acc[k]);
            quad_fma<4, NB>(r1, w.vv, acc[k]);
          } else if (w.any16 & 0x000fu) {
            const Quad<NB> r = quad_read<0, NB>(qbase, w.a);
            quad_fma<0, NB>(r, w.vv, acc[k]);
          }
        } else {
          if (w.any16 & 0x000fu) { const Quad<NB> r = quad_read<0, NB>(qbase, w.a); quad_fma<0, NB>(r, w.vv, acc[k]); }
          if (w.any16 & 0x00f0u) { const Quad<NB> r = quad_read<4, NB>(qbase, w.a); quad_fma<4, NB>(r, w.vv, acc[k]); }
        }
        if (w.any16 & 0xff00u) {
          // sorted rows fill the window from slot 0: bit e set => every lower bit is set
          { const Pair<NB> r = pair_read<8, NB>(qbase, w.a); pair_fma<8, NB>(r, w.vv, acc[k]); }
          if (w.any16 & 0x0c00u) { const Pair<NB> r = pair_read<10, NB>(qbase, w.a); pair_fma<10, NB>(r, w.vv, acc[k]); }
          if (w.any16 & 0x3000u) { const Pair<NB> r = pair_read<12, NB>(qbase, w.a); pair_fma<12, NB>(r, w.vv, acc[k]); }
          if (w.any16 & 0xc000u) { const Pair<NB> r = pair_read<14, NB>(qbase, w.a); pair_fma<14, NB>(r, w.vv, acc[k]); }
        }
      }
    };

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
      // A(k+1) is issued before B(k): the scalar part of the next pass runs under this pass' gathers
      next_piece(0);
      Win w = stage_a(std::integral_constant<int, 0>{}, std::false_type{});
      static_for<K>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        Win wn = w;
        if constexpr (k + 1 < K) {
          if constexpr (k + 1 < kMyPieces) next_piece(k + 1);
          wn = stage_a(std::integral_constant<int, k + 1>{}, std::false_type{});
        }
        stage_b(kc, w);
        w = wn;
      });
    }
#pragma unroll
    for (int u = K; u < kMyPieces; ++u) next_piece(u);  // K < kMyPieces: the pieces left over
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
      } while (again);
      // an overflow request of row-set k is younger than the main-pass requests the next slab's
      // counted waits are set against: drain, so that the count only ever guards main-pass requests
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // The last DMA piece of the next slab went out before the requests of passes kMyPieces-1 ..
    // K-1: allowing that many outstanding VMEM operations proves the slab landed.
    unsigned tb0 = 0;
    if constexpr (MODE & 64) tb0 = now();
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kWaitSlab) : "memory");
    __syncthreads();  // next slab visible; everyone finished reading this one
    if constexpr (MODE & 64) t_bar += now() - tb0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requests still in flight target v[kNX ..]
  if constexpr (MODE & 64) {
    float dep = 0.f;  // (keeps the FMAs of stage B alive)
    static_for<K>([&](auto kc) {
#pragma unroll
      for (int c = 0; c < NB; ++c) dep += acc[decltype(kc)::value][c];
    });
    if (__ballot(dep == 1.2345e-30f)) t_b += 1;
    if (lane < 5) {
      const unsigned t = lane == 0 ? t_wait : lane == 1 ? t_a : lane == 2 ? t_b : lane == 3 ? t_bar : t_dma;
      Y[((int64_t)blockIdx.x * W + wave) * (16 * NB) + lane] = (float)t;
    }
    return;
  }
  static_for<K>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int64_t p = rb0 + ((int64_t)wave * K + k) * 4 + g;
    if (p < rb1) {
      const int64_t out = perm ? (int64_t)perm[p] : p;  // position -> row of the product (-1: none)
      if (out >= 0) *reinterpret_cast<acc_t*>(Y + out * (16 * NB) + sub * NB) = acc[k];
    }
  });
}

#define MU_KARGS                                                                                  \
  int64_t n_pos, int64_t n_cols, const int64_t *__restrict__ indptr,                              \
      const int32_t *__restrict__ indices, const float *__restrict__ values,                      \
      const int32_t *__restrict__ perm, const float *__restrict__ Q, float *__restrict__ Y
template <int K, int MODE, int NB, bool PAIRS>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(55))) void k_spmm_win(MU_KARGS) {
  spmm_win_body<K, MODE, NB, PAIRS>(n_pos, n_cols, indptr, indices, values, perm, Q, Y);
}

// pair stream of a CSR: ent[p] = (column, value bits); a streaming copy (8 B in, 8 B out per entry)
__global__ __launch_bounds__(256) void k_pairs_fill(int64_t nnz, const int32_t* __restrict__ indices,
                                                    const float* __restrict__ values,
                                                    unsigned long long* __restrict__ ent) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < nnz; i += stride) {
    if (i + 4 <= nnz) {
      const int4 c = *reinterpret_cast<const int4*>(indices + i);
      const float4 v = *reinterpret_cast<const float4*>(values + i);
      ulonglong2 a, b;
      a.x = (unsigned long long)(unsigned)c.x | ((unsigned long long)__builtin_bit_cast(unsigned, v.x) << 32);
      a.y = (unsigned long long)(unsigned)c.y | ((unsigned long long)__builtin_bit_cast(unsigned, v.y) << 32);
      b.x = (unsigned long long)(unsigned)c.z | ((unsigned long long)__builtin_bit_cast(unsigned, v.z) << 32);
      b.y = (unsigned long long)(unsigned)c.w | ((unsigned long long)__builtin_bit_cast(unsigned, v.w) << 32);
      *reinterpret_cast<ulonglong2*>(ent + i) = a;
      *reinterpret_cast<ulonglong2*>(ent + i + 2) = b;
    } else {
      for (int64_t j = i; j < nnz; ++j)
        ent[j] = (unsigned long long)(unsigned)indices[j] |
                 ((unsigned long long)__builtin_bit_cast(unsigned, values[j]) << 32);
    }
  }
}

// K row-sets per wave: the smallest number of full-chip rounds R whose 64*K-row blocks fit the
// register budget (K <= kKMax); one workgroup per CU (128 KiB of LDS at B = 64).
int pick_k(int64_t n_rows) {
  const int64_t cus = mu_num_cus();
  for (int64_t R = 1; R <= 4096; ++R) {
    const int64_t k = (n_rows + 64 * cus * R - 1) / (64 * cus * R);
    if (k <= kKMax) return (int)(k < 1 ? 1 : k);
  }
  return kKMax;
}

template <int K, int MODE, bool PAIRS>
int launch(int B, hipStream_t st, int64_t n_pos, int64_t n_cols, const int64_t* indptr,
           const int32_t* indices, const float* values, const int32_t* perm, const float* Q, float* Y) {
  const int64_t wgs = (n_pos + 64 * K - 1) / (64 * K);
  if (B == 64)
    hipLaunchKernelGGL((k_spmm_win<K, MODE, 4, PAIRS>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols,
                       indptr, indices, values, perm, Q, Y);
  else if (B == 32)
    hipLaunchKernelGGL((k_spmm_win<K, MODE, 2, PAIRS>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols,
                       indptr, indices, values, perm, Q, Y);
  else
    hipLaunchKernelGGL((k_spmm_win<K, MODE, 1, PAIRS>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols,
                       indptr, indices, values, perm, Q, Y);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

template <bool PAIRS>
int dispatch(int64_t n_pos, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices,
             const float* d_values, const int32_t* d_perm, int k_layout, const float* d_Q, int B,
             float* d_Y, hipStream_t st) {
  int K = (k_layout >= 1 && k_layout <= kKMax) ? k_layout : pick_k(n_pos);
  const int force_k = mu_tune_get("spmm_k");  // tests / tuning only (mu_tune_set); 0 in production
  if (force_k >= 1 && force_k <= kKMax) K = force_k;
  const int mode = mu_tune_get("spmm_mode");
#define MU_ARGS B, st, n_pos, n_cols, d_indptr, d_indices, d_values, d_perm, d_Q, d_Y
  if (mode != 0) {
    if (K != 8 && K != 4 && K != 6) {
      mu_set_error("ablation modes exist for K = 4, 6 and 8 only");
      return MU_ERR_ARG;
    }
    switch (mode + 100 * K) {
      case 801: return launch<8, 1, PAIRS>(MU_ARGS);
      case 809: return launch<8, 9, PAIRS>(MU_ARGS);
      case 864: return launch<8, 64, PAIRS>(MU_ARGS);
      case 601: return launch<6, 1, PAIRS>(MU_ARGS);
      case 664: return launch<6, 64, PAIRS>(MU_ARGS);
      case 632: return launch<6, 32, PAIRS>(MU_ARGS);
      case 696: return launch<6, 96, PAIRS>(MU_ARGS);
      case 401: return launch<4, 1, PAIRS>(MU_ARGS);
      case 409: return launch<4, 9, PAIRS>(MU_ARGS);
      case 464: return launch<4, 64, PAIRS>(MU_ARGS);
      default: break;
    }
    mu_set_error("spmm_mode %d has no compiled instance", mode);
    return MU_ERR_ARG;
  }
  switch (K) {
    case 1: return launch<1, 0, PAIRS>(MU_ARGS);
    case 2: return launch<2, 0, PAIRS>(MU_ARGS);
    case 3: return launch<3, 0, PAIRS>(MU_ARGS);
    case 4: return launch<4, 0, PAIRS>(MU_ARGS);
    case 5: return launch<5, 0, PAIRS>(MU_ARGS);
    case 6: return launch<6, 0, PAIRS>(MU_ARGS);
    case 7: return launch<7, 0, PAIRS>(MU_ARGS);
    default: return launch<8, 0, PAIRS>(MU_ARGS);
  }
#undef MU_ARGS
}

}  // namespace

extern "C" {

int mu_spmm_csr_k(int64_t n_rows) { return pick_k(n_rows); }

int mu_spmm_csr_f32(int64_t n_pos, int64_t n_cols, const int64_t* d_indptr, const int32_t* d_indices,
                    const float* d_values, const int32_t* d_perm, int k_layout, const float* d_Q, int B,
                    float* d_Y, void* stream) {
  MU_REQUIRE(B == 16 || B == 32 || B == 64, "B must be 16, 32 or 64");
  MU_REQUIRE(n_pos >= 0 && n_cols > 0 && n_cols < ((int64_t)1 << 31), "shape out of range");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_indices && d_values && d_Q && d_Y, "null pointer");
  return dispatch<false>(n_pos, n_cols, d_indptr, d_indices, d_values, d_perm, k_layout, d_Q, B, d_Y,
                         (hipStream_t)stream);
}

int mu_csr_pairs_fill(int64_t nnz, const int32_t* d_indices, const float* d_values, void* d_ent,
                      void* stream) {
  MU_REQUIRE(nnz >= 0, "negative size");
  if (nnz == 0) return MU_OK;
  MU_REQUIRE(d_indices && d_values && d_ent, "null pointer");
  int64_t blocks = (nnz / 4 + 255) / 256;
  const int64_t cap = (int64_t)mu_num_cus() * 32;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_pairs_fill, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, nnz,
                     d_indices, d_values, (unsigned long long*)d_ent);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_spmm_pairs_f32(int64_t n_pos, int64_t n_cols, const int64_t* d_indptr, const void* d_ent,
                      const int32_t* d_perm, int k_layout, const float* d_Q, int B, float* d_Y,
                      void* stream) {
  MU_REQUIRE(B == 16 || B == 32 || B == 64, "B must be 16, 32 or 64");
  MU_REQUIRE(n_pos >= 0 && n_cols > 0 && n_cols < ((int64_t)1 << 31), "shape out of range");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_ent && d_Q && d_Y, "null pointer");
  return dispatch<true>(n_pos, n_cols, d_indptr, (const int32_t*)d_ent, nullptr, d_perm, k_layout, d_Q, B,
                        d_Y, (hipStream_t)stream);
}

}  // extern "C"
