// SpMM on the packed chunked-row copy of a CSR (B = 64, f32):  Y[n x 64] = X[n x d] * Q[d x 64].
//
// This is the kernel the block subspace iteration of muon_amd.atac.tl.lsi spends its time in; it
// stands where ARPACK's reverse-communication loop calls csr_matvec / csr_matvecs through
// scipy.sparse.linalg.svds (/root/reference/muon/_atac/tools.py:53, scipy _svds.py:441-466,516).
//
// Layout ("PCR16", built once per lsi() call by mu_csr_pack_*):
//   * every row is cut into chunks of 16 (column, value) pairs, 8 bytes per pair, so a chunk is
//     exactly one aligned 128-byte line; the last chunk is padded with (0x7fffffff, 0) and one
//     all-padding chunk closes every row (so "the next chunk" always exists);
//   * cptr int64[n+1] are chunk offsets.
// A 16-lane group streams one row: it holds the current and the next chunk in registers, one
// pair per lane, and every line of the matrix is fetched exactly once, a full slab sweep ahead
// of its use.
//
// Kernel structure (one 1024-thread workgroup = 64*K rows, K <= 8, per CU):
//   * the columns of X are swept in slabs of 256; the slab's 256 Q rows (64 KiB) are copied to
//     LDS by LDS-DMA, double buffered;
//   * a wave is four 16-lane groups, group g walks row 4k+g of row-set k and keeps its K float4
//     accumulators in registers; lane `sub` owns dense columns 4 sub .. 4 sub + 3;
//   * per (row-set, slab) the 16-slot window starting at the row's cursor is cut out of
//     (current chunk ++ next chunk) with a select and a ds_bpermute rotation, the entries that
//     fall into the slab are a prefix of it (sorted rows), and entry e's (LDS address, value) is
//     broadcast inside the group with DPP row_newbcast, so one ds_read_b128 serves four rows;
//   * a group that used up its current chunk promotes the next one and requests the chunk after
//     it with an EXEC-masked global_load_dwordx2 (inline asm: exactly one VMEM instruction per
//     row-set and slab, so completion is tracked with counted s_waitcnt vmcnt(K-1) instead of
//     the vmcnt(0) the compiler has to fall back to for conditionally issued loads).
//
// All VMEM traffic of the main loop is issued from inline asm (hipcc neither counts nor waits for
// it): the LDS-DMA pieces and the chunk requests.  In-order completion makes the counted waits
// safe however many extra (overflow-pass) requests are interleaved.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.hpp"

namespace {

constexpr int kSlabCols = 256;   // Q rows per slab: 256 x 256 B = 64 KiB, double buffered
constexpr int kKMaxAny = 16;
constexpr int kPadCol = 0x7fffffff;

// Geometry per workgroup size.  A workgroup of W waves owns 4*W*K rows; one workgroup per CU (the
// two slab buffers take 128 KiB of LDS), so W fixes the register budget per lane: 512 / (W / 4).
// hipcc is capped at v[0 .. NX-1] by amdgpu_num_vgpr(NX / 2) on the kernel wrappers; the asm
// statements own v[NX .. NX + 2 KMAX] (see below).
// (768- and 512-thread workgroups - 168 / 256 registers per lane, K up to 11 / 16, two quads of LDS
//  reads in flight - were built and measured 3...20 % slower than 1024 threads: DESIGN.md 4.2)
template <int W> struct Geo;
template <> struct Geo<16> { static constexpr int NX = 110, KMAX = 8; };   // 128 regs: 110 + 17

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
// a = 0, v = 0 (a broadcast read of slab row 0 and FMAs with zero).
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

// The next chunk of every (row-set, group) lives in v[NX + 2k], v[NX + 2k + 1] (column, value
// bits).  These registers are written by the asm chunk requests while the wave keeps running, so
// they must never be visible to hipcc as values: a compiler-made copy of a register whose load
// is still in flight reads stale data (it did happen with "+v" operands).  The kernel wrappers
// are compiled with amdgpu_num_vgpr(NX / 2) - on the unified gfx950 register file that caps
// hipcc's own allocation at v[0 .. NX-1] - and the asm statements name the registers above
// literally; the clobber lists make the kernel descriptor allocate them (hipcc warns that they
// are "reserved", which is the point).  tests/test_layout.py audits the generated ISA.
#pragma clang diagnostic ignored "-Winline-asm"
#define MU_CLOB_16                                                                                \
  "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", \
      "v122", "v123", "v124", "v125", "v126"

// EXEC-masked chunk request: lanes of `mask` overwrite their pair, the others keep it.  Always
// exactly one VMEM instruction, also when the mask is empty: on gfx950 a VMEM instruction issued
// with EXEC = 0 still takes part in the in-order vmcnt accounting (scripts/probes/
// exec0_vmcnt.hip: 524288/524288 lanes read the loaded value behind eight such loads and
// vmcnt(8), 0/524288 in the control arm), which is what the counted waits below rely on.
#define MU_REQUEST_ASM(CLOB)                                                          \
  asm volatile(                                                                       \
      "s_mov_b64 %0, exec\n\t"                                                        \
      "s_and_b64 exec, exec, %3\n\t"                                                  \
      "global_load_dwordx2 v[%c4:%c5], %1, %2\n\t"                                    \
      "s_mov_b64 exec, %0"                                                            \
      : "=&s"(save)                                                                   \
      : "v"(byte_off), "s"(base), "s"(mask), "i"(NX + 2 * k), "i"(NX + 2 * k + 1)     \
      : CLOB, "scc")
template <int W, int k>
__device__ __forceinline__ void request_chunk(unsigned byte_off, const void* base,
                                              unsigned long long mask) {
  constexpr int NX = Geo<W>::NX;
  unsigned long long save;
  static_assert(W == 16, "register ownership is laid out for 1024-thread workgroups");
  MU_REQUEST_ASM(MU_CLOB_16);
}

// wait until at most N VMEM operations are outstanding, then read the next chunk of row-set k
#define MU_WAIT_ASM(CLOB)                                                  \
  asm volatile(                                                            \
      "s_waitcnt vmcnt(%c2)\n\t"                                           \
      "v_mov_b32 %0, v%c3\n\t"                                             \
      "v_mov_b32 %1, v%c4"                                                 \
      : "=v"(col), "=v"(valbits)                                           \
      : "i"(N), "i"(NX + 2 * k), "i"(NX + 2 * k + 1)                       \
      : CLOB)
template <int W, int k, int N>
__device__ __forceinline__ void wait_next_chunk(int& col, int& valbits) {
  constexpr int NX = Geo<W>::NX;
  MU_WAIT_ASM(MU_CLOB_16);
}

#define MU_SET_ASM(CLOB)                                                   \
  asm volatile(                                                            \
      "v_mov_b32 v%c2, %0\n\t"                                             \
      "v_mov_b32 v%c3, %1"                                                 \
      :                                                                    \
      : "v"(col), "v"(valbits), "i"(NX + 2 * k), "i"(NX + 2 * k + 1)       \
      : CLOB)
template <int W, int k>
__device__ __forceinline__ void set_next_chunk(int col, int valbits) {
  constexpr int NX = Geo<W>::NX;
  MU_SET_ASM(MU_CLOB_16);
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
  int a;          // lane e of a group: LDS byte offset (inside the slab) of window entry e, 0 if unused
  float vv;       // lane e: value of window entry e, 0 if unused
  unsigned any16; // bit e: some group of the wave uses window entry e
};

// MODE is 0 in production; the other bits switch parts of the kernel off for timing ablations
// (results are then wrong on purpose, except 4): 1 no LDS gathers / FMAs, 2 no slab DMA, 4 upper window
// half gated by quads as in r01a-n (right results, 2-3 % slower), 8 no chunk
// requests (and no overflow passes), 16 no window rotation, 32 overflow passes do not wait for their chunk.
// PIPE: 0 = stage A(k) then B(k); 1 = A(k+1) is issued before B(k) (the dispatcher's choice; tune knob
// spmm_pipe = 1 selects PIPE 0 for comparison).
template <int W, int K, int MODE, int PIPE, int NB>
__device__ __forceinline__ void spmm_pcr64_body(int64_t n_rows, int64_t n_cols,
                                                const int64_t* __restrict__ cptr,
                                                const unsigned long long* __restrict__ ent,
                                                const int32_t* __restrict__ perm,
                                                const float* __restrict__ Q, float* __restrict__ Y) {
  static_assert(K >= 1 && K <= Geo<W>::KMAX && K <= 16, "K out of range for this workgroup size");
  typedef typename Vec<NB>::type acc_t;
  constexpr int kRowBytes = 64 * NB;                         // one Q row: 16 NB floats
  constexpr int kSlabBytes = kSlabCols * kRowBytes;          // 64 / 32 / 16 KiB
  constexpr int kRowShift = NB == 4 ? 8 : (NB == 2 ? 7 : 6);
  constexpr int kPieces = kSlabBytes / 1024;                 // 1 KiB LDS-DMA pieces per slab
  __shared__ float4 qs[2][kSlabBytes / 16];  // double buffer; Q row c of a slab at byte c * kRowBytes
  const int lane = threadIdx.x & 63;
  const int wave = uniform32(threadIdx.x >> 6);
  const int sub = lane & 15, g = lane >> 4;
  const int sub_off = sub * (4 * NB);  // this lane's byte offset inside a Q row
  const int rot_base = (lane & 48) << 2;  // ds_bpermute byte address of the group's lane 0
  const int64_t rb0 = (int64_t)blockIdx.x * (4 * W * K);
  const int64_t rb1 = (rb0 + 4 * W * K) < n_rows ? (rb0 + 4 * W * K) : n_rows;
  const int64_t cbase = uniform64(cptr[rb0]);
  const unsigned long long* __restrict__ entb = ent + cbase * 16;  // wave-uniform
  const float4* __restrict__ Q4 = reinterpret_cast<const float4*>(Q);
  const int64_t q4_total = n_cols * (4 * NB);
  const int ncols32 = (int)n_cols;
  const unsigned qs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&qs[0][0]);

  acc_t acc[K];
  int cc[K], cv[K];  // current chunk: column / value bits of entry `sub`
  int posv;          // lane 16 g + k: consumed entries of the current chunk of (row-set k, group g)
  int cidv;          // lane 16 g + k: chunk to request next (relative to the workgroup's first chunk)
  int lastv;         // lane 16 g + k: the row's closing (all padding) chunk
  int hasv;          // lane 16 g + k: first chunk of the row, -1 if there is no such row
  {
    const int64_t row = rb0 + ((int64_t)wave * K + sub) * 4 + g;
    const bool ok = (sub < K) && (row < rb1);
    const int c0 = ok ? (int)(cptr[row] - cbase) : 0;
    const int c1 = ok ? (int)(cptr[row + 1] - cbase) : 1;
    posv = 0;
    lastv = c1 - 1;
    cidv = (c0 + 2) < (c1 - 1) ? (c0 + 2) : (c1 - 1);
    hasv = ok ? c0 : -1;
  }
  // prologue: chunk 0 and chunk min(1, last) of every row (plain loads, hipcc waits for them)
  static_for<K>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
#pragma unroll
    for (int c = 0; c < NB; ++c) acc[k][c] = 0.f;
    const int c0 = bcast_i<k>(hasv);
    const int l = bcast_i<k>(lastv);
    const bool has = c0 >= 0;
    const int a = has ? c0 : 0;
    const int b = has ? ((c0 + 1) < l ? (c0 + 1) : l) : 0;
    const unsigned long long e0 = entb[(int64_t)a * 16 + sub];
    const unsigned long long e1 = entb[(int64_t)b * 16 + sub];
    cc[k] = has ? (int)(unsigned)e0 : kPadCol;
    cv[k] = (int)(unsigned)(e0 >> 32);
    set_next_chunk<W, k>(has ? (int)(unsigned)e1 : kPadCol, (int)(unsigned)(e1 >> 32));
  });

  auto slab_dma = [&](int64_t s0, int buf) {
    for (int piece = wave; piece < kPieces; piece += W) {  // 1 KiB pieces of the slab
      int64_t i = s0 * (4 * NB) + piece * 64 + lane;       // float4 index into Q
      if (i >= q4_total) i = q4_total - 1;                 // tail slab: clamp (never consumed)
      dma_piece(Q4 + i, qs_lds + (unsigned)buf * (unsigned)kSlabBytes + (unsigned)piece * 1024u);
    }
  };
  constexpr int kMyPiecesMax = (kPieces + W - 1) / W;  // DMA pieces one wave issues per slab
  slab_dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int buf = 0;
  for (int64_t s0 = 0; s0 < n_cols; s0 += kSlabCols, buf ^= 1) {
    if ((s0 + kSlabCols) < n_cols && !(MODE & 2)) slab_dma(s0 + kSlabCols, buf ^ 1);  // lands while this slab is consumed
    const int s_lo = (int)s0;
    const int s_hi = (s_lo + kSlabCols) < ncols32 ? (s_lo + kSlabCols) : ncols32;
    const unsigned qbase = qs_lds + (unsigned)buf * (unsigned)kSlabBytes + (unsigned)sub_off;
    unsigned again = 0;

    // Stage A of a pass: cut the 16-slot window of (row-set k, every group) out of (current chunk ++
    // next chunk), find the prefix that belongs to this slab, advance the cursors, request the next
    // chunk.  SLOW = overflow pass (a row had more than 16 entries in this slab): its request was
    // issued just now, so drain everything; a main pass only needs the request of the previous slab.
    auto stage_a = [&](auto kc, auto slowc) -> Win {
      constexpr int k = decltype(kc)::value;
      constexpr bool SLOW = decltype(slowc)::value;
      int ncol, nval;
      if constexpr (SLOW && !(MODE & 32)) wait_next_chunk<W, k, 0>(ncol, nval);
      else if constexpr (SLOW) wait_next_chunk<W, k, 63>(ncol, nval);  // ablation: no wait (wrong data)
      else wait_next_chunk<W, k, K - 1>(ncol, nval);
      const int p = bcast_i<k>(posv);
      const bool from_cur = sub >= p;
      const int mc = from_cur ? cc[k] : ncol;
      const int mv = from_cur ? cv[k] : nval;
      const int src = rot_base + (((sub + p) & 15) << 2);
      const int wc = (MODE & 16) ? mc : __builtin_amdgcn_ds_bpermute(src, mc);
      const int wv = (MODE & 16) ? mv : __builtin_amdgcn_ds_bpermute(src, mv);
      const bool valid = wc < s_hi;  // sorted rows: the slab's entries are a prefix
      const unsigned long long m = __ballot(valid);
      const int cnt = __popc((unsigned)(m >> (16 * g)) & 0xffffu);
      const unsigned mm = (unsigned)m | (unsigned)(m >> 32);
      Win w;
      w.any16 = (mm | (mm >> 16)) & 0xffffu;  // bit e: some group has entry e
      w.a = valid ? ((wc - s_lo) << kRowShift) : 0;
      w.vv = valid ? __builtin_bit_cast(float, wv) : 0.f;
      const int np = p + cnt;
      const bool shift = np >= 16;
      const unsigned long long smask = (MODE & 8) ? 0ull : __ballot(shift);
      cc[k] = shift ? ncol : cc[k];
      cv[k] = shift ? nval : cv[k];
      const int cid = bcast_i<k>(cidv);
      request_chunk<W, k>(((unsigned)cid << 7) | ((unsigned)sub << 3), entb, smask);
      if (sub == k) {
        posv = np & 15;
        cidv = shift ? (cidv < lastv ? cidv + 1 : lastv) : cidv;
      }
      if (__ballot(cnt == 16) && !(MODE & 8)) again |= 1u << k;  // window used up: maybe more in this slab
      return w;
    };
    // Stage B: the LDS gathers and FMAs of the window.
    auto stage_b = [&](auto kc, const Win& w) {
      constexpr int k = decltype(kc)::value;
      if constexpr (MODE & 1) {
        acc[k][0] += w.vv + (float)w.a;
      } else {
        { const Quad<NB> r = quad_read<0, NB>(qbase, w.a); quad_fma<0, NB>(r, w.vv, acc[k]); }
        if (w.any16 & 0x00f0u) { const Quad<NB> r = quad_read<4, NB>(qbase, w.a); quad_fma<4, NB>(r, w.vv, acc[k]); }
        if constexpr (MODE & 4) {
          if (w.any16 & 0x0f00u) { const Quad<NB> r = quad_read<8, NB>(qbase, w.a); quad_fma<8, NB>(r, w.vv, acc[k]); }
          if (w.any16 & 0xf000u) { const Quad<NB> r = quad_read<12, NB>(qbase, w.a); quad_fma<12, NB>(r, w.vv, acc[k]); }
        } else if (w.any16 & 0xff00u) {
          // sorted rows fill the window from slot 0: bit e set => every lower bit is set
          { const Pair<NB> r = pair_read<8, NB>(qbase, w.a); pair_fma<8, NB>(r, w.vv, acc[k]); }
          if (w.any16 & 0x0c00u) { const Pair<NB> r = pair_read<10, NB>(qbase, w.a); pair_fma<10, NB>(r, w.vv, acc[k]); }
          if (w.any16 & 0x3000u) { const Pair<NB> r = pair_read<12, NB>(qbase, w.a); pair_fma<12, NB>(r, w.vv, acc[k]); }
          if (w.any16 & 0xc000u) { const Pair<NB> r = pair_read<14, NB>(qbase, w.a); pair_fma<14, NB>(r, w.vv, acc[k]); }
        }
      }
    };

    if constexpr (PIPE == 1) {
      Win w = stage_a(std::integral_constant<int, 0>{}, std::false_type{});
      static_for<K>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        Win wn = w;
        if constexpr (k + 1 < K) wn = stage_a(std::integral_constant<int, k + 1>{}, std::false_type{});
        stage_b(kc, w);
        w = wn;
      });
    } else {
      static_for<K>([&](auto kc) {
        const Win w = stage_a(kc, std::false_type{});
        stage_b(kc, w);
      });
    }
    if (again) {
      do {
        const unsigned pend = again;
        again = 0;
        static_for<K>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if (pend & (1u << k)) {
            const Win w = stage_a(kc, std::true_type{});
            stage_b(kc, w);
          }
        });
      } while (again);
      // an overflow request of row-set k is younger than the main-pass requests the next slab's
      // vmcnt(K-1) is counted against: drain, so that the count only ever guards main-pass requests
      if constexpr (!(MODE & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // The DMA pieces of the next slab were issued before this slab's K requests: allowing K
    // outstanding VMEM operations proves they landed without draining the requests.
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(K) : "memory");
    __syncthreads();  // next slab visible; everyone finished reading this one
  }
  (void)kMyPiecesMax;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requests still in flight target v[NX ..]
  static_for<K>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int64_t row = rb0 + ((int64_t)wave * K + k) * 4 + g;
    if (row < rb1) {
      const int64_t out = perm ? (int64_t)perm[row] : row;  // position -> row of the product (-1: none)
      if (out >= 0) *reinterpret_cast<acc_t*>(Y + out * (16 * NB) + sub * NB) = acc[k];
    }
  });
}

#define MU_KARGS                                                                          \
  int64_t n_rows, int64_t n_cols, const int64_t *__restrict__ cptr,                       \
      const unsigned long long *__restrict__ ent, const int32_t *__restrict__ perm,       \
      const float *__restrict__ Q, float *__restrict__ Y
template <int K, int MODE, int PIPE, int NB>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(55))) void k_spmm_pcr64_w16(MU_KARGS) {
  spmm_pcr64_body<16, K, MODE, PIPE, NB>(n_rows, n_cols, cptr, ent, perm, Q, Y);
}
// ---- packing -----------------------------------------------------------------------------
// Position p of the packed copy holds row perm[p] of the matrix (perm == nullptr: the identity;
// perm[p] < 0: no row, only the closing chunk).  The host deals the rows, sorted by length, round
// robin to workgroups and waves (muon_amd/_backend.py: packed_layout) - see DESIGN.md 4.1.
__global__ __launch_bounds__(256) void k_pack_count(int64_t n_pos, const int32_t* __restrict__ perm,
                                                    const int64_t* __restrict__ indptr,
                                                    int64_t* __restrict__ row_chunks) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pos) return;
  const int64_t r = perm ? (int64_t)perm[p] : p;
  row_chunks[p] = (r < 0) ? 1 : ((indptr[r + 1] - indptr[r] + 15) >> 4) + 1;
}

// one 16-lane group per chunk would waste the closing chunks; a wave per row streams instead
__global__ __launch_bounds__(256) void k_pack_fill(int64_t n_pos, const int32_t* __restrict__ perm,
                                                   const int64_t* __restrict__ indptr,
                                                   const int32_t* __restrict__ indices,
                                                   const float* __restrict__ values,
                                                   const int64_t* __restrict__ cptr,
                                                   unsigned long long* __restrict__ ent) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = uniform64(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t pos = wave0; pos < n_pos; pos += n_waves) {
    const int64_t row = perm ? (int64_t)uniform32(perm[pos]) : pos;
    const int64_t lo = row < 0 ? 0 : uniform64(indptr[row]);
    const int64_t hi = row < 0 ? 0 : uniform64(indptr[row + 1]);
    const int64_t o0 = uniform64(cptr[pos]) * 16, o1 = uniform64(cptr[pos + 1]) * 16;
    for (int64_t j = lane; j < o1 - o0; j += 64) {
      const int64_t p = lo + j;
      unsigned long long e = (unsigned long long)(unsigned)kPadCol;
      if (p < hi)
        e = (unsigned long long)(unsigned)indices[p] |
            ((unsigned long long)__builtin_bit_cast(unsigned, values[p]) << 32);
      ent[o0 + j] = e;
    }
  }
}

// K row-sets per wave: the smallest number of full-chip rounds R whose 4*W*K-row blocks fit the
// register budget (K <= KMAX); one workgroup per CU (128 KiB of LDS).
template <int W>
int pick_k(int64_t n_rows) {
  const int64_t cus = mu_num_cus();
  for (int64_t R = 1; R <= 4096; ++R) {
    const int64_t k = (n_rows + 4 * W * cus * R - 1) / (4 * W * cus * R);
    if (k <= Geo<W>::KMAX) return (int)(k < 1 ? 1 : k);
  }
  return Geo<W>::KMAX;
}

#define MU_GO(KERNEL, W, KK, M, P)                                                              \
  {                                                                                             \
    const int64_t wgs = (n_rows + 4 * W * KK - 1) / (4 * W * KK);                               \
    if (B == 64)                                                                                \
      hipLaunchKernelGGL((KERNEL<KK, M, P, 4>), dim3((unsigned)wgs), dim3(64 * W), 0, st, n_rows,   \
                         n_cols, cptr, ent, perm, Q, Y);                                        \
    else if (B == 32)                                                                           \
      hipLaunchKernelGGL((KERNEL<KK, M, P, 2>), dim3((unsigned)wgs), dim3(64 * W), 0, st, n_rows,   \
                         n_cols, cptr, ent, perm, Q, Y);                                        \
    else                                                                                        \
      hipLaunchKernelGGL((KERNEL<KK, M, P, 1>), dim3((unsigned)wgs), dim3(64 * W), 0, st, n_rows,   \
                         n_cols, cptr, ent, perm, Q, Y);                                        \
    MU_CHECK_LAUNCH();                                                                          \
    return MU_OK;                                                                               \
  }

}  // namespace


extern "C" {

int mu_spmm_packed_k(int64_t n_rows) { return pick_k<16>(n_rows); }

int mu_csr_pack_count(int64_t n_pos, const int32_t* d_perm, const int64_t* d_indptr,
                      int64_t* d_row_chunks, void* stream) {
  MU_REQUIRE(n_pos >= 0, "negative size");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_row_chunks, "null pointer");
  hipLaunchKernelGGL(k_pack_count, dim3((unsigned)((n_pos + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, n_pos, d_perm, d_indptr, d_row_chunks);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_csr_pack_fill(int64_t n_pos, const int32_t* d_perm, const int64_t* d_indptr,
                     const int32_t* d_indices, const float* d_values, const int64_t* d_cptr,
                     void* d_ent, void* stream) {
  MU_REQUIRE(n_pos >= 0, "negative size");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_indptr && d_cptr && d_ent, "null pointer");
  int64_t blocks = (n_pos + 3) / 4;
  // workgroups per CU: 32 alone; a caller that runs the copy next to a kernel that needs whole CUs
  // (the transpose-pack fill on another stream) lowers it so that both stay resident (tune pack_wg)
  const int per_cu = mu_tune_get("pack_wg") > 0 ? mu_tune_get("pack_wg") : 32;
  const int64_t cap = (int64_t)mu_num_cus() * per_cu;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_pack_fill, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_pos,
                     d_perm, d_indptr, d_indices, d_values, d_cptr, (unsigned long long*)d_ent);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_spmm_packed_f32(int64_t n_rows, int64_t n_cols, const int64_t* d_cptr, const void* d_ent,
                       const int32_t* d_perm, int k_layout, const float* d_Q, int B, float* d_Y,
                       void* stream) {
  MU_REQUIRE(B == 16 || B == 32 || B == 64, "B must be 16, 32 or 64");
  MU_REQUIRE(n_rows >= 0 && n_cols > 0 && n_cols <= ((int64_t)1 << 22), "shape out of range");
  if (n_rows == 0) return MU_OK;
  MU_REQUIRE(d_cptr && d_ent && d_Q && d_Y, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t* cptr = d_cptr;
  const unsigned long long* ent = (const unsigned long long*)d_ent;
  const int32_t* perm = d_perm;
  const float* Q = d_Q;
  float* Y = d_Y;
  // tests / tuning only (mu_tune_set); all 0 in production
  const int force_k = mu_tune_get("spmm_k");
  const int mode = mu_tune_get("spmm_mode");
  const int waves = mu_tune_get("spmm_waves") ? mu_tune_get("spmm_waves") : 16;
  const int pipe = mu_tune_get("spmm_pipe");
  if (waves == 16) {
    int K = (k_layout >= 1 && k_layout <= Geo<16>::KMAX) ? k_layout : pick_k<16>(n_rows);
    if (force_k >= 1 && force_k <= Geo<16>::KMAX) K = force_k;
    if (mode != 0) {
      MU_REQUIRE((K == 7 || K == 8), "ablation modes exist for W = 16, K = 7 / 8 only");
#define MU_ABL(KK)                                             \
  switch (mode) {                                              \
    case 1: MU_GO(k_spmm_pcr64_w16, 16, KK, 1, 0)              \
    case 9: MU_GO(k_spmm_pcr64_w16, 16, KK, 9, 0)              \
    case 11: MU_GO(k_spmm_pcr64_w16, 16, KK, 11, 0)            \
    case 32: MU_GO(k_spmm_pcr64_w16, 16, KK, 32, 1)            \
    case 4: MU_GO(k_spmm_pcr64_w16, 16, KK, 4, 1)              \
    default: break;                                            \
  }
      if (K == 7) MU_ABL(7) else MU_ABL(8)
#undef MU_ABL
      mu_set_error("spmm_mode %d has no compiled instance", mode);
      return MU_ERR_ARG;
    }
#define MU_W16(KK)                                            \
  case KK:                                                    \
    if (pipe == 1) MU_GO(k_spmm_pcr64_w16, 16, KK, 0, 0)      \
    MU_GO(k_spmm_pcr64_w16, 16, KK, 0, 1)
    switch (K) {
      MU_W16(1) MU_W16(2) MU_W16(3) MU_W16(4) MU_W16(5) MU_W16(6) MU_W16(7) MU_W16(8)
      default: break;
    }
#undef MU_W16
  }
  mu_set_error("mu_spmm_packed_f32: no compiled instance for waves=%d k=%d pipe=%d", waves, force_k, pipe);
  return MU_ERR_ARG;
}

}  // extern "C"
