// Row-stream SpMM, wave-specialised (B = 64, f32): Y[n x 64] = X * Q with X a row stream
// (csrc/spmm_win.hip has the format and the single-role kernel this one grew out of).
// ARCHIVED EXPERIMENT (r02; not compiled into libmuon_amd.so since r03): bit-identical to k_spmm_win and 17 % slower
// (5.7 against 4.8 ms at 122 880 x 200 000); DESIGN.md 4.2 and profiles/r02_spmm_ws_accounting.txt.
//
// Why.  In k_spmm_win every wave runs stage A (window cut, cursor update, next request: a chain of
// scalar work and waits) and then stage B (the LDS gathers and FMAs) of a pass, and the 16 waves of
// the CU's one workgroup move through a slab together, so the gather pipes idle while waves are in
// stage A or at the slab barrier: 4.8 ms against 3.25 ms for stage B alone, and stage B alone runs
// as fast on 12 waves as on 16 (spmm_mode 128 / 128 + 65536: the LDS and VALU pipes bound it, not the
// number of waves).  Here 4 "window" waves of the workgroup do stage A for 12 "gather" waves and hand
// the prepared windows - per lane the LDS offset and the value of one entry, plus a header word with
// the slot mask - over through small rings in LDS:
//   * gather wave c owns kKC = 5 row-sets of 4 rows (accumulators in registers) and per slab takes its
//     5 windows, then any overflow windows (rows with more than 16 entries in the slab; tagged with
//     their row-set and announced by the "more" bit of the entry before), from ring c;
//   * window wave p serves gather waves 3p .. 3p+2: 15 row-sets in groups of three whose stage A
//     chains are interleaved, 15 window requests in flight (asm-owned v[96 .. 125]; hipcc cannot be
//     held below v96) and 16 of the next slab's 64 LDS-DMA pieces; VMEM order per slab
//     R9 .. R14 | D0 .. D15 | R0 .. R8, so a window is waited for with the exact counts 30 / 29 / 28;
//   * ring protocol: 5 slots per gather wave, an entry is one ds_write_b64 per lane and one tagged
//     header word (sequence number mod 128, plus 1) - the LDS executes a wave's operations in order,
//     so "data, then header" / "header, then data" need no s_waitcnt; the gather wave's count of taken
//     entries is the only counter, looked at once per slab (before the windows of the NEXT slab's head
//     are published ahead of the slab barrier) and before an overflow window;
//   * the slab barrier stays - one per slab for all 16 waves.
// Results are bit-identical to k_spmm_win (a row's entries are accumulated in column order).
// (Also measured: 8 window + 8 gather waves x 6 row-sets, one gather wave per window wave - the window
//  waves then have time to spare, but eight waves gather 16 % slower than twelve (spmm_mode 128 +
//  131072) and a slab took 8.7k cycles for 192 rows: 8.5 ms.  Twelve gather waves need more than four
//  window waves' worth of instruction issue, and sixteen is all a CU's one workgroup has.)
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.hpp"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
#pragma clang diagnostic ignored "-Winline-asm"

namespace {

constexpr int kSlabCols = 256;
constexpr int kGather = 12;                   // gather waves per workgroup
constexpr int kWindow = 4;                    // window waves
constexpr int kKC = 5;                        // row-sets per gather wave (hipcc cannot be held below v96, so
                                              // a window wave has 15 register pairs for requests in flight)
constexpr int kRS = kGather * kKC / kWindow;  // row-sets per window wave: 15
constexpr int kSetsPerWg = kGather * kKC;     // 60 row-sets = 240 rows
constexpr int kNX = 126 - 2 * kRS;            // asm-owned v[96 .. 125]
constexpr int kRing = 5;                       // slots per ring: the 5 windows of a slab
constexpr int kAhead = 3;                      // windows of the NEXT slab a window wave prepares before the slab barrier
constexpr int kPadCol = 0x7fffffff;
constexpr int kRowBytes = 256, kRowShift = 8, kSlabBytes = kSlabCols * kRowBytes;  // 64 KiB
constexpr int kPiecesPerWave = kSlabBytes / 1024 / kWindow;                           // 16
constexpr int kWaitMain = (kRS - 1) + kPiecesPerWave;                                 // 30
// header of an entry: any16 | flags | row-set << 20 | sequence tag << 24 (the tag - entry number mod 128,
// plus 1 - is what tells a gather wave that the slot holds the entry it waits for: no separate counter)
constexpr unsigned kHdrExtra = 1u << 16, kHdrMore = 1u << 17;

typedef float f4 __attribute__((ext_vector_type(4)));

template <int E>
__device__ __forceinline__ int bcast_i(int x) {
  return __builtin_amdgcn_update_dpp(0, x, 0x150 + E, 0xf, 0xf, true);  // row_newbcast:E
}
template <int E>
__device__ __forceinline__ float bcast_f(float x) {
  return __builtin_bit_cast(float, bcast_i<E>(__builtin_bit_cast(int, x)));
}
struct Quad { f4 q0, q1, q2, q3; };
struct Pair { f4 q0, q1; };
typedef __attribute__((address_space(3))) const f4* lds_f4;
template <int E>
__device__ __forceinline__ Quad quad_read(unsigned base, int a) {
  Quad r;
  r.q0 = *(lds_f4)((unsigned)bcast_i<E>(a) + base);
  r.q1 = *(lds_f4)((unsigned)bcast_i<E + 1>(a) + base);
  r.q2 = *(lds_f4)((unsigned)bcast_i<E + 2>(a) + base);
  r.q3 = *(lds_f4)((unsigned)bcast_i<E + 3>(a) + base);
  return r;
}
template <int E>
__device__ __forceinline__ void quad_fma(const Quad& r, float v, f4& acc) {
  const float v0 = bcast_f<E>(v), v1 = bcast_f<E + 1>(v), v2 = bcast_f<E + 2>(v), v3 = bcast_f<E + 3>(v);
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = fmaf(v0, r.q0[c], acc[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = fmaf(v1, r.q1[c], acc[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = fmaf(v2, r.q2[c], acc[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = fmaf(v3, r.q3[c], acc[c]);
}
template <int E>
__device__ __forceinline__ Pair pair_read(unsigned base, int a) {
  Pair r;
  r.q0 = *(lds_f4)((unsigned)bcast_i<E>(a) + base);
  r.q1 = *(lds_f4)((unsigned)bcast_i<E + 1>(a) + base);
  return r;
}
template <int E>
__device__ __forceinline__ void pair_fma(const Pair& r, float v, f4& acc) {
  const float v0 = bcast_f<E>(v), v1 = bcast_f<E + 1>(v);
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = fmaf(v0, r.q0[c], acc[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = fmaf(v1, r.q1[c], acc[c]);
}

// Stage B of a pass (the same batches as k_spmm_win): slots 0-7 as one batch of eight reads, the
// upper half sized by the highest slot in use.
__device__ __forceinline__ void stage_b(unsigned qbase, int a, float vv, unsigned any16, f4& acc) {
  if (any16 & 0x00f0u) {
    const Quad r0 = quad_read<0>(qbase, a);
    const Quad r1 = quad_read<4>(qbase, a);
    quad_fma<0>(r0, vv, acc);
    quad_fma<4>(r1, vv, acc);
    asm volatile("; eight reads in flight" ::: "memory");
  } else if (any16 & 0x000fu) {
    const Quad r = quad_read<0>(qbase, a);
    quad_fma<0>(r, vv, acc);
  }
  if (any16 & 0xf000u) {
    const Quad r0 = quad_read<8>(qbase, a);
    const Quad r1 = quad_read<12>(qbase, a);
    quad_fma<8>(r0, vv, acc);
    quad_fma<12>(r1, vv, acc);
  } else if (any16 & 0x0c00u) {
    const Quad r = quad_read<8>(qbase, a);
    quad_fma<8>(r, vv, acc);
  } else if (any16 & 0x0300u) {
    const Pair r = pair_read<8>(qbase, a);
    pair_fma<8>(r, vv, acc);
  }
}

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

// asm-owned window registers of a window wave: v[kNX + 2r] (column), v[kNX + 2r + 1] (value bits)
#define MU_WS_CLOB                                                                                   \
  "v96", "v97", "v98", "v99", "v100", "v101", "v102",      \
      "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", \
      "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125"

template <int r>
__device__ __forceinline__ void request_window(unsigned off, const void* base, unsigned long long mask) {
  unsigned long long save;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "v_mov_b32 v%c4, 0x7fffffff\n\t"
      "s_and_b64 exec, exec, %3\n\t"
      "global_load_dwordx2 v[%c4:%c5], %1, %2\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(save)
      : "v"(off), "s"(base), "s"(mask), "i"(kNX + 2 * r), "i"(kNX + 2 * r + 1)
      : MU_WS_CLOB, "scc");
}
template <int r, int N>
__device__ __forceinline__ void wait_window(int& col, int& valbits) {
  asm volatile(
      "s_waitcnt vmcnt(%c2)\n\t"
      "v_mov_b32 %0, v%c3\n\t"
      "v_mov_b32 %1, v%c4"
      : "=v"(col), "=v"(valbits)
      : "i"(N), "i"(kNX + 2 * r), "i"(kNX + 2 * r + 1)
      : MU_WS_CLOB);
}
template <int r>
__device__ __forceinline__ void set_window(int col, int valbits) {
  asm volatile(
      "v_mov_b32 v%c2, %0\n\t"
      "v_mov_b32 v%c3, %1"
      :
      : "v"(col), "v"(valbits), "i"(kNX + 2 * r), "i"(kNX + 2 * r + 1)
      : MU_WS_CLOB);
}

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct Rings {
  uint2 slot[kGather][kRing][64];   // per lane: (LDS byte offset inside the slab, value bits)
  unsigned hdr[kGather][kRing];
  unsigned cons[kGather];           // entries taken
};

// Ring counters.  The LDS executes the instructions of one wave in order, so "data, then counter" on the
// writing side and "counter, then data" on the reading side need no s_waitcnt in between - only the
// compiler has to keep the order (a release / acquire pair would drain lgkmcnt on every hand-over:
// 8.7 ms instead of 5.1 for the product at 125k x 200k).
__device__ __forceinline__ unsigned lds_load_relaxed(const unsigned* p) {
  asm volatile("" ::: "memory");
  const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
__device__ __forceinline__ void lds_store_relaxed(unsigned* p, unsigned v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// DBG: per-wave cycle accounting (s_memtime; coarse on the window waves - reading the counter waits
// for every LDS operation in flight) instead of the product, written to Y[(workgroup * 16 + wave) * 64
// + i]: window waves i = 0 issuing the DMA pieces, 1 the slab's last row-sets + overflow + end markers,
// 2 the head of the next slab, 4 slab end (DMA landing + barrier); gather waves 0 waiting for a
// window, 1 stage B, 4 barrier
template <bool DBG>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(48))) void k_spmm_ws(
    int64_t n_pos, int64_t n_cols, const int64_t* __restrict__ sptr,
    const unsigned long long* __restrict__ ent, const int32_t* __restrict__ perm,
    const float* __restrict__ Q, float* __restrict__ Y) {
  __shared__ float4 qs[2][kSlabBytes / 16];
  __shared__ Rings rings;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane & 15, g = lane >> 4;
  const int64_t rb0 = (int64_t)blockIdx.x * (4 * kSetsPerWg);
  const int64_t rb1 = (rb0 + 4 * kSetsPerWg) < n_pos ? (rb0 + 4 * kSetsPerWg) : n_pos;
  const float4* __restrict__ Q4 = reinterpret_cast<const float4*>(Q);
  const int64_t q4_total = n_cols * (kRowBytes / 16);
  const int ncols32 = (int)n_cols;
  const unsigned qs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&qs[0][0]);
  if (threadIdx.x < kGather) rings.cons[threadIdx.x] = 0u;
  if (threadIdx.x < kGather * kRing) (&rings.hdr[0][0])[threadIdx.x] = 0u;  // no tag is 0

  if (wave >= kGather) {
    // ------------------------------------------------------------------ window wave
    const int pw = wave - kGather;
    // one window wave per SIMD next to three gather waves: its short dependent chains must not queue
    // behind their gathers
    __builtin_amdgcn_s_setprio(3);
    const int64_t wg0 = uniform64(sptr[rb0]);
    const char* __restrict__ entb = reinterpret_cast<const char*>(ent + wg0);
    const unsigned gmask = (g & 1) ? 0xffff0000u : 0x0000ffffu;
    unsigned off[kRS];
    // row-set r of this wave: gather wave 3 pw + r % 3, its row-set r / 3
    auto set_of = [&](int r) { return (3 * pw + (r % 3)) * kKC + (r / 3); };
    static_assert(kRS <= 16, "row ends of a window wave's row-sets live in the lanes of one register");
    // lane 16 g + r (r < 16) / lane 16 g + r - 16 of the second register: end of the row of (r, g)
    unsigned endv0 = 0, endv1 = 0, lo0 = 0, lo1 = 0;
    {
      const int r_a = sub, r_b = 16 + sub;
      const int64_t pa = rb0 + (int64_t)set_of(r_a) * 4 + g;
      const bool oka = pa < rb1;
      lo0 = oka ? (unsigned)((sptr[pa] - wg0) << 3) : 0u;
      endv0 = oka ? (unsigned)((sptr[pa + 1] - wg0) << 3) : 0u;
      if (r_b < kRS) {
        const int64_t pb = rb0 + (int64_t)set_of(r_b) * 4 + g;
        const bool okb = pb < rb1;
        lo1 = okb ? (unsigned)((sptr[pb] - wg0) << 3) : 0u;
        endv1 = okb ? (unsigned)((sptr[pb + 1] - wg0) << 3) : 0u;
      }
    }
    auto end_of = [&](auto rc) -> unsigned {
      constexpr int r = decltype(rc)::value;
      if constexpr (r < 16) return (unsigned)bcast_i<r>((int)endv0);
      else return (unsigned)bcast_i<r - 16>((int)endv1);
    };
    static_for<kRS>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      unsigned lo;
      if constexpr (r < 16) lo = (unsigned)bcast_i<r>((int)lo0);
      else lo = (unsigned)bcast_i<r - 16>((int)lo1);
      off[r] = lo + (unsigned)sub * 8u;
      const bool in = off[r] < end_of(rc);
      const unsigned long long e = in ? *reinterpret_cast<const unsigned long long*>(entb + off[r]) : 0ull;
      set_window<r>(in ? (int)(unsigned)e : kPadCol, (int)(unsigned)(e >> 32));
    });
    auto dma_one = [&](int64_t s0, int buf, int u) {
      const int piece = pw + u * kWindow;
      int64_t i = s0 * (kRowBytes / 16) + piece * 64 + lane;
      if (i >= q4_total) i = q4_total - 1;
      dma_piece(Q4 + i, qs_lds + (unsigned)buf * (unsigned)kSlabBytes + (unsigned)piece * 1024u);
    };
#pragma unroll
    for (int u = 0; u < kPiecesPerWave; ++u) dma_one(0, 0, u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    unsigned tm[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    auto now = [&]() -> unsigned { return (unsigned)__builtin_amdgcn_s_memtime(); };
    unsigned nseq[3] = {0u, 0u, 0u};  // entries published to gather waves 3 pw, 3 pw + 1, 3 pw + 2
    // (a window wave runs a ring ahead of its gather waves, so nearly every publish needs a fresh look at
    //  the consumer counter: it is read right after the previous publish to the same ring, three
    //  row-sets ago, so that the LDS round trip - ~300 cycles under the gathers - is not waited for)
    // Flow control: a ring holds exactly the entries of one slab (5 windows + the end marker), and the
    // gather waves have taken all of them when they pass the slab barrier, so the entries a window wave
    // publishes between two barriers need no check; the kAhead windows of the next slab that it
    // prepares BEFORE the barrier (so that the gather waves find work when they come out of it) and
    // overflow windows do: make_room() looks at the consumer counters (one LDS round trip - ~300
    // cycles under the gathers, which is why it is not done per entry).
    unsigned slot_of[3] = {0u, 0u, 0u};  // next slot of each ring
    auto make_room = [&](int ci, unsigned want) {  // until `want` more entries fit
      const int c = 3 * pw + ci;
      while (true) {
        const unsigned taken = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_load_relaxed(&rings.cons[c]));
        if ((int)(nseq[ci] + want - taken) <= kRing) break;
        __builtin_amdgcn_s_sleep(1);
      }
    };
    // the three consumer counters of this wave's rings with one LDS read (lanes 0-2), to be looked at
    // ~1000 cycles later: an LDS round trip takes that long under the gathers of twelve waves
    auto peek_counters = [&]() -> unsigned {
      return lds_load_relaxed(&rings.cons[3 * pw + (lane < 3 ? lane : 0)]);
    };
    auto room_from_peek = [&](unsigned peek, int ci, unsigned want) -> bool {
      const unsigned taken = (unsigned)__builtin_amdgcn_readlane((int)peek, ci);
      return (int)(nseq[ci] + want - taken) <= kRing;
    };
    auto publish = [&](int ci, int a, int vbits, unsigned hdr) {
      const int c = 3 * pw + ci;
      const unsigned sl = slot_of[ci];
      rings.slot[c][sl][lane] = make_uint2((unsigned)a, (unsigned)vbits);
      asm volatile("" ::: "memory");  // (data, then the tagged header: the LDS keeps a wave's order)
      if (lane == 0) lds_store_relaxed(&rings.hdr[c][sl], hdr | (((nseq[ci] & 0x7fu) + 1u) << 24));
      nseq[ci] += 1u;
      slot_of[ci] = (sl + 1u == (unsigned)kRing) ? 0u : sl + 1u;
    };

    unsigned again = 0, again_next = 0;
    // `rest`: overflow windows of this slab still to come behind this entry, as a row-set mask (the last
    // window of a slab and every overflow window say whether the gather wave has more to take)
    auto stage_a = [&](auto rc, auto slowc, unsigned extra, int s_hi, unsigned& ag, unsigned rest) {
      constexpr int r = decltype(rc)::value;
      constexpr bool SLOW = decltype(slowc)::value;
      constexpr unsigned kMine = (1u << (r % 3)) * 0x1249u;  // row-sets r % 3, + 3, + 6, + 9, + 12
      int col, valbits;
      if constexpr (SLOW) wait_window<r, 0>(col, valbits);
      else wait_window<r, kWaitMain>(col, valbits);
      const bool valid = col < s_hi;
      const unsigned long long m = __ballot(valid);
      const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
      const unsigned mm = mlo | mhi;
      const unsigned any16 = (mm | (mm >> 16)) & 0xffffu;
      const int a = (col & (kSlabCols - 1)) << kRowShift;
      const int vb = valid ? valbits : 0;
      const unsigned mine = (lane & 32) ? mhi : mlo;
      const unsigned cnt = (unsigned)__popc(mine & gmask);
      off[r] += cnt << 3;
      request_window<r>(off[r], entb, __ballot(off[r] < end_of(rc)));
      if (__ballot(cnt == 16u)) ag |= 1u << r;
      unsigned more = 0u;
      if constexpr (SLOW || r / 3 == kKC - 1) more = ((ag | rest) & kMine) ? kHdrMore : 0u;
      publish(r % 3, a, vb, any16 | extra | more | ((unsigned)(r / 3) << 20));
    };
    // Three row-sets (one per gather wave of this window wave) at once.  A row-set's stage A is a chain
    // of ~12 dependent vector <-> scalar steps (compare, ballot, count, cursor, compare, mask); in the
    // single-role kernel four waves per SIMD cover each other's chains, a window wave is alone on its
    // SIMD with this work, so it interleaves three independent chains itself.  The three windows are
    // waited for before any of the three new requests goes out: counts 30, 29, 28.
    auto stage_a3 = [&](auto r0c, int s_hi, unsigned& ag) {
      constexpr int r0 = decltype(r0c)::value;
      static_assert(r0 % 3 == 0, "a group is row-sets 3q, 3q + 1, 3q + 2");
      int col[3], valbits[3];
      wait_window<r0, kWaitMain>(col[0], valbits[0]);
      wait_window<r0 + 1, kWaitMain - 1>(col[1], valbits[1]);
      wait_window<r0 + 2, kWaitMain - 2>(col[2], valbits[2]);
      unsigned any16[3], cnt[3];
      int a[3], vb[3];
      unsigned long long rq[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const bool valid = col[i] < s_hi;
        const unsigned long long m = __ballot(valid);
        const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
        const unsigned mm = mlo | mhi;
        any16[i] = (mm | (mm >> 16)) & 0xffffu;
        a[i] = (col[i] & (kSlabCols - 1)) << kRowShift;
        vb[i] = valid ? valbits[i] : 0;
        const unsigned mine = (lane & 32) ? mhi : mlo;
        cnt[i] = (unsigned)__popc(mine & gmask);
      }
      off[r0] += cnt[0] << 3;
      off[r0 + 1] += cnt[1] << 3;
      off[r0 + 2] += cnt[2] << 3;
      rq[0] = __ballot(off[r0] < end_of(std::integral_constant<int, r0>{}));
      rq[1] = __ballot(off[r0 + 1] < end_of(std::integral_constant<int, r0 + 1>{}));
      rq[2] = __ballot(off[r0 + 2] < end_of(std::integral_constant<int, r0 + 2>{}));
      request_window<r0>(off[r0], entb, rq[0]);
      request_window<r0 + 1>(off[r0 + 1], entb, rq[1]);
      request_window<r0 + 2>(off[r0 + 2], entb, rq[2]);
      if (__ballot(cnt[0] == 16u)) ag |= 1u << r0;
      if (__ballot(cnt[1] == 16u)) ag |= 1u << (r0 + 1);
      if (__ballot(cnt[2] == 16u)) ag |= 1u << (r0 + 2);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        unsigned more = 0u;
        if constexpr (r0 / 3 == kKC - 1) more = (ag & ((1u << i) * 0x1249u)) ? kHdrMore : 0u;
        publish(i, a[i], vb[i], any16[i] | more | ((unsigned)(r0 / 3) << 20));
      }
    };
    auto slab_hi = [&](int64_t s0) -> int {
      return ((int)s0 + kSlabCols) < ncols32 ? ((int)s0 + kSlabCols) : ncols32;
    };
    // The first kHead row-sets of a slab (a ring's worth per gather wave) are prepared BEFORE the
    // barrier that ends the slab before it: the gather waves find windows waiting when they come out
    // of the barrier.  VMEM order per slab: R0 .. R11 | barrier | D0 .. D15 R12 .. R17 - still 33
    // younger operations between a request and its use, whatever the row-set.
    constexpr int kHead = 3 * kAhead;  // 9
    static_assert(kHead <= kRS, "head of a slab");
    static_for<kHead / 3>([&](auto qc) { stage_a3(std::integral_constant<int, 3 * decltype(qc)::value>{}, slab_hi(0), again); });
    int buf = 0;
    for (int64_t s0 = 0; s0 < n_cols; s0 += kSlabCols, buf ^= 1) {
      const int s_hi = slab_hi(s0);
      if (s0 > 0) {
        unsigned tb0 = 0;
        if constexpr (DBG) tb0 = now();
        // this slab's pieces went out one iteration ago, followed by the head's requests
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kHead) : "memory");
        __syncthreads();  // end of the previous slab
        if constexpr (DBG) tm[4] += now() - tb0;
      }
      unsigned tq0 = 0, tq1 = 0;
      if constexpr (DBG) tq1 = now();
      static_for<(kRS - kHead) / 3>([&](auto qc) {
        stage_a3(std::integral_constant<int, kHead + 3 * decltype(qc)::value>{}, s_hi, again);
      });
      if (again) {
        do {
          const unsigned pend = again;
          again = 0;
          static_for<kRS>([&](auto rc) {
            if (pend & (1u << decltype(rc)::value)) {
              constexpr int r = decltype(rc)::value;
              make_room(r % 3, 1u);  // (rare: one LDS round trip)
              stage_a(rc, std::true_type{}, kHdrExtra, s_hi, again, pend & ~((2u << r) - 1u));
            }
          });
        } while (again);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // The next slab's pieces go out AFTER the overflow windows: those end in a full drain
      // (s_waitcnt vmcnt(0): an overflow request is younger than the counted waits assume), and with 15
      // row-sets per window wave most slabs have one - behind the DMA burst that drain waited for the
      // whole slab to land (2.9k cycles per slab).  R12 .. R14 | D0 .. D15 | R0 .. R11: still 30.
      if constexpr (DBG) { tq0 = now(); tm[1] += tq0 - tq1; }
      const unsigned peek = peek_counters();  // (used after the burst)
#pragma unroll
      for (int u = 0; u < kPiecesPerWave; ++u) dma_one(s0 + kSlabCols, buf ^ 1, u);
      unsigned tq2 = 0;
      if constexpr (DBG) { tq2 = now(); tm[0] += tq2 - tq0; }
      if (s0 + kSlabCols < n_cols) {
        again_next = 0;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
          if (!room_from_peek(peek, ci, (unsigned)kAhead)) make_room(ci, (unsigned)kAhead);
        static_for<kHead / 3>([&](auto qc) {
          stage_a3(std::integral_constant<int, 3 * decltype(qc)::value>{}, slab_hi(s0 + kSlabCols), again_next);
        });
        again = again_next;
      }
      if constexpr (DBG) tm[2] += now() - tq2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // end of the last slab
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (DBG) {
      if (lane < 6) Y[((int64_t)blockIdx.x * 16 + wave) * 64 + lane] = (float)tm[lane == 0 ? 0 : lane == 1 ? 1 : lane == 2 ? 2 : lane == 3 ? 3 : lane == 4 ? 4 : 5];
    }
    return;
  }

  // -------------------------------------------------------------------- gather wave
  const int c = wave;
  const int sub_off = sub * 16;
  f4 acc[kKC];
#pragma unroll
  for (int k = 0; k < kKC; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // slab 0 in LDS, ring counters cleared
  unsigned nget = 0;
  unsigned tc[3] = {0u, 0u, 0u};
  auto nowc = [&]() -> unsigned { return (unsigned)__builtin_amdgcn_s_memtime(); };
  // The next window is read while the current one is being worked on: counter, header and data in
  // one round trip, issued before stage B; they are valid if the counter says published (the LDS
  // executes a wave's reads in order: counter first), otherwise the read is repeated.
  struct Pre { unsigned x, y, hdr; };
  unsigned slot_c = 0;  // slot of the entry to take next
  auto issue = [&]() -> Pre {
    Pre p;
    // header first, then the data: if the header carries the expected tag, the data read behind it is
    // the entry's (the window wave wrote data, then header; both sides execute in order)
    p.hdr = lds_load_relaxed(&rings.hdr[c][slot_c]);
    const volatile unsigned* sp = reinterpret_cast<const volatile unsigned*>(&rings.slot[c][slot_c][lane]);
    p.x = sp[0];
    p.y = sp[1];
    asm volatile("" ::: "memory");
    return p;
  };
  auto finish = [&](Pre p, int& a, float& vv) -> unsigned {
    unsigned tt0 = 0;
    if constexpr (DBG) tt0 = nowc();
    while (((unsigned)__builtin_amdgcn_readfirstlane((int)p.hdr) >> 24) != (nget & 0x7fu) + 1u) {
      __builtin_amdgcn_s_sleep(2);
      p = issue();
    }
    a = (int)p.x;
    vv = __builtin_bit_cast(float, p.y);
    ++nget;
    slot_c = (slot_c + 1u == (unsigned)kRing) ? 0u : slot_c + 1u;
    if (lane == 0) lds_store_relaxed(&rings.cons[c], nget);
    if constexpr (DBG) tc[0] += nowc() - tt0;
    return (unsigned)__builtin_amdgcn_readfirstlane((int)p.hdr);
  };
  Pre pre = issue();
  int buf = 0;
  for (int64_t s0 = 0; s0 < n_cols; s0 += kSlabCols, buf ^= 1) {
    const unsigned qbase = qs_lds + (unsigned)buf * (unsigned)kSlabBytes + (unsigned)sub_off;
    unsigned more = 0u;
    static_for<kKC>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      int a;
      float vv;
      const unsigned hdr = finish(pre, a, vv);
      pre = issue();
      if constexpr (k == kKC - 1) more = hdr & kHdrMore;
      unsigned tb0 = 0;
      if constexpr (DBG) tb0 = nowc();
      __builtin_amdgcn_s_setprio(1);
      stage_b(qbase, a, vv, hdr & 0xffffu, acc[k]);
      __builtin_amdgcn_s_setprio(0);
      if constexpr (DBG) {
        if (__ballot(acc[k][0] == 1.2345e-30f)) tc[1] += 1;  // (the FMAs must have issued)
        tc[1] += nowc() - tb0;
      }
    });
    while (more) {
      int a;
      float vv;
      const unsigned hdr = finish(pre, a, vv);
      pre = issue();
      more = hdr & kHdrMore;
      const int k = (int)((hdr >> 20) & 7u);
      static_for<kKC>([&](auto kc) {
        if (k == decltype(kc)::value) stage_b(qbase, a, vv, hdr & 0xffffu, acc[decltype(kc)::value]);
      });
    }
    unsigned tq0 = 0;
    if constexpr (DBG) tq0 = nowc();
    __syncthreads();
    if constexpr (DBG) tc[2] += nowc() - tq0;
  }
  if constexpr (DBG) {
    if (lane < 6) Y[((int64_t)blockIdx.x * 16 + wave) * 64 + lane] = lane == 0 ? (float)tc[0] : lane == 1 ? (float)tc[1] : lane == 4 ? (float)tc[2] : 0.f;
    return;
  }
  static_for<kKC>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int64_t p = rb0 + ((int64_t)c * kKC + k) * 4 + g;
    if (p < rb1) {
      const int64_t out = perm ? (int64_t)perm[p] : p;
      if (out >= 0) *reinterpret_cast<f4*>(Y + out * 64 + sub * 4) = acc[k];
    }
  });
}

}  // namespace

extern "C" {

/* rows per workgroup of the wave-specialised kernel: the layout (muon_amd/_backend.py
 * launch_layout(waves = 12, K = 6)) deals row-sets to 12 gather waves x 6 row-sets */
int mu_spmm_ws_rows_per_wg(void) { return 4 * kSetsPerWg; }
int mu_spmm_ws_gather_waves(void) { return kGather; }

int mu_spmm_ws_f32(int64_t n_pos, int64_t n_cols, const int64_t* d_sptr, const void* d_ent,
                   const int32_t* d_perm, const float* d_Q, int B, float* d_Y, void* stream) {
  MU_REQUIRE(B == 64, "the wave-specialised SpMM exists for B = 64");
  MU_REQUIRE(n_pos >= 0 && n_cols > 0 && n_cols < ((int64_t)1 << 31), "shape out of range");
  if (n_pos == 0) return MU_OK;
  MU_REQUIRE(d_sptr && d_ent && d_Q && d_Y, "null pointer");
  const int64_t wgs = (n_pos + 4 * kSetsPerWg - 1) / (4 * kSetsPerWg);
  if (mu_tune_get("spmm_mode") == 64)
    hipLaunchKernelGGL(k_spmm_ws<true>, dim3((unsigned)wgs), dim3(1024), 0, (hipStream_t)stream, n_pos, n_cols,
                       d_sptr, (const unsigned long long*)d_ent, d_perm, d_Q, d_Y);
  else
    hipLaunchKernelGGL(k_spmm_ws<false>, dim3((unsigned)wgs), dim3(1024), 0, (hipStream_t)stream, n_pos, n_cols,
                       d_sptr, (const unsigned long long*)d_ent, d_perm, d_Q, d_Y);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // extern "C"
