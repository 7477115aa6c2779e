// X^T as a row stream (or CSR) from X - fourth generation of the transposition's fill (r05).
//
// Same job, same output bytes as its predecessor (archived: scripts/probes/tpack_v3.hip; the rmatvec operand of scipy svds, _svds.py:441-466, reached from
// /root/reference/muon/_atac/tools.py:53): a workgroup stages a (row block x column tile) in LDS sorted by (column, cell)
// and writes every column's run with consecutive lanes; stable, no global atomics.  What changed is how a tile is
// sorted, because of what bounded the third generation: its row visits fetched 64 (column, value)
// entries = 3 + 3 lines of 128 bytes from two arrays to use the ~20 that fall into the tile (it needed the NEXT
// tile's entries inside the window for its look-ahead count), one row per wave step, in a chain of LDS atomic round
// trips - 250 GB through the fabric for 50 GB of input at 1e6 x 200k.
//
//   * Source: the ROW STREAM of X ((column, value) pairs, 8 bytes each, one row contiguous - what the TF-IDF scale
//     sweep now writes, csrc/tfidf.hip) or, for callers without one, the CSR arrays.
//   * A wave owns 32 consecutive rows of the block and has ALL of them in flight: 16 load instructions, each the next
//     32 pairs of TWO rows (lanes 0-31 / 32-63, EXEC-masked to the entries the row still has; 256 contiguous bytes =
//     2-3 lines per row and visit), issued one tile ahead into registers the compiler never sees (v94..v127).
//   * No count walk and no look-ahead: a tile is ranked with a BITMAP.  Phase 1: every entry of the tile sets the bit
//     of its row in word (wave, column) - `ds_or_rtn_b32`, 16 independent ones in flight, the returned word holds the
//     wave's earlier rows with that column = the entry's rank inside the wave.  Phase 2: a thread per column turns the
//     16 words of its column into the waves' first staging slots (prefix of popcounts).  Phase 3: slot = word + rank,
//     pair -> staging buffer - from the window registers, nothing is fetched twice.  Phase 4: runs written out.
//   * A row with more than 32 entries in the tile (rare by the host's choice of the tile width: ~4 sigma) takes the
//     wave's one overflow slot (two rows, one more window each); anything beyond that - or a tile that does not fit the
//     staging buffer - retries the tile at half its width (a 16-column tile of 512 rows always fits).
#include <type_traits>
#include <utility>

#include "common.hpp"
#pragma clang diagnostic ignored "-Winline-asm"

namespace {

constexpr int kT = 1024, kNW = 16;   // threads / waves of a workgroup
constexpr int kRW = 32;              // rows of a wave: one bit each in a 32-bit word
constexpr int kMaxC = 512;           // widest tile (columns): 16 waves x 512 columns x (bitmap word, first slot) = 64 KiB
constexpr int kCap = 10240;          // staged pairs: 80 KiB
constexpr int kH0 = 88;              // asm-owned v[88..91]: the next tile's header (base of this block, of the next, run start)
constexpr int kW0 = 92;              // asm-owned registers v[kW0 ..]: slot j = (v[kW0 + 2j] column, v[kW0 + 2j + 1] value bits)
constexpr int kSlotP = 16;           // overflow slot of the rows PREDICTED to overflow (they did in the tile before): prefetched
constexpr int kSlotX = 17;           // overflow slot of the rows that overflow unannounced: loaded on the spot
constexpr int kCountSlab = 8192;     // = sweep.hpp kSlab: the slab pointers are shared

#define MU_T4_CLOB                                                                                                  \
  "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108",  \
      "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", \
      "v123", "v124", "v125", "v126", "v127"

struct T4Out {
  unsigned long long* ent;
  int32_t* idx;
  float* val;
};
__device__ __forceinline__ void t4_store(const T4Out& o, int64_t pos, unsigned long long e) {
  if (o.idx) {
    o.idx[pos] = (int32_t)(unsigned)e;
    o.val[pos] = __builtin_bit_cast(float, (unsigned)(e >> 32));
  } else {
    o.ent[pos] = e;
  }
}

// lane `L` of v <- the wave-uniform x (this clang has no writelane builtin)
template <int L>
__device__ __forceinline__ int writelane_c(int v, int x) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(x), "n"(L));
  return v;
}
__device__ __forceinline__ int writelane_s(int v, int x, int l) {  // (lane select in M0: one SGPR operand per instruction)
  asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(x), "s"(l) : "m0");
  return v;
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}

// ---- count: entries per (row block, column), row blocks of `rpb` consecutive rows ------------------------------------
// (the r04 pipelined count sweep with fixed row blocks: the pieces of a wave's strip of rows as one flat sequence of
//  iterations, unconditional clamped loads from asm, two iterations in flight, bins of M x 8192 columns)
struct T4Walk {
  int l, pb, hi, nrow, step;
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
template <int N>
__device__ __forceinline__ void t4c_wait(int32_t (&c)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : "n"(N) : "memory");
}
template <int M>
__global__ __launch_bounds__(kT) void k_t4_count(int64_t n_rows, int64_t n_cols, int64_t S, int64_t rpb,
                                                 const int64_t* __restrict__ indptr,
                                                 const int32_t* __restrict__ indices,
                                                 const int64_t* __restrict__ sp, uint32_t* __restrict__ cnt) {
  __shared__ uint32_t bins[kCountSlab * M];
  const int g = blockIdx.x;
  const int64_t r0 = (int64_t)g * rpb;
  const int64_t r1 = (r0 + rpb) < n_rows ? (r0 + rpb) : n_rows;
  const int wave = uniform32(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t wg_base = uniform64(indptr[r0 < n_rows ? r0 : n_rows]);
  const int32_t* ib = indices + wg_base;  // (32-bit byte offsets from here: the host keeps a block under 2^29 entries)
  for (int64_t s = 0; s < S; s += M) {
    for (int t = threadIdx.x; t < kCountSlab * M; t += kT) bins[t] = 0u;
    __syncthreads();
    const int32_t cbase = (int32_t)(s * kCountSlab);
    const int64_t s_hi = s + M < S ? s + M : S;
    for (int64_t strip = r0 + wave; strip < r1; strip += (int64_t)kNW * 64) {
      const int64_t myrow = strip + (int64_t)kNW * lane;
      int lo_l = 0, hi_l = 0;
      if (myrow < r1) {
        lo_l = (int)(sp[myrow * (S + 1) + s] - wg_base);
        hi_l = (int)(sp[myrow * (S + 1) + s_hi] - wg_base);
      }
      asm volatile("" ::"v"(lo_l), "v"(hi_l));  // (the compiler's wait for these loads: here, not inside the walk)
      const int64_t left = (r1 - strip + kNW - 1) / kNW;
      T4Walk w{-1, 0, 0, left < 64 ? (int)left : 64, 256};
      int32_t ca[4], cb[4];
      auto load = [&](int32_t (&c)[4], int pb, int hi) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int q = pb + lane + 64 * u;
          q = q < hi ? q : hi - 1;
          asm volatile("global_load_dword %0, %1, %2" : "=&v"(c[u]) : "v"((unsigned)q * 4u), "s"(ib) : "memory");
        }
      };
      auto work = [&](int32_t (&c)[4], int pb, int hi) {
        t4c_wait<4>(c);  // (the next iteration's four loads stay in flight)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = pb + lane + 64 * u < hi;  // a lane past the end adds 0 to a bin of its own
          atomicAdd(&bins[ok ? c[u] - cbase : lane], ok ? 1u : 0u);
        }
      };
      if (w.advance(lo_l, hi_l)) {
        int a_pb = w.pb, a_hi = w.hi, b_pb, b_hi;
        load(ca, a_pb, a_hi);
        for (;;) {
          bool more = w.advance(lo_l, hi_l);
          b_pb = more ? w.pb : a_pb, b_hi = more ? w.hi : a_hi;
          load(cb, b_pb, b_hi);
          work(ca, a_pb, a_hi);
          if (!more) break;
          more = w.advance(lo_l, hi_l);
          a_pb = more ? w.pb : b_pb, a_hi = more ? w.hi : b_hi;
          load(ca, a_pb, a_hi);
          work(cb, b_pb, b_hi);
          if (!more) break;
        }
        t4c_wait<0>(ca);
        t4c_wait<0>(cb);
      }
    }
    __syncthreads();
    const int64_t here = (n_cols - (int64_t)cbase) < kCountSlab * M ? (n_cols - (int64_t)cbase) : kCountSlab * M;
    uint32_t* dst = cnt + (int64_t)g * n_cols + cbase;
    for (int t = threadIdx.x; t < here; t += kT) dst[t] = bins[t];
    __syncthreads();
  }
}

// cnt[g][c] <- sum_{g' < g} cnt[g'][c];  coltot[c] = col_nnz[c] = sum_g cnt[g][c]
__global__ __launch_bounds__(256) void k_t4_base(int64_t n_cols, int G, uint32_t* __restrict__ cnt,
                                                 int64_t* __restrict__ coltot, int64_t* __restrict__ col_nnz) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  uint32_t run = 0;
  for (int g = 0; g < G; ++g) {
    const uint32_t t = cnt[(int64_t)g * n_cols + c];
    cnt[(int64_t)g * n_cols + c] = run;
    run += t;
  }
  cnt[(int64_t)G * n_cols + c] = run;  // row G: the totals (the last block's "next block")
  coltot[c] = (int64_t)run;
  col_nnz[c] = (int64_t)run;
}

// cdst[c] = first pair of column c's output row in the target (the header of a tile then has no dependent load)
__global__ __launch_bounds__(256) void k_t4_cdst(int64_t n_cols, const int64_t* __restrict__ cptr,
                                                 const int32_t* __restrict__ inv, int64_t* __restrict__ cdst) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_cols) cdst[c] = cptr[inv ? (int64_t)inv[c] : c];
}

// ---- the window registers ----------------------------------------------------------------------------------------
// Slot j holds the next 32 pairs of two rows: lanes 0-31 row a, lanes 32-63 row b; a lane past its row's end keeps
// the padding column.  Always exactly the same instructions whatever the masks (a VMEM instruction with EXEC = 0 is
// issued and counted in order on gfx950: scripts/probes/exec0_vmcnt.hip).
template <int J>
__device__ __forceinline__ void t4_issue_pairs(uint64_t a0, uint64_t a1, int n0, int n1, unsigned sub8) {
  unsigned long long save, m0, m1;
  // (n0, n1 in 0 .. 32: the caller clamps the counts of all rows with ONE vector instruction per tile.  Written here in
  //  C++ the clamps came back as v_med3 into VGPRs, which s_bfm cannot read; as four scalar instructions per slot they
  //  were a fifth of the issue sequence)
  asm volatile(
      "s_bfm_b64 %[m0], %[n0], 0\n\t"
      "s_bfm_b64 %[m1], %[n1], 32\n\t"
      "s_mov_b64 %[save], exec\n\t"
      "v_mov_b32 v%c[C], 0x7fffffff\n\t"
      "s_mov_b64 exec, %[m0]\n\t"
      "global_load_dwordx2 v[%c[C]:%c[V]], %[off], %[a0]\n\t"
      "s_mov_b64 exec, %[m1]\n\t"
      "global_load_dwordx2 v[%c[C]:%c[V]], %[off], %[a1]\n\t"
      "s_mov_b64 exec, %[save]"
      : [save] "=&s"(save), [m0] "=&s"(m0), [m1] "=&s"(m1)
      : [off] "v"(sub8), [a0] "s"(a0), [a1] "s"(a1), [n0] "s"(n0), [n1] "s"(n1), [C] "i"(kW0 + 2 * J),
        [V] "i"(kW0 + 2 * J + 1)
      : MU_T4_CLOB, "memory");
}
// the same from the CSR arrays: i0 / i1 = address of the row's next column index, v0 / v1 = of its next value
template <int J>
__device__ __forceinline__ void t4_issue_csr(uint64_t i0, uint64_t v0, uint64_t i1, uint64_t v1, int n0, int n1,
                                             unsigned sub4) {
  unsigned long long save, m0, m1;
  asm volatile(
      "s_bfm_b64 %[m0], %[n0], 0\n\t"
      "s_bfm_b64 %[m1], %[n1], 32\n\t"
      "s_mov_b64 %[save], exec\n\t"
      "v_mov_b32 v%c[C], 0x7fffffff\n\t"
      "s_mov_b64 exec, %[m0]\n\t"
      "global_load_dword v%c[C], %[off], %[i0]\n\t"
      "global_load_dword v%c[V], %[off], %[v0]\n\t"
      "s_mov_b64 exec, %[m1]\n\t"
      "global_load_dword v%c[C], %[off], %[i1]\n\t"
      "global_load_dword v%c[V], %[off], %[v1]\n\t"
      "s_mov_b64 exec, %[save]"
      : [save] "=&s"(save), [m0] "=&s"(m0), [m1] "=&s"(m1)
      : [off] "v"(sub4), [i0] "s"(i0), [v0] "s"(v0), [i1] "s"(i1), [v1] "s"(v1), [n0] "s"(n0), [n1] "s"(n1),
        [C] "i"(kW0 + 2 * J), [V] "i"(kW0 + 2 * J + 1)
      : MU_T4_CLOB, "memory");
}
// CIRCULAR windows (r05, PAIRS only; tune tpack4_circ): pair number p of the stream lives in lane p % 32 of its row's
// half, so a window keeps the pairs a tile did not consume where they are and only the lanes whose pairs were consumed
// are loaded again - the next pairs of the row, which land in exactly those lanes.  b = address of the 256-byte block the
// first new pair lies in, s = its lane, m = the 32 lanes to load (a run of the new pairs' number starting at s, wrapped);
// a lane before s belongs to the NEXT 256-byte block.  The plain windows request 32 pairs = three lines per row and
// tile to consume ~15 (162 GB of requests for 50 GB of pairs, profiles/r05_tpack4_ablations.txt).
template <int J>
__device__ __forceinline__ void t4_issue_circ(uint64_t b0, uint64_t b1, unsigned s0, unsigned s1, unsigned m0, unsigned m1,
                                              unsigned sub, unsigned sub8, unsigned sub8w) {
  unsigned long long save;
  unsigned t0, t1;
  asm volatile(
      "s_mov_b64 %[save], exec\n\t"
      "v_cmp_gt_u32 vcc, %[s0], %[sub]\n\t"
      "v_cndmask_b32 %[t0], %[sub8], %[sub8w], vcc\n\t"
      "v_cmp_gt_u32 vcc, %[s1], %[sub]\n\t"
      "v_cndmask_b32 %[t1], %[sub8], %[sub8w], vcc\n\t"
      "s_mov_b32 exec_lo, %[m0]\n\t"
      "s_mov_b32 exec_hi, 0\n\t"
      "global_load_dwordx2 v[%c[C]:%c[V]], %[t0], %[b0]\n\t"
      "s_mov_b32 exec_lo, 0\n\t"
      "s_mov_b32 exec_hi, %[m1]\n\t"
      "global_load_dwordx2 v[%c[C]:%c[V]], %[t1], %[b1]\n\t"
      "s_mov_b64 exec, %[save]"
      : [save] "=&s"(save), [t0] "=&v"(t0), [t1] "=&v"(t1)
      : [sub] "v"(sub), [sub8] "v"(sub8), [sub8w] "v"(sub8w), [b0] "s"(b0), [b1] "s"(b1), [s0] "s"(s0), [s1] "s"(s1),
        [m0] "s"(m0), [m1] "s"(m1), [C] "i"(kW0 + 2 * J), [V] "i"(kW0 + 2 * J + 1)
      : MU_T4_CLOB, "vcc", "memory");
}
// every lane of slot J holds the padding column (before the first circular request)
template <int J>
__device__ __forceinline__ void t4_pad_slot() {
  asm volatile("v_mov_b32 v%c0, 0x7fffffff" ::"i"(kW0 + 2 * J) : MU_T4_CLOB);
}
// the pairs of slot J that the tile took (column < cend) become padding: their lanes are loaded again only if the row
// has pairs left for them
template <int J>
__device__ __forceinline__ void t4_retire_slot(int cend, int pad) {  // (pad = 0x7fffffff in a register: vcc takes the constant bus)
  asm volatile(
      "v_cmp_le_i32 vcc, %0, v%c1\n\t"
      "v_cndmask_b32 v%c1, %2, v%c1, vcc" ::"s"(cend),
      "i"(kW0 + 2 * J), "v"(pad)
      : MU_T4_CLOB, "vcc");
}
__device__ __forceinline__ void t4_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: MU_T4_CLOB, "memory"); }
template <int J>
__device__ __forceinline__ int t4_col() {
  int c;
  asm volatile("v_mov_b32 %0, v%c1" : "=v"(c) : "i"(kW0 + 2 * J) : MU_T4_CLOB);
  return c;
}
template <int J>
__device__ __forceinline__ unsigned t4_val() {
  unsigned v;
  asm volatile("v_mov_b32 %0, v%c1" : "=v"(v) : "i"(kW0 + 2 * J + 1) : MU_T4_CLOB);
  return v;
}

// A tile's header - this block's count prefix of the tile's columns (b0), the next block's (b1), where the columns' runs
// start in the target (cd) - is requested ONE TILE AHEAD into v[kH0 .. kH0 + 3] (three loads by every thread, clamped
// inside the arrays) and taken at the tile's top behind the wait that the windows need anyway.
__device__ __forceinline__ void t4_issue_header(const uint32_t* bg, const uint32_t* bn, const int64_t* cd, unsigned off4) {
  asm volatile(
      "global_load_dword v%c4, %0, %1\n\t"
      "global_load_dword v%c5, %0, %2\n\t"
      "v_lshlrev_b32 v%c6, 1, %0\n\t"
      "global_load_dwordx2 v[%c6:%c7], v%c6, %3"
      :
      : "v"(off4), "s"(bg), "s"(bn), "s"(cd), "i"(kH0), "i"(kH0 + 1), "i"(kH0 + 2), "i"(kH0 + 3)
      : MU_T4_CLOB, "memory");
}
__device__ __forceinline__ void t4_take_header(uint32_t& b0, uint32_t& b1, int64_t& cd) {
  unsigned lo, hi;
  asm volatile("v_mov_b32 %0, v%c4\n\tv_mov_b32 %1, v%c5\n\tv_mov_b32 %2, v%c6\n\tv_mov_b32 %3, v%c7"
               : "=v"(b0), "=v"(b1), "=v"(lo), "=v"(hi)
               : "i"(kH0), "i"(kH0 + 1), "i"(kH0 + 2), "i"(kH0 + 3)
               : MU_T4_CLOB);
  cd = (int64_t)(((unsigned long long)hi << 32) | lo);
}

template <int... I, class F>
__device__ __forceinline__ void t4_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void t4_for(F&& f) {
  t4_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

__device__ unsigned long long g_t4_phase[8];  // tune tpack_dbg: cycles of header / phase 1 / 2 / 3 / write-out / retries

// PAIRS: the source is the row stream of X (src0 = ent, row_dst[row] = pair index of the row's first pair);
// otherwise the CSR arrays (src0 = indices, src1 = values).  rw = rows of a wave (<= 32), rpb = 16 rw rows per block.
template <bool PAIRS, bool DBG = false, bool OUT_CSR = false, bool CIRC = false>
__global__ __launch_bounds__(kT) __attribute__((amdgpu_num_vgpr(44))) void k_t4_fill(
    int64_t n_rows, int64_t n_cols, int C, int rw, int G, const int64_t* __restrict__ indptr,
    const int64_t* __restrict__ row_dst, const void* __restrict__ src0, const void* __restrict__ src1,
    const int64_t* __restrict__ cdst, const uint32_t* __restrict__ base, const int64_t* __restrict__ coltot,
    T4Out out, int* __restrict__ err, int abl) {
  const bool late = (abl & 8) != 0;
  __shared__ unsigned long long stage[kCap];  // 80 KiB
  __shared__ uint2 bm[kNW][kMaxC];            // 64 KiB: (wave, column): .x bitmap of the wave's rows, .y its first slot
  // per column of a tile: pairs << 16 | first staging slot, and where the run goes - TWO sets: a tile is written out
  // while the next one is being ranked (see the loop)
  __shared__ uint32_t lrun[2][kMaxC];
  __shared__ int64_t gdst[2][kMaxC];
  __shared__ uint32_t wsum[kNW];
  __shared__ int s_flag;
  // Workgroup -> row block, XCD-aware: workgroup w runs on XCD w % 8 (observed dispatch order; for speed only), and the
  // 32 CUs of an XCD get CONSECUTIVE row blocks.  The runs that neighbouring row blocks write for one column are
  // neighbours in the output row (~120 bytes each): written through the same L2 at about the same time - the blocks of
  // an XCD walk the column tiles in step - they leave it as whole lines instead of one partial line per block and XCD.
  // (G < 0: the plain order, for comparison.)
  int g = blockIdx.x;
  if (G > 0) {
    const int per = (G + 7) / 8;
    g = (g % 8) * per + (g / 8);
    if (g >= G) return;
  } else {
    G = -G;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform32(tid >> 6);
  const int half = lane >> 5, sub = lane & 31;
  const int64_t rpb = (int64_t)kNW * rw;
  const int64_t r0 = (int64_t)g * rpb;
  const int64_t r1 = (r0 + rpb) < n_rows ? (r0 + rpb) : n_rows;
  const int64_t wr0 = r0 + (int64_t)wave * rw;  // this wave's first row
  uint2* bmw = &bm[wave][0];
  for (int t = tid; t < kNW * kMaxC; t += kT) (&bm[0][0])[t] = make_uint2(0u, 0u);
  if (tid == 0) s_flag = 0;

  // lane l < 32: the state of row wr0 + l - address of its next entry and the entries it has left
  uint64_t A = 0, A2 = 0;
  int rem = 0;
  {
    const int64_t row = wr0 + sub;
    const bool ok = sub < rw && row < r1;
    const int64_t p0 = ok ? indptr[row] : 0;
    rem = ok ? (int)(indptr[row + 1] - p0) : 0;
    if (PAIRS) {
      A = (uint64_t)src0 + 8ull * (uint64_t)(ok ? row_dst[row] : 0);
    } else {
      A = (uint64_t)src0 + 4ull * (uint64_t)p0;
      A2 = (uint64_t)src1 + 4ull * (uint64_t)p0;
    }
  }
  static_assert(!CIRC || PAIRS, "circular windows read the row stream");
  const unsigned subo = (unsigned)sub * (PAIRS ? 8u : 4u);
  int pa = -1, pb = -1;  // rows (of this wave) whose continuation window the next issue prefetches into slot kSlotP
  int ld = 0;            // CIRC, lane l < 32: pairs of row l behind its cursor that sit in its window already
  if constexpr (CIRC) t4_for<16>([&](auto jc) { t4_pad_slot<decltype(jc)::value>(); });
  auto issue_all = [&]() {
    const int remc = rem < 32 ? rem : 32;  // (rem >= 0 always)
    if constexpr (CIRC) {
      const int nn = remc - ld;                                  // new pairs of the row (>= 0) ...
      const uint64_t st = A + 8ull * (uint64_t)(unsigned)ld;     // ... from this address on
      const unsigned s5 = (unsigned)(st >> 3) & 31u;
      const unsigned rot = __builtin_rotateleft32((unsigned)((1ull << nn) - 1ull), s5);
      const uint64_t b256 = st & ~255ull;
      ld = remc;
      t4_for<16>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        t4_issue_circ<J>(readlane_u64(b256, 2 * J), readlane_u64(b256, 2 * J + 1),
                         (unsigned)__builtin_amdgcn_readlane((int)s5, 2 * J), (unsigned)__builtin_amdgcn_readlane((int)s5, 2 * J + 1),
                         (unsigned)__builtin_amdgcn_readlane((int)rot, 2 * J), (unsigned)__builtin_amdgcn_readlane((int)rot, 2 * J + 1),
                         (unsigned)sub, subo, subo + 256u);
      });
    } else {
    t4_for<16>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const int n0 = __builtin_amdgcn_readlane(remc, 2 * J), n1 = __builtin_amdgcn_readlane(remc, 2 * J + 1);
      if constexpr (PAIRS)
        t4_issue_pairs<J>(readlane_u64(A, 2 * J), readlane_u64(A, 2 * J + 1), n0, n1, subo);
      else
        t4_issue_csr<J>(readlane_u64(A, 2 * J), readlane_u64(A2, 2 * J), readlane_u64(A, 2 * J + 1),
                        readlane_u64(A2, 2 * J + 1), n0, n1, subo);
    });
    }
    if (pa >= 0) {
      // the continuation windows (entries 32 .. 63 behind the cursor) of the rows that overflowed in the tile before: a
      // row inside a dense stretch of columns overflows tile after tile, and a continuation requested only when phase 1
      // finds it missing is a memory round trip all sixteen waves wait for (92 % of the tiles of the bench matrix)
      const unsigned step = PAIRS ? 256u : 128u;
      const int lb = pb >= 0 ? pb : pa;
      int na = __builtin_amdgcn_readlane(rem, pa) - 32, nb = pb >= 0 ? __builtin_amdgcn_readlane(rem, pb) - 32 : 0;
      na = __builtin_amdgcn_readfirstlane(na < 0 ? 0 : (na > 32 ? 32 : na));
      nb = __builtin_amdgcn_readfirstlane(nb < 0 ? 0 : (nb > 32 ? 32 : nb));
      if constexpr (PAIRS)
        t4_issue_pairs<kSlotP>(readlane_u64(A, pa) + step, readlane_u64(A, lb) + step, na, nb, subo);
      else
        t4_issue_csr<kSlotP>(readlane_u64(A, pa) + step, readlane_u64(A2, pa) + step, readlane_u64(A, lb) + step,
                             readlane_u64(A2, lb) + step, na, nb, subo);
    }
  };
  issue_all();
  __syncthreads();

  // (the count array has G + 1 rows - row G holds the column totals - so a tile's header is three independent loads)
  const uint32_t* base_g = base + (int64_t)g * n_cols;
  const uint32_t* base_n = base + (int64_t)(g + 1) * n_cols;
  constexpr bool dbg = DBG;  // (phase accounting: its own instance, the counters cost registers)
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
  auto mark = [&](int i) {
    if constexpr (DBG) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      ph[i] += t - t_prev;
      t_prev = t;
    }
  };

  // ---- phase 4: write-out, one 16-lane group per column, consecutive lanes = consecutive pairs of the run -----------
  // (four columns of a group at a time: their four table entries first, then the four staged pairs, then the four
  //  stores - three LDS round trips per FOUR columns.)  Two sets of run tables: `late` (tune tpack4_late = 1) runs it ONE
  //  TILE LATE, between the first and the second barrier of the next tile, so that the stores have a whole tile to drain
  //  before this wave's next `s_waitcnt vmcnt(0)`.  Measured (profiles/r05_tpack4_ablations.txt): 58.9 against 55.8 ms -
  //  the drain leaves the top of the tile, and the window requests of phase 3 queue behind the stores instead (phase 3
  //  160 instead of 74 hundred cycles): the stores cost their ~20 ms wherever they sit.  Not the default.
  auto put = [&](int64_t pos, unsigned long long e) {
    if constexpr (OUT_CSR) {
      out.idx[pos] = (int32_t)(unsigned)e;
      out.val[pos] = __builtin_bit_cast(float, (unsigned)(e >> 32));
    } else {
      out.ent[pos] = e;
    }
  };
  auto write_out = [&](int set, int ncol) {
    const int grp = tid >> 4, s16 = tid & 15;
    for (int c0 = grp; c0 < ncol; c0 += 4 * (kT / 16)) {  // (uniform trip count per wave up to the last round)
      uint32_t lr[4];
      int64_t gd[4];
      unsigned long long e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cl = c0 + u * (kT / 16);
        const bool in = cl < ncol;
        lr[u] = in ? lrun[set][in ? cl : 0] : 0u;
        gd[u] = gdst[set][in ? cl : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t L = lr[u] >> 16, src = lr[u] & 0xffffu;
        e[u] = stage[(uint32_t)s16 < L ? src + s16 : 0];
      }
      if (abl & 2) continue;  // (timing ablations, tune tpack4_abl: 2 no stores, 4 every run to the start of the target)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t L = lr[u] >> 16, src = lr[u] & 0xffffu;
        const int64_t dd = (abl & 4) ? (int64_t)((c0 + u) & 63) * 16 : gd[u];
        if ((uint32_t)s16 < L) put(dd + s16, e[u]);
        for (uint32_t i = 16 + s16; i < L; i += 16) put(dd + i, stage[src + i]);  // (a column with more than 16 pairs here)
      }
    }
  };
  int pend_cols = 0, pend_set = 0, tset = 0;  // the tile staged and not yet written out (its columns, its table set)

  int Ct = C;
  int64_t hdr_cb = -1;  // the tile (its first column) whose header sits in / is on its way to v[kH0 ..]
  for (int64_t cb = 0; cb < n_cols;) {
    const int32_t cbase = (int32_t)cb;
    const int32_t cend = (int32_t)((cb + Ct) < n_cols ? (cb + Ct) : n_cols);
    // (the per-slot constants made from `half` - row bit, row id, rank mask - are two instructions each: recomputed per
    //  tile, not kept in 50 registers across the loop, which is what loop-invariant code motion did and spilled for)
    int hf = half;
    asm volatile("" : "+v"(hf));
    // ---- header: this block's pairs per column of the tile, their exclusive scan, where the runs go ----------------
    // (requested one tile ahead - see t4_issue_header; a retried tile or the first one asks now.  The wait covers the
    //  windows of phase 1 too: everything this wave has in flight was issued at least a phase ago)
    if (hdr_cb != cb) {
      int64_t ofs = n_cols - 1 - cb;
      ofs = ofs < 0 ? 0 : ofs;
      const unsigned o4 = (unsigned)((int64_t)tid < ofs ? (int64_t)tid : ofs) * 4u;
      t4_issue_header(base_g + cb, base_n + cb, cdst + cb, o4);
    }
    t4_wait_all();
    uint32_t mine = 0;
    int64_t gd = 0;
    {
      uint32_t b0, b1;
      int64_t cd;
      t4_take_header(b0, b1, cd);
      if (tid < Ct && cbase + tid < cend) {
        mine = b1 - b0;
        gd = cd + (int64_t)b0;
      }
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    mark(0);

    // ---- phase 1: the tile's entries set their row's bit in word (wave, column) ---------------------------------------
    int cntv = 0;          // lane r < 32: entries of row r consumed by this tile
    t4_for<16>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const int c = t4_col<J>();
      const bool valid = c < cend;  // sorted rows: a prefix of each half; padding lanes hold INT_MAX
      const unsigned long long m = __ballot(valid);
      // (no result asked for: sixteen of these go out back to back; the two rows of a slot may share a column - the
      //  same word in one instruction - which an OR does not mind)
      if (valid)
        __hip_atomic_fetch_or(&bmw[c - cbase].x, (1u << (2 * J)) << hf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      cntv = writelane_c<2 * J>(cntv, __popc((unsigned)m));
      cntv = writelane_c<2 * J + 1>(cntv, __popc((unsigned)(m >> 32)));
    });
    // rows whose window was used up and that have entries left: more of them may fall into this tile (not when the tile
    // has 32 columns or fewer: a row has at most one entry per column)
    unsigned ov = (unsigned)__ballot(half == 0 && cntv == 32 && rem > 32);
    if (Ct <= 32) ov = 0u;
    const unsigned ov_all = ov;
    int xa = -1, xb = -1;
    bool again = false;  // (wave-uniform) this wave cannot finish the tile at this width
    // one continuation window of two rows: its entries of the tile set their bits, the rows' counts grow
    auto continuation = [&](auto jc, int ra, int rb) {
      constexpr int J = decltype(jc)::value;
      const int c = t4_col<J>();
      const bool valid = c < cend;
      const unsigned long long m = __ballot(valid);
      const int c0 = __popc((unsigned)m), c1 = __popc((unsigned)(m >> 32));
      const int lb = rb >= 0 ? rb : ra;
      if (valid)
        __hip_atomic_fetch_or(&bmw[c - cbase].x, 1u << (hf ? lb : ra), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      cntv = writelane_s(cntv, __builtin_amdgcn_readlane(cntv, ra) + c0, ra);
      if (rb >= 0) cntv = writelane_s(cntv, __builtin_amdgcn_readlane(cntv, rb) + c1, rb);
      // 64 entries of one row in the tile and more to come: half the width
      if ((c0 == 32 && __builtin_amdgcn_readlane(rem, ra) > 64) || (rb >= 0 && c1 == 32 && __builtin_amdgcn_readlane(rem, rb) > 64))
        again = true;
    };
    if (pa >= 0 && Ct > 32) {
      // the prefetched continuations (they arrived with the regular windows).  A predicted row that did NOT use its
      // window up has nothing of this tile in its continuation: sorted rows, every lane fails `c < cend`
      continuation(std::integral_constant<int, kSlotP>{}, pa, pb);
      ov &= ~((1u << pa) | (pb >= 0 ? 1u << pb : 0u));
    }
    if (ov) {  // overflowing rows nobody announced: up to two, loaded on the spot
      xa = __builtin_ctz(ov);
      ov &= ov - 1u;
      if (ov) {
        xb = __builtin_ctz(ov);
        ov &= ov - 1u;
      }
      if (ov) {  // more than that in one wave: retry the tile at half its width
        again = true;
        xa = xb = -1;
      } else {
        const unsigned step = PAIRS ? 256u : 128u;
        const int na = __builtin_amdgcn_readlane(rem, xa) - 32;  // (> 0: the row has entries left)
        const int nb = xb >= 0 ? __builtin_amdgcn_readlane(rem, xb) - 32 : 0;
        const int lb = xb >= 0 ? xb : xa;
        const int nac = __builtin_amdgcn_readfirstlane(na < 32 ? na : 32), nbc = __builtin_amdgcn_readfirstlane(nb < 32 ? nb : 32);
        if constexpr (PAIRS)
          t4_issue_pairs<kSlotX>(readlane_u64(A, xa) + step, readlane_u64(A, lb) + step, nac, nbc, subo);
        else
          t4_issue_csr<kSlotX>(readlane_u64(A, xa) + step, readlane_u64(A2, xa) + step, readlane_u64(A, lb) + step,
                               readlane_u64(A2, lb) + step, nac, nbc, subo);
        t4_wait_all();  // (exposed: the first tile of a dense stretch)
        continuation(std::integral_constant<int, kSlotX>{}, xa, xb);
      }
    }
    if (again && lane == 0) s_flag = 1;
    mark(1);
    __syncthreads();  // B1: wsum, every wave's bitmap rows, s_flag
    uint32_t wpre = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kNW; ++w) {
      const uint32_t t = wsum[w];
      if (w < wave) wpre += t;
      total += t;
    }
    const uint32_t my_lpos = wpre + incl - mine;
    const bool retry = (s_flag != 0) || total > (uint32_t)kCap;
    if (retry || total == 0) {
      // (nothing of this block in the tile: next tile.  Retry: the bitmap rows are cleared and the same windows are
      //  looked at again against a tile of half the width; a 16-column tile always fits)
      if (retry)
        for (int t = lane; t < Ct; t += 64) bmw[t].x = 0u;
      __syncthreads();
      if (tid == 0) s_flag = 0;
      if (retry) {
        if (Ct <= 16) {  // cannot happen (16 columns x 512 rows fit, a row has <= 16 entries): do not spin
          if (tid == 0) atomicOr(err, 2);
          cb += Ct;
          Ct = C;
        } else {
          Ct = ((Ct / 2) / 16) * 16;
          if (Ct < 16) Ct = 16;
        }
        if (DBG && tid == 0) ph[5] += 1;
      } else {
        cb += Ct;
        Ct = C;
      }
      __syncthreads();
      continue;
    }
    if (tid < Ct) {
      lrun[tset][tid] = (mine << 16) | my_lpos;  // (both < 2^14: the tile fits the staging buffer)
      gdst[tset][tid] = gd;
    }
    {  // the next tile's header, one tile ahead (the registers were taken at the top)
      const int64_t ncb = cb + Ct;
      if (ncb < n_cols) {
        int64_t ofs = n_cols - 1 - ncb;
        const unsigned o4 = (unsigned)((int64_t)tid < ofs ? (int64_t)tid : ofs) * 4u;
        t4_issue_header(base_g + ncb, base_n + ncb, cdst + ncb, o4);
      }
      hdr_cb = ncb;
    }
    // ---- phase 2: per column, the waves' first staging slots (exclusive prefix of the popcounts over the waves) ------
    if (tid < Ct) {
      uint32_t run = my_lpos;
#pragma unroll
      for (int w = 0; w < kNW; ++w) {
        const uint32_t x = bm[w][tid].x;
        bm[w][tid].y = run;
        run += (uint32_t)__popc(x);
      }
      if (run - my_lpos != mine) atomicOr(err, 1);  // the bitmap and the count pass disagree: never
    }
    mark(2);
    // the tile before (staged behind ITS third barrier, untouched since: phase 3 below is the next writer of the
    // staging buffer, behind this tile's second barrier)
    if (pend_cols > 0) write_out(pend_set, pend_cols);
    pend_cols = 0;
    __syncthreads();  // B2
    // ---- phase 3: every entry to its slot, from the window registers -------------------------------------------------
    // (eight slots at a time: their eight (bitmap, first slot) pairs first - one LDS round trip - then the eight stores.
    //  The bits below an entry's own row in its wave's word are its rank among the wave's rows with that column)
    struct Ent { int c; unsigned v; uint2 wd; };
    auto fetch = [&](auto jc) -> Ent {
      constexpr int J = decltype(jc)::value;
      Ent e;
      e.c = t4_col<J>();
      e.v = t4_val<J>();
      e.wd = bmw[e.c < cend ? e.c - cbase : 0];
      return e;
    };
    auto place = [&](const Ent& e, int ra, int rb) {
      if (e.c < cend) {
        const int rloc = hf ? rb : ra;
        const uint32_t below = (1u << rloc) - 1u;  // (rloc <= 31)
        const uint32_t slot = e.wd.y + (uint32_t)__popc(e.wd.x & below);
        stage[slot] = (unsigned long long)(unsigned)(wr0 + rloc) | ((unsigned long long)e.v << 32);
      }
    };
    {
      Ent fs[8];
      t4_for<8>([&](auto jc) { fs[decltype(jc)::value] = fetch(jc); });
      t4_for<8>([&](auto jc) { place(fs[decltype(jc)::value], 2 * decltype(jc)::value, 2 * decltype(jc)::value + 1); });
      t4_for<8>([&](auto jc) { fs[decltype(jc)::value] = fetch(std::integral_constant<int, 8 + decltype(jc)::value>{}); });
      t4_for<8>([&](auto jc) {
        constexpr int J = 8 + decltype(jc)::value;
        place(fs[decltype(jc)::value], 2 * J, 2 * J + 1);
      });
    }
    if (pa >= 0 && Ct > 32) place(fetch(std::integral_constant<int, kSlotP>{}), pa, pb >= 0 ? pb : pa);
    if (xa >= 0) place(fetch(std::integral_constant<int, kSlotX>{}), xa, xb >= 0 ? xb : xa);
    // the rows that overflowed here are the ones expected to overflow in the next tile
    pa = pb = -1;
    if (ov_all) {
      unsigned o2 = ov_all;
      pa = __builtin_ctz(o2);
      o2 &= o2 - 1u;
      if (o2) pb = __builtin_ctz(o2);
    }
    // this wave's cursors move on; its bitmap row is cleared for the next tile; the next windows are requested
    if constexpr (CIRC) {
      int padv = 0x7fffffff;
      asm volatile("" : "+v"(padv));
      t4_for<16>([&](auto jc) { t4_retire_slot<decltype(jc)::value>(cend, padv); });
    }
    if (half == 0) {
      A += (uint64_t)(unsigned)cntv * (PAIRS ? 8u : 4u);
      if (!PAIRS) A2 += (uint64_t)(unsigned)cntv * 4u;
      rem -= cntv;
      if (CIRC) ld = ld > cntv ? ld - cntv : 0;  // (a row that went into its continuation has nothing left in the window)
    }
    for (int t = lane; t < Ct; t += 64) bmw[t].x = 0u;
    issue_all();
    mark(3);
    __syncthreads();  // B3: the staged tile is complete
    // ---- phase 4: the write-out (tune tpack4_late = 1: one tile late, between the next tile's first two barriers) ----
    if (late) {
      pend_cols = cend - cbase;
      pend_set = tset;
      tset ^= 1;
    } else {
      write_out(tset, cend - cbase);
    }
    mark(4);
    cb += Ct;
    Ct = C;
  }
  // (the last staged tile; the barrier orders it behind every wave's phase 3 - it is the third barrier of that tile)
  if (pend_cols > 0) write_out(pend_set, pend_cols);
  t4_wait_all();  // (the windows requested for a tile that does not exist)
  if (DBG && tid == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_t4_phase[i], ph[i]);
}

inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
struct T4Geo {
  int rw;        // rows of a wave
  int64_t rpb;   // rows of a block
  int G;         // blocks
  int C;         // tile width
};
inline T4Geo t4_geometry(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  T4Geo q;
  const int64_t cus = mu_num_cus();
  int64_t rw = (n_rows + kNW * cus - 1) / (kNW * cus);
  if (rw < 1) rw = 1;
  if (rw > kRW) rw = kRW;
  q.rw = (int)rw;
  q.rpb = kNW * rw;
  q.G = (int)((n_rows + q.rpb - 1) / q.rpb);
  if (q.G < 1) q.G = 1;
  // tile width: the block's pairs of a tile fill ~88 % of the staging buffer, and a row has ~m entries in a tile
  // (tune tpack4_m, default 16: measured on the bench matrix - whose rows are burstier than Poisson - 14 / 16 / 18
  //  entries per row and tile retried 0.1 / 3 / 50 % of the tiles with ONE overflow slot per wave)
  const double per_col = (double)nnz / (double)q.G / (double)(n_cols > 0 ? n_cols : 1);
  double Cc = per_col > 0 ? 0.88 * kCap / per_col : (double)kMaxC;
  const int m = mu_tune_get("tpack4_m") > 0 ? mu_tune_get("tpack4_m") : 16;
  const double row_avg = (double)nnz / (double)(n_rows > 0 ? n_rows : 1);
  const double Cm = row_avg > 0 ? (double)m * (double)n_cols / row_avg : (double)kMaxC;
  if (Cm < Cc) Cc = Cm;
  int64_t C = (int64_t)Cc;
  C = (C / 32) * 32;
  if (C < 32) C = 32;
  if (C > kMaxC) C = kMaxC;
  if (mu_tune_get("tpack4_c") > 0) C = mu_tune_get("tpack4_c");
  if (C > kMaxC) C = kMaxC;
  if (C < 16) C = 16;
  q.C = (int)C;
  return q;
}
struct T4Work {
  int64_t* sp;
  uint32_t* cnt;
  int64_t* coltot;
  int64_t* cdst;
  int* err;
};
inline size_t t4_worksize(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  const int64_t S = (n_cols + kCountSlab - 1) / kCountSlab;
  const T4Geo q = t4_geometry(n_rows, n_cols, nnz);
  return al((size_t)(n_rows * (S + 1)) * sizeof(int64_t)) + al((size_t)(q.G + 1) * (size_t)n_cols * sizeof(uint32_t)) +
         2 * al((size_t)n_cols * sizeof(int64_t)) + 256 + 256;
}
struct T4Off {  // byte offsets of the pieces inside the work buffer
  size_t cnt, coltot, cdst, err;
};
inline T4Off t4_offsets(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  const int64_t S = (n_cols + kCountSlab - 1) / kCountSlab;
  const T4Geo q = t4_geometry(n_rows, n_cols, nnz);
  T4Off o;
  o.cnt = al((size_t)(n_rows * (S + 1)) * sizeof(int64_t));
  o.coltot = o.cnt + al((size_t)(q.G + 1) * (size_t)n_cols * sizeof(uint32_t));
  o.cdst = o.coltot + al((size_t)n_cols * sizeof(int64_t));
  o.err = o.cdst + al((size_t)n_cols * sizeof(int64_t));
  return o;
}
inline T4Work t4_carve(void* work, int64_t n_rows, int64_t n_cols, int64_t nnz) {
  const T4Off o = t4_offsets(n_rows, n_cols, nnz);
  char* w = (char*)work;
  T4Work t;
  t.sp = (int64_t*)w;
  t.cnt = (uint32_t*)(w + o.cnt);
  t.coltot = (int64_t*)(w + o.coltot);
  t.cdst = (int64_t*)(w + o.cdst);
  t.err = (int*)(w + o.err);
  return t;
}

int t4_fill_impl(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr, const int32_t* d_indices,
                 const float* d_values, const int64_t* d_row_dst, const void* d_x_ent, const int64_t* d_cptr,
                 const int32_t* d_inv, T4Out out, void* d_work, hipStream_t st) {
  const T4Geo q = t4_geometry(n_rows, n_cols, nnz);
  const T4Work w = t4_carve(d_work, n_rows, n_cols, nnz);
  hipLaunchKernelGGL(k_t4_cdst, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, n_cols, d_cptr, d_inv,
                     w.cdst);
  MU_CHECK_LAUNCH();
  const bool dbg = mu_tune_get("tpack_dbg") > 0;
  const bool xcd = mu_tune_get("tpack4_plain") != 1;  // (tune tpack4_plain = 1: workgroup = row block, for comparison)
  const unsigned grid = xcd ? (unsigned)(8 * ((q.G + 7) / 8)) : (unsigned)q.G;
#define MU_T4_LAUNCH(PAIRS_, DBG_, CSR_, RD_, S0_, S1_) MU_T4_LAUNCH4(PAIRS_, DBG_, CSR_, false, RD_, S0_, S1_)
#define MU_T4_LAUNCH4(PAIRS_, DBG_, CSR_, CIRC_, RD_, S0_, S1_)                                                        \
  hipLaunchKernelGGL((k_t4_fill<PAIRS_, DBG_, CSR_, CIRC_>), dim3(grid), dim3(kT), 0, st, n_rows, n_cols, q.C, q.rw,    \
                     xcd ? q.G : -q.G, d_indptr, RD_, (const void*)(S0_), (const void*)(S1_), w.cdst, w.cnt, w.coltot,  \
                     out, w.err, mu_tune_get("tpack4_abl") | (mu_tune_get("tpack4_late") == 1 ? 8 : 0))
  const bool csr_out = out.idx != nullptr;
  const int64_t* no_rd = nullptr;
  // (circular windows: the row stream as source, 256-byte aligned - the lane of a pair is its number mod 32
  //  - the default; tune tpack4_circ = 2: plain windows, for comparison)
  const bool circ = mu_tune_get("tpack4_circ") != 2 && (reinterpret_cast<uintptr_t>(d_x_ent) & 255) == 0;
  if (d_x_ent && dbg && !csr_out) MU_T4_LAUNCH(true, true, false, d_row_dst, d_x_ent, nullptr);
  else if (d_x_ent && !csr_out && circ) MU_T4_LAUNCH4(true, false, false, true, d_row_dst, d_x_ent, nullptr);
  else if (d_x_ent && !csr_out) MU_T4_LAUNCH(true, false, false, d_row_dst, d_x_ent, nullptr);
  else if (d_x_ent) MU_T4_LAUNCH(true, false, true, d_row_dst, d_x_ent, nullptr);
  else if (dbg && !csr_out) MU_T4_LAUNCH(false, true, false, no_rd, d_indices, d_values);
  else if (!csr_out) MU_T4_LAUNCH(false, false, false, no_rd, d_indices, d_values);
  else MU_T4_LAUNCH(false, false, true, no_rd, d_indices, d_values);
#undef MU_T4_LAUNCH
#undef MU_T4_LAUNCH4
  MU_CHECK_LAUNCH();
  return MU_OK;
}

}  // namespace

int launch_slab_ptr(int64_t n_rows, int64_t n_cols, const int64_t* indptr, const int32_t* indices, int64_t* sp,
                    hipStream_t stream);  // csrc/tfidf.hip (the same 8192-column slabs)

extern "C" {

/* 1 when the fourth-generation transposition can take the shape: row ids and block-relative offsets in 32 bits */
int mu_tpack4_supported(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  if (n_rows <= 0 || n_cols <= 0 || nnz <= 0) return 0;
  if (n_rows >= ((int64_t)1 << 31) || n_cols >= ((int64_t)1 << 31) - 1) return 0;
  const T4Geo q = t4_geometry(n_rows, n_cols, nnz);
  // the count sweep addresses a block's entries with 32-bit byte offsets
  if ((double)q.rpb * (double)n_cols >= (double)((int64_t)1 << 29) && nnz >= ((int64_t)1 << 29)) return 0;
  return 1;
}

int mu_tpack4_geometry(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t* rows_per_block, int* n_blocks,
                       int* tile_cols) {
  const T4Geo q = t4_geometry(n_rows, n_cols, nnz);
  if (rows_per_block) *rows_per_block = q.rpb;
  if (n_blocks) *n_blocks = q.G;
  if (tile_cols) *tile_cols = q.C;
  return MU_OK;
}

size_t mu_tpack4_worksize(int64_t n_rows, int64_t n_cols, int64_t nnz) { return t4_worksize(n_rows, n_cols, nnz); }

int mu_tpack4_count(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr, const int32_t* d_indices,
                    int64_t* d_col_nnz, void* d_work, size_t work_bytes, const int64_t* d_slab_ptr, void* stream) {
  MU_REQUIRE(mu_tpack4_supported(n_rows, n_cols, nnz), "shape out of range (mu_tpack4_supported)");
  MU_REQUIRE(d_indptr && d_indices && d_col_nnz && d_work, "null pointer");
  MU_REQUIRE(work_bytes >= t4_worksize(n_rows, n_cols, nnz), "work buffer too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t S = (n_cols + kCountSlab - 1) / kCountSlab;
  const T4Geo q = t4_geometry(n_rows, n_cols, nnz);
  const T4Work w = t4_carve(d_work, n_rows, n_cols, nnz);
  MU_CHECK_HIP(hipMemsetAsync(w.err, 0, sizeof(int), st));
  if (!d_slab_ptr) {
    const int rc = launch_slab_ptr(n_rows, n_cols, d_indptr, d_indices, w.sp, st);
    if (rc) return rc;
  }
  hipLaunchKernelGGL((k_t4_count<4>), dim3(q.G), dim3(kT), 0, st, n_rows, n_cols, S, q.rpb, d_indptr, d_indices,
                     d_slab_ptr ? d_slab_ptr : w.sp, w.cnt);
  MU_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_t4_base, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, n_cols, q.G, w.cnt, w.coltot,
                     d_col_nnz);
  MU_CHECK_LAUNCH();
  return MU_OK;
}

int mu_tpack4_fill_stream(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr,
                          const int32_t* d_indices, const float* d_values, const int64_t* d_x_row_dst,
                          const void* d_x_ent, const int64_t* d_sptr, const int32_t* d_inv, void* d_ent, void* d_work,
                          size_t work_bytes, void* stream) {
  MU_REQUIRE(mu_tpack4_supported(n_rows, n_cols, nnz), "shape out of range (mu_tpack4_supported)");
  MU_REQUIRE(d_indptr && d_sptr && d_ent && d_work, "null pointer");
  MU_REQUIRE((d_x_ent && d_x_row_dst) || (d_indices && d_values), "neither a row stream nor CSR arrays to read");
  MU_REQUIRE(work_bytes >= t4_worksize(n_rows, n_cols, nnz), "work buffer too small");
  const T4Out out{(unsigned long long*)d_ent, nullptr, nullptr};
  return t4_fill_impl(n_rows, n_cols, nnz, d_indptr, d_indices, d_values, d_x_row_dst, d_x_ent, d_sptr, d_inv, out,
                      d_work, (hipStream_t)stream);
}

int mu_tpack4_fill_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr, const int32_t* d_indices,
                       const float* d_values, const int64_t* d_x_row_dst, const void* d_x_ent,
                       const int64_t* d_t_indptr, int32_t* d_t_indices, float* d_t_values, void* d_work,
                       size_t work_bytes, void* stream) {
  MU_REQUIRE(mu_tpack4_supported(n_rows, n_cols, nnz), "shape out of range (mu_tpack4_supported)");
  MU_REQUIRE(d_indptr && d_t_indptr && d_t_indices && d_t_values && d_work, "null pointer");
  MU_REQUIRE((d_x_ent && d_x_row_dst) || (d_indices && d_values), "neither a row stream nor CSR arrays to read");
  MU_REQUIRE(work_bytes >= t4_worksize(n_rows, n_cols, nnz), "work buffer too small");
  const T4Out out{nullptr, d_t_indices, d_t_values};
  return t4_fill_impl(n_rows, n_cols, nnz, d_indptr, d_indices, d_values, d_x_row_dst, d_x_ent, d_t_indptr, nullptr,
                      out, d_work, (hipStream_t)stream);
}

/* byte offset of the fill's error word inside d_work: callers that must not synchronise copy the word to the host with
 * their next fetch (muon_amd/_backend.py: tpack4 fills are checked at lsi's first Gram fetch) */
size_t mu_tpack4_err_offset(int64_t n_rows, int64_t n_cols, int64_t nnz) { return t4_offsets(n_rows, n_cols, nnz).err; }

/* byte offset inside d_work of the count pass' prefix table: uint32 cnt[(n_blocks + 1)][n_cols], cnt[g][c] = entries of
 * column c in the row blocks before g (row n_blocks: the column totals).  The fill only reads it: the cells of a row-block
 * range are a contiguous piece of every row of X^T and this table says where it begins (mu_spmm_stream_ranges_f32) */
size_t mu_tpack4_cnt_offset(int64_t n_rows, int64_t n_cols, int64_t nnz) { return t4_offsets(n_rows, n_cols, nnz).cnt; }

/* the fill's error word (0 = fine; 1: bitmap and count pass disagreed, 2: a tile could not be narrowed) - synchronises */
int mu_tpack4_status(const void* d_work, int64_t n_rows, int64_t n_cols, int64_t nnz, int* h_err) {
  MU_REQUIRE(d_work && h_err, "null pointer");
  const T4Work w = t4_carve(const_cast<void*>(d_work), n_rows, n_cols, nnz);
  MU_CHECK_HIP(hipDeviceSynchronize());
  MU_CHECK_HIP(hipMemcpy(h_err, w.err, sizeof(int), hipMemcpyDeviceToHost));
  return MU_OK;
}

/* tune tpack_dbg = 1: cycles (s_memtime, wave 0 of every block) of header / phase 1 / 2 / 3 / write-out, [5] = retried tiles */
int mu_tpack4_phase_cycles(unsigned long long* h_out6, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (reset) {
    MU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_t4_phase), z, sizeof(z)));
    return MU_OK;
  }
  MU_CHECK_HIP(hipDeviceSynchronize());
  MU_CHECK_HIP(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_t4_phase), sizeof(z)));
  for (int i = 0; i < 6; ++i) h_out6[i] = z[i];
  return MU_OK;
}

}  // extern "C"
