// SpMM on the row stream for a NARROW dense block (f32):  Y[n x 16] = X[n x d] * Q[d x 16].
//
// MOFA's sparse views multiply by a factor block of <= 16 columns twice per iteration (A = Y (tau o W)
// and B = Y^T Z: mofapy2's Z / W node updates reached from /root/reference/muon/_core/tools.py:585);
// half of an iteration of BASELINE configs[3] went into these two products with the B = 64 kernel's
// lane layout (csrc/spmm_win.hip: a 16-lane group per row, ONE dense column per lane, three VALU and
// one LDS instruction per 4 stored entries, a window bookkeeping pass per 8 entries).  With 16 columns
// a Q row is 64 bytes, so the same 128 KiB of LDS hold slabs of 1024 columns instead of 256 and a row
// has ~32 entries per slab instead of 8.  This kernel is built around that:
//
//   * lane (c, e) = 16 c + e owns dense columns 4c..4c+3 (one ds_read_b128) of window slot e: one
//     instruction group handles 16 stored entries of ONE row (16 x 4 lanes), a wave walks its rows one
//     after the other and keeps a float4 partial sum per row and lane (summed over the 16 slots with
//     DPP row rotations when the sweep is over);
//   * the window of a (row, slab) visit is ONE 512-byte request: lane l reads pair cursor + l.  Rows
//     are sorted, so the entries of the slab are a prefix: compare, ballot, popcount, advance the
//     cursor (cursors live in the lanes of one register: v_readlane / v_writelane), request the row's
//     next window right away - it is needed one slab sweep later;
//   * entry 16 w + e has to reach the four lanes (c, e): v_permlane16_swap + 2 v_permlane32_swap turn a
//     register into its four 16-lane rows replicated (no LDS traffic, 6 VALU per value and 64 entries);
//   * Q slabs by LDS-DMA, double buffered, as in spmm_win.hip; a wave's four 1 KiB pieces go out in
//     front of its first four rows and are older than every window request after them, so
//     s_waitcnt vmcnt(RW - 3) before the slab barrier proves they landed without draining the requests;
//   * everything else is plain C++: the window requests are unconditional loads (clamped index), so
//     hipcc's own counted waits are exact in the unrolled row loop.
//
// Same operand (row stream in launch order, csrc/spmm_win.hip), same workgroup extents (64 K positions,
// wave w owns 4 K consecutive positions); a wave sweeps its rows RW at a time (accumulator registers).
// The sum of a row is taken in a fixed order that depends on the row alone (slot partial sums in entry
// order, then a fixed rotation tree): bit-reproducible and independent of the layout.
#include <utility>

#include "common.hpp"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"  // 32-bit LDS addresses made from integers

namespace {

constexpr int kNSlab = 1024;             // Q rows per slab
constexpr int kNSlabBytes = kNSlab * 64; // 64 KiB, double buffered
constexpr int kNW = 16;                  // waves per workgroup

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <int... I, class F>
__device__ __forceinline__ void n_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void n_static_for(F&& f) {
  n_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// one LDS-DMA piece: 64 lanes x 16 B land contiguously at the wave-uniform LDS byte address
__device__ __forceinline__ void n_dma_piece(const void* base, unsigned byte_off, unsigned lds_dst) {
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

// The four 16-lane rows of a register, each replicated into all four rows:
// x[w] in lane 16 c + e  =  the input's lane 16 w + e.
struct Rows4 { unsigned x[4]; };
__device__ __forceinline__ Rows4 rows4(unsigned v) {
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);        // [0 0 2 2], [1 1 3 3]
  const auto a = __builtin_amdgcn_permlane32_swap(r[0], r[0], false, false);  // [0 0 0 0], [2 2 2 2]
  const auto b = __builtin_amdgcn_permlane32_swap(r[1], r[1], false, false);  // [1 1 1 1], [3 3 3 3]
  Rows4 o;
  o.x[0] = a[0];
  o.x[1] = b[0];
  o.x[2] = a[1];
  o.x[3] = b[1];
  return o;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

// ABL: timing ablations (wrong results on purpose): 1 only lanes 0-31 request their pair, 2 no gathers / FMAs,
// 3 every request fetches the NEXT 32 pairs of one sequential stream per wave (what a slab-major operand with a
// count table would ask the memory system for), 4 cycle accounting (s_memtime per wave: window wait + count,
// spread, gathers + FMAs, slab barrier; the sums replace the product)
// STAGE: a window's (LDS offset, value) pairs reach their four lanes through a per-wave LDS staging row
// (one ds_write_b64 + a broadcast ds_read_b64 per 16 entries; lane 4 e + c) instead of the lane swaps
// (lane 16 c + e): 12 VALU instructions per visit less, 3.5 cheap LDS instructions more.
template <int RW, int ABL = 0, bool STAGE = false>
__global__ __launch_bounds__(1024) void k_spmm_narrow(int64_t n_pos, int64_t n_cols, int K,
                                                      const int64_t* __restrict__ sptr,
                                                      const unsigned long long* __restrict__ ent,
                                                      const int32_t* __restrict__ perm,
                                                      const float* __restrict__ Q, float* __restrict__ Y) {
  static_assert(RW >= 4 && RW <= 12, "a sweep covers 4 .. 12 rows per wave");
  constexpr bool HALF = ABL == 1 || ABL == 3;
  typedef __attribute__((address_space(3))) const f4* lds_p;
  __shared__ __attribute__((aligned(65536))) f4 qs[2][kNSlabBytes / 16];  // Q row j of a slab at byte 64 j (64 KiB aligned: the XOR addressing)
  __shared__ unsigned long long stage[STAGE ? kNW : 1][64];
  const int lane = threadIdx.x & 63;
  const int wave = uniform32(threadIdx.x >> 6);
  const int c = STAGE ? (lane & 3) : (lane >> 4);  // this lane's column quad
  const int rows_w = 4 * K;  // positions per wave
  const int64_t rb0 = (int64_t)blockIdx.x * (64 * (int64_t)K);
  const int64_t rb1 = (rb0 + 64 * (int64_t)K) < n_pos ? (rb0 + 64 * (int64_t)K) : n_pos;
  const int64_t pw0 = rb0 + (int64_t)wave * rows_w;
  const int64_t wg0 = uniform64(sptr[rb0]);
  const unsigned wg_n = (unsigned)(uniform64(sptr[rb1]) - wg0);  // stored entries of this workgroup (host: < 2^29)
  const unsigned long long* __restrict__ entw = ent + wg0;      // wave-uniform
  const unsigned wg_last = wg_n ? wg_n - 1u : 0u;
  const int ncols32 = (int)n_cols;
  const unsigned qs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&qs[0][0]);
  if ((qs_lds & 0xffffu) != 0u) __builtin_trap();  // the XOR addressing below needs the buffers 64 KiB aligned
  const unsigned q_last = (unsigned)(n_cols * 64 - 16);  // byte offset of Q's last 16 bytes

  auto store_row = [&](int r_abs, f4 v) {  // lanes e == 0 hold the sums of columns 4c..4c+3
    const int64_t p = pw0 + r_abs;
    if (r_abs < rows_w && p < rb1 && (STAGE ? lane < 4 : (lane & 15) == 0)) {
      const int64_t out = perm ? (int64_t)perm[p] : p;
      if (out >= 0) *reinterpret_cast<f4*>(Y + out * 16 + 4 * c) = v;
    }
  };
  if (wg_n == 0) {  // uniform over the workgroup: only empty rows
    for (int r = 0; r < rows_w; ++r) store_row(r, (f4)(0.f));
    return;
  }

  // Q row j of a slab sits at LDS byte 64 j with its four 16-byte quads XOR-swizzled: logical quad c at
  // physical position c ^ ((j >> 2) & 3).  The 16 lanes that read quad c of 16 different rows would
  // otherwise all fall into the four 16-byte bank groups {c, c+4, c+8, c+12} (measured: 8 of 12 LDS cycles
  // per ds_read_b128 were bank conflicts); swizzled they spread over all sixteen.
  auto dma_one = [&](int s0, int buf, int u) {  // 1 KiB piece u of this wave: lane l lands at piece * 1024 + 16 l
    const int piece = wave + u * kNW;
    const int j = piece * 16 + (lane >> 2);                       // Q row inside the slab
    const int cq = (lane & 3) ^ ((j >> 2) & 3);                   // the logical quad that belongs there
    unsigned off = (unsigned)s0 * 64u + (unsigned)(j * 64 + cq * 16);
    off = off < q_last ? off : q_last;  // tail / past the end: clamp (never consumed)
    n_dma_piece(Q, off, qs_lds + (unsigned)buf * (unsigned)kNSlabBytes + (unsigned)piece * 1024u);
  };
  const char* __restrict__ entb = reinterpret_cast<const char*>(entw);
  const unsigned lane8 = (unsigned)lane * 8u, last8 = wg_last * 8u;
  unsigned seq = 0;  // (ABL 3)
  unsigned long long t_win = 0, t_spread = 0, t_back = 0, t_bar = 0;  // (ABL 4)
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  auto request = [&](unsigned cur, int& col, float& val) {  // lane l: pair cur + l (clamped into the workgroup's stream)
    unsigned off = cur * 8u + lane8;
    if constexpr (ABL == 3) {
      off = ((unsigned)wave * (wg_n / kNW) + seq) * 8u + lane8;
      seq += 32u;
      if (seq + 64u > wg_n / kNW) seq = 0;
    }
    off = off < last8 ? off : last8;
    unsigned long long e = 0x000000007fffffffull;
    if (!HALF || lane < 32)
      e = *reinterpret_cast<const unsigned long long*>(entb + (size_t)off);  // scalar base + 32-bit offset
    col = (int)(unsigned)e;
    val = __builtin_bit_cast(float, (unsigned)(e >> 32));
  };
  // x where bit `lane` of the wave-uniform mask is set, else 0 (the mask rides in an SGPR pair: no v_cmp)
  auto keep = [&](unsigned x, unsigned long long mask) -> unsigned {
    unsigned o;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(o) : "v"(x), "s"(mask));
    return o;
  };

  for (int r0 = 0; r0 < rows_w; r0 += RW) {  // uniform over the workgroup
    unsigned cur[RW], end[RW];  // cursor / end of row r of this sweep (pairs from wg0): scalar registers
    {
      unsigned curv = 0, endv = 0;
      const int64_t p = pw0 + r0 + lane;
      if (lane < RW && r0 + lane < rows_w && p < rb1) {
        curv = (unsigned)(sptr[p] - wg0);
        endv = (unsigned)(sptr[p + 1] - wg0);
      }
      n_static_for<RW>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        cur[r] = (unsigned)__builtin_amdgcn_readlane((int)curv, r);
        end[r] = (unsigned)__builtin_amdgcn_readlane((int)endv, r);
      });
    }
    f4 acc[RW];
    int wcol[RW];
    float wval[RW];
    n_static_for<RW>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      acc[r] = (f4)(0.f);
      request(cur[r], wcol[r], wval[r]);
    });
#pragma unroll
    for (int u = 0; u < 4; ++u) dma_one(0, 0, u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int buf = 0;
    for (int s0 = 0; s0 < ncols32; s0 += kNSlab, buf ^= 1) {
      const int s_hi = (s0 + kNSlab) < ncols32 ? (s0 + kNSlab) : ncols32;
      // LDS address of (row j, logical quad c) = qx ^ (64 j | 16 ((j >> 2) & 3)); the buffer base has no
      // bits below 2^16 in common with the row part (checked: qs is the kernel's only LDS object)
      const unsigned qx = qs_lds + (unsigned)buf * (unsigned)kNSlabBytes + (unsigned)c * 16u;
      const unsigned neg = 0u - (unsigned)s0 * 64u;
      unsigned again = 0;
      // A visit of row r in two stages.  front: count the window's entries of this slab, advance, request the
      // row's next window, spread the entries (LDS offset, value) over their four lanes.  back: gather the Q
      // rows, accumulate.  The row loop runs front(r + 1) BEFORE back(r): a wave issues in order, and one
      // visit is a chain of four dependent round trips (window -> staging row -> entries -> Q rows); with
      // both stages of a visit back to back a wave spent ~900 cycles per visit, most of them waiting, with
      // the VALU 42 % and the LDS 67 % busy.
      // What front hands to back: a0..a3 = LDS offset (swizzle bits included) of window entry 16 w + e,
      // v0..v3 = its value (0: not of this slab), n = entries of this slab in the window (uniform).
      // (Individual scalars passed by reference: hipcc kept half of a struct - also of a struct with scalar
      //  members only - in memory, one store and one scratch load per visit.)
#define MU_SPREAD_PARAMS unsigned &a0, unsigned &a1, unsigned &a2, unsigned &a3, float &v0, float &v1, float &v2, float &v3, int &n_out
      auto front = [&](auto rc, MU_SPREAD_PARAMS) {
        constexpr int r = decltype(rc)::value;
        unsigned long long tc0 = 0;
        if constexpr (ABL == 4) tc0 = __builtin_amdgcn_s_memtime();
        const int col = wcol[r];
        const float val = wval[r];
        // sorted rows: the entries of this slab are a prefix of the lanes still inside the row
        const unsigned left = end[r] - cur[r];
        const unsigned long long rowmask = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(col < s_hi) & rowmask;
        const int n = __builtin_amdgcn_readfirstlane(__builtin_popcountll(m));
        n_out = n;
        if constexpr (ABL == 4) {
          const unsigned long long tc1 = __builtin_amdgcn_s_memtime();
          t_win += tc1 - tc0;
          tc0 = tc1;
        }
        cur[r] += (unsigned)n;
        request(cur[r], wcol[r], wval[r]);
        if (n == 64) again |= 1u << r;  // the row goes on in this slab: its next window is the continuation
        if (n == 0) {  // uniform: nothing of this row in the slab (an empty slot of the layout, a short row)
          a0 = a1 = a2 = a3 = 0u;
          v0 = v1 = v2 = v3 = 0.f;
          return;
        }
        // 64 j + 16 ((j >> 2) & 3) with j = col - s0 (s0 is a multiple of 1024: the swizzle bits are col's own)
        unsigned a = ((unsigned)col << 6) + neg;
        a |= ((unsigned)col & 12u) << 2;
        if (HALF) a &= 0xffffu;
        if constexpr (STAGE) {
          typedef __attribute__((address_space(3))) unsigned long long* st_p;
          const st_p row = (st_p)(&stage[wave][0]);
          // (a pair of 32-bit vector elements would do; hipcc 7.2 folds "element 1 of the loaded pair" into
          //  the packed FMA's op_sel bits and picks element 0 - the offset was multiplied, not the value)
          row[lane] = (unsigned long long)keep(a, m) | ((unsigned long long)keep(__builtin_bit_cast(unsigned, val), m) << 32);
          __builtin_amdgcn_wave_barrier();  // (LDS operations of a wave execute in order: no wait between write and reads)
          const unsigned long long e0 = row[(lane >> 2)], e1 = row[16 + (lane >> 2)];
          const unsigned long long e2 = row[32 + (lane >> 2)], e3 = row[48 + (lane >> 2)];
          a0 = (unsigned)e0, a1 = (unsigned)e1, a2 = (unsigned)e2, a3 = (unsigned)e3;
          v0 = __builtin_bit_cast(float, (unsigned)(e0 >> 32));
          v1 = __builtin_bit_cast(float, (unsigned)(e1 >> 32));
          v2 = __builtin_bit_cast(float, (unsigned)(e2 >> 32));
          v3 = __builtin_bit_cast(float, (unsigned)(e3 >> 32));
          __builtin_amdgcn_wave_barrier();
        } else {
          const Rows4 A = rows4(keep(a, m));
          const Rows4 V = rows4(keep(__builtin_bit_cast(unsigned, val), m));
          a0 = A.x[0], a1 = A.x[1], a2 = A.x[2], a3 = A.x[3];
          v0 = __builtin_bit_cast(float, V.x[0]);
          v1 = __builtin_bit_cast(float, V.x[1]);
          v2 = __builtin_bit_cast(float, V.x[2]);
          v3 = __builtin_bit_cast(float, V.x[3]);
        }
        if constexpr (ABL == 4) {
          asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
          t_spread += __builtin_amdgcn_s_memtime() - tc0;
        }
      };
      // (the empty asm statements: the row's FMAs happen HERE - hipcc would sink them to the end of the
      //  slab and keep the gathered Q rows of every row of the sweep alive, 500+ registers - and the
      //  branch stays a branch instead of eight FMAs and four selects)
      auto back = [&](auto rc, unsigned a0, unsigned a1, unsigned a2, unsigned a3, float v0, float v1, float v2,
                      float v3, int n) {
        constexpr int r = decltype(rc)::value;
        if constexpr (ABL == 2) {
          acc[r][0] += v0 + v1 + __builtin_bit_cast(float, a0 ^ a1);
          return;
        }
        unsigned long long tb0 = 0;
        if constexpr (ABL == 4) tb0 = __builtin_amdgcn_s_memtime();
        const f4 q0 = *(lds_p)(a0 ^ qx);
        const f4 q1 = *(lds_p)(a1 ^ qx);
        if (n > 32) {  // uniform
          const f4 q2 = *(lds_p)(a2 ^ qx);
          const f4 q3 = *(lds_p)(a3 ^ qx);
          acc[r] += v0 * q0;
          acc[r] += v1 * q1;
          acc[r] += v2 * q2;
          acc[r] += v3 * q3;
          asm volatile("" : "+v"(acc[r]));
        } else {
          acc[r] += v0 * q0;
          acc[r] += v1 * q1;
          asm volatile("" : "+v"(acc[r]));
        }
        if constexpr (ABL == 4) t_back += __builtin_amdgcn_s_memtime() - tb0;
      };
      // Three stages per row, in this order inside a turn: the gathers of row r go out, the front of row r + 1
      // runs under their latency (it has no LDS operation: the spread is lane swaps), then the FMAs of row r.
      // (cycle accounting of the two-stage order: gathers + FMAs 39 % of a wave's time, most of it the wait)
      auto gather = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3, int n, f4& q0, f4& q1, f4& q2, f4& q3) {
        if (n == 0) return;  // uniform
        q0 = *(lds_p)(a0 ^ qx);
        q1 = *(lds_p)(a1 ^ qx);
        if (n > 32) {  // uniform
          q2 = *(lds_p)(a2 ^ qx);
          q3 = *(lds_p)(a3 ^ qx);
        }
        __builtin_amdgcn_sched_barrier(0);  // (the reads stay in front of the next row's front stage)
      };
      auto fmas = [&](auto rc, const f4& q0, const f4& q1, const f4& q2, const f4& q3, float v0, float v1, float v2,
                      float v3, int n) {
        constexpr int r = decltype(rc)::value;
        if (n > 32) {  // uniform
          acc[r] += v0 * q0;
          acc[r] += v1 * q1;
          acc[r] += v2 * q2;
          acc[r] += v3 * q3;
          asm volatile("" : "+v"(acc[r]));
        } else if (n > 0) {
          acc[r] += v0 * q0;
          acc[r] += v1 * q1;
          asm volatile("" : "+v"(acc[r]));
        }
      };
      {
        unsigned ea0, ea1, ea2, ea3, oa0, oa1, oa2, oa3;  // even / odd rows
        float ev0, ev1, ev2, ev3, ov0, ov1, ov2, ov3;
        int en, on;
        dma_one(s0 + kNSlab, buf ^ 1, 0);  // the next slab (the last slab's pieces are never read)
        front(std::integral_constant<int, 0>{}, ea0, ea1, ea2, ea3, ev0, ev1, ev2, ev3, en);
        n_static_for<RW>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          constexpr bool kPipe3 = (ABL == 0) && !STAGE;
          f4 q0, q1, q2, q3;
          if constexpr (kPipe3) {
            if constexpr (r & 1) gather(oa0, oa1, oa2, oa3, on, q0, q1, q2, q3);
            else gather(ea0, ea1, ea2, ea3, en, q0, q1, q2, q3);
          }
          if constexpr (r + 1 < RW) {
            if constexpr (r + 1 < 4) dma_one(s0 + kNSlab, buf ^ 1, r + 1);
            if constexpr ((r + 1) & 1) front(std::integral_constant<int, r + 1>{}, oa0, oa1, oa2, oa3, ov0, ov1, ov2, ov3, on);
            else front(std::integral_constant<int, r + 1>{}, ea0, ea1, ea2, ea3, ev0, ev1, ev2, ev3, en);
          }
          if constexpr (kPipe3) {
            if constexpr (r & 1) fmas(rc, q0, q1, q2, q3, ov0, ov1, ov2, ov3, on);
            else fmas(rc, q0, q1, q2, q3, ev0, ev1, ev2, ev3, en);
          } else {
            if constexpr (r & 1) back(rc, oa0, oa1, oa2, oa3, ov0, ov1, ov2, ov3, on);
            else back(rc, ea0, ea1, ea2, ea3, ev0, ev1, ev2, ev3, en);
          }
        });
      }
      while (again) {  // rows with 64 and more entries in one slab (uniform, rare)
        const unsigned pend = again;
        again = 0;
        n_static_for<RW>([&](auto rc) {
          if (pend & (1u << decltype(rc)::value)) {
            unsigned a0, a1, a2, a3;
            float v0, v1, v2, v3;
            int n;
            front(rc, a0, a1, a2, a3, v0, v1, v2, v3, n);
            back(rc, a0, a1, a2, a3, v0, v1, v2, v3, n);
          }
        });
      }
#undef MU_SPREAD_PARAMS
      // the DMA pieces are older than the window requests of rows 3 .. RW-1 (and of every revisit)
      unsigned long long tq0 = 0;
      if constexpr (ABL == 4) tq0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(RW - 3) : "memory");
      __syncthreads();  // next slab visible; everyone finished reading this one
      if constexpr (ABL == 4) t_bar += __builtin_amdgcn_s_memtime() - tq0;
    }

    // slot partial sums -> row sums in a fixed order; lanes e == 0 store
    n_static_for<RW>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      f4 v = acc[r];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float x = v[k];
        if constexpr (STAGE) {  // lane 4 e + c: slots 4 apart inside a 16-lane row, then the four rows
          x += dpp_f<0x128>(x);  // row_ror:8
          x += dpp_f<0x124>(x);  // row_ror:4
          x += __shfl_xor(x, 16, 64);
          x += __shfl_xor(x, 32, 64);
        } else {               // lane 16 c + e: a rotation tree inside every 16-lane row
          x += dpp_f<0x128>(x);  // row_ror:8
          x += dpp_f<0x124>(x);  // row_ror:4
          x += dpp_f<0x122>(x);  // row_ror:2
          x += dpp_f<0x121>(x);  // row_ror:1
        }
        v[k] = x;
      }
      store_row(r0 + r, v);
    });
  }
  if constexpr (ABL == 4) {  // wave `w` of workgroup `b`: row 16 b + w of Y holds the sums
    if (lane < 5 && blockIdx.x * 16 + wave < n_pos) {
      const unsigned long long v = lane == 0 ? t_win : lane == 1 ? t_spread : lane == 2 ? t_back : lane == 3 ? t_bar
                                                                 : __builtin_amdgcn_s_memtime() - t_begin;
      Y[((int64_t)blockIdx.x * 16 + wave) * 16 + lane] = (float)v;
    }
  }
}

}  // namespace

// rows a wave sweeps at a time for a layout of K row-sets (4 K positions per wave): at most 12 (the
// accumulators and windows of 12 rows take 125 registers, 14 spill)
static int narrow_rw(int K) {
  switch (K) {
    case 1: return 4;
    case 2: return 8;
    case 3: return 12;
    case 4: return 8;    // 2 sweeps
    case 5: return 10;   // 2
    case 6: return 12;   // 2
    case 7: return 10;   // 3: 10 + 10 + 8
    default: return 12;  // K = 8: 12 + 12 + 8
  }
}

int mu_spmm_narrow_f32_launch(hipStream_t st, int64_t n_pos, int64_t n_cols, int K, const int64_t* sptr,
                              const unsigned long long* ent, const int32_t* perm, const float* Q, float* Y) {
  const int64_t wgs = (n_pos + 64 * (int64_t)K - 1) / (64 * (int64_t)K);
#define MU_NARROW(RW_)                                                                                   \
  hipLaunchKernelGGL((k_spmm_narrow<RW_>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols, K, sptr, \
                     ent, perm, Q, Y)
  if (mu_tune_get("spmm_mode") == 3 && narrow_rw(K) == 10) {  // A/B: window entries spread through an LDS staging row
    hipLaunchKernelGGL((k_spmm_narrow<10, 0, true>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols, K, sptr,
                       ent, perm, Q, Y);
    MU_CHECK_LAUNCH();
    return MU_OK;
  }
  if (mu_tune_get("spmm_mode") == 6 && narrow_rw(K) == 10) {  // cycle accounting
    hipLaunchKernelGGL((k_spmm_narrow<10, 4>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols, K, sptr, ent, perm,
                       Q, Y);
    MU_CHECK_LAUNCH();
    return MU_OK;
  }
  if (mu_tune_get("spmm_mode") == 5 && narrow_rw(K) == 10) {  // ablation: sequential 32-pair requests per wave
    hipLaunchKernelGGL((k_spmm_narrow<10, 3>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols, K, sptr, ent, perm,
                       Q, Y);
    MU_CHECK_LAUNCH();
    return MU_OK;
  }
  if (mu_tune_get("spmm_mode") == 4 && narrow_rw(K) == 10) {  // ablation: no gathers / FMAs
    hipLaunchKernelGGL((k_spmm_narrow<10, 2>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols, K, sptr, ent, perm,
                       Q, Y);
    MU_CHECK_LAUNCH();
    return MU_OK;
  }
  if (mu_tune_get("spmm_mode") == 2 && narrow_rw(K) == 10) {
    hipLaunchKernelGGL((k_spmm_narrow<10, 1>), dim3((unsigned)wgs), dim3(1024), 0, st, n_pos, n_cols, K, sptr, ent,
                       perm, Q, Y);
    MU_CHECK_LAUNCH();
    return MU_OK;
  }
  switch (narrow_rw(K)) {
    case 4: MU_NARROW(4); break;
    case 8: MU_NARROW(8); break;
    case 10: MU_NARROW(10); break;
    default: MU_NARROW(12); break;
  }
#undef MU_NARROW
  MU_CHECK_LAUNCH();
  return MU_OK;
}
