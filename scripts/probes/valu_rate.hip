// How many cycles does a wave64 VALU instruction hold a SIMD of gfx950 for?  (r03: the SpMM's stage B
// is 4 VALU per e-step - v_add_u32_dpp, v_mov_b32_dpp, 2 x v_pk_fma_f32 - and VERDICT r02 proposes
// 4 x v_fmac_f32_dpp + 1 x v_add_u32_dpp instead.  Which is cheaper depends on whether a plain / DPP
// VALU op issues in 2 or 4 cycles and what v_pk_fma_f32 costs.)
// Each wave runs REPS x 16 independent instructions of one kind between two s_memtime reads; the
// kernel is launched with 1 and with 4 waves per SIMD on every CU.  Reported: cycles per instruction
// per SIMD (wave time / instructions per wave / waves per SIMD ... the SIMD is shared).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REPS 256
#define RUN16(body)                                                            \
  asm volatile(                                                                \
      "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 0.5\n v_mov_b32 v43, 0.5\n" \
      "v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n v_mov_b32 v46, 0.5\n v_mov_b32 v47, 0.5\n" \
      "v_mov_b32 v48, 1.0\n v_mov_b32 v49, 1.0\n v_mov_b32 v50, 0.5\n v_mov_b32 v51, 0.5\n" \
      "v_mov_b32 v52, 1.0\n v_mov_b32 v53, 1.0\n v_mov_b32 v54, 0.5\n v_mov_b32 v55, 0.5\n" \
      "v_mov_b32 v56, 1.0\n v_mov_b32 v57, 1.0\n v_mov_b32 v58, 0.5\n v_mov_b32 v59, 0.5\n" \
      "v_mov_b32 v60, 1.0\n v_mov_b32 v61, 1.0\n v_mov_b32 v62, 0.5\n v_mov_b32 v63, 0.5\n" \
      "v_mov_b32 v64, 1.0\n v_mov_b32 v65, 1.0\n v_mov_b32 v66, 0.5\n v_mov_b32 v67, 0.5\n" \
      "v_mov_b32 v68, 1.0\n v_mov_b32 v69, 1.0\n v_mov_b32 v70, 0.5\n v_mov_b32 v71, 0.5\n" \
      "v_mov_b32 v72, 0.25\n v_mov_b32 v73, 0.25\n v_mov_b32 v74, 0.125\n v_mov_b32 v75, 0.125\n" \
      "v_mbcnt_lo_u32_b32 v76, -1, 0\n v_mbcnt_hi_u32_b32 v76, -1, v76\n"         \
      "v_lshlrev_b32 v77, 3, v76\n v_lshlrev_b32 v76, 4, v76\n"                \
      "s_waitcnt lgkmcnt(0)\n s_barrier\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n"  \
      ".rept " "256" "\n" body ".endr\n"                                        \
      "s_nop 7\n s_memtime %1\n s_waitcnt lgkmcnt(0)\n"                          \
      : "=s"(t0), "=s"(t1)                                                    \
      :                                                                        \
      : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52",  \
        "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65",  \
        "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "memory")

#define DPPX " row_newbcast:3 row_mask:0xf bank_mask:0xf\n"

template <int OP>
__global__ void probe(unsigned long long* out) {
  __shared__ float4 lds[1024 + 64];  // 16 KiB + the largest offset's reach
  if (threadIdx.x == 0xffff) lds[0] = float4{0, 0, 0, 0};
  unsigned long long t0 = 0, t1 = 0;
  if (OP == 0) {  // 16 independent v_fma_f32
    RUN16("v_fma_f32 v40, v72, v74, v40\n v_fma_f32 v41, v72, v74, v41\n v_fma_f32 v42, v72, v74, v42\n v_fma_f32 v43, v72, v74, v43\n"
          "v_fma_f32 v44, v72, v74, v44\n v_fma_f32 v45, v72, v74, v45\n v_fma_f32 v46, v72, v74, v46\n v_fma_f32 v47, v72, v74, v47\n"
          "v_fma_f32 v48, v72, v74, v48\n v_fma_f32 v49, v72, v74, v49\n v_fma_f32 v50, v72, v74, v50\n v_fma_f32 v51, v72, v74, v51\n"
          "v_fma_f32 v52, v72, v74, v52\n v_fma_f32 v53, v72, v74, v53\n v_fma_f32 v54, v72, v74, v54\n v_fma_f32 v55, v72, v74, v55\n");
  } else if (OP == 1) {  // 16 independent v_pk_fma_f32
    RUN16("v_pk_fma_f32 v[40:41], v[72:73], v[74:75], v[40:41]\n v_pk_fma_f32 v[42:43], v[72:73], v[74:75], v[42:43]\n"
          "v_pk_fma_f32 v[44:45], v[72:73], v[74:75], v[44:45]\n v_pk_fma_f32 v[46:47], v[72:73], v[74:75], v[46:47]\n"
          "v_pk_fma_f32 v[48:49], v[72:73], v[74:75], v[48:49]\n v_pk_fma_f32 v[50:51], v[72:73], v[74:75], v[50:51]\n"
          "v_pk_fma_f32 v[52:53], v[72:73], v[74:75], v[52:53]\n v_pk_fma_f32 v[54:55], v[72:73], v[74:75], v[54:55]\n"
          "v_pk_fma_f32 v[56:57], v[72:73], v[74:75], v[56:57]\n v_pk_fma_f32 v[58:59], v[72:73], v[74:75], v[58:59]\n"
          "v_pk_fma_f32 v[60:61], v[72:73], v[74:75], v[60:61]\n v_pk_fma_f32 v[62:63], v[72:73], v[74:75], v[62:63]\n"
          "v_pk_fma_f32 v[64:65], v[72:73], v[74:75], v[64:65]\n v_pk_fma_f32 v[66:67], v[72:73], v[74:75], v[66:67]\n"
          "v_pk_fma_f32 v[68:69], v[72:73], v[74:75], v[68:69]\n v_pk_fma_f32 v[70:71], v[72:73], v[74:75], v[70:71]\n");
  } else if (OP == 2) {  // 16 independent v_mov_b32_dpp row_newbcast
    RUN16("v_mov_b32_dpp v40, v72" DPPX "v_mov_b32_dpp v41, v72" DPPX "v_mov_b32_dpp v42, v72" DPPX "v_mov_b32_dpp v43, v72" DPPX
          "v_mov_b32_dpp v44, v72" DPPX "v_mov_b32_dpp v45, v72" DPPX "v_mov_b32_dpp v46, v72" DPPX "v_mov_b32_dpp v47, v72" DPPX
          "v_mov_b32_dpp v48, v72" DPPX "v_mov_b32_dpp v49, v72" DPPX "v_mov_b32_dpp v50, v72" DPPX "v_mov_b32_dpp v51, v72" DPPX
          "v_mov_b32_dpp v52, v72" DPPX "v_mov_b32_dpp v53, v72" DPPX "v_mov_b32_dpp v54, v72" DPPX "v_mov_b32_dpp v55, v72" DPPX);
  } else if (OP == 3) {  // 16 independent v_add_u32_dpp
    RUN16("v_add_u32_dpp v40, v72, v74" DPPX "v_add_u32_dpp v41, v72, v74" DPPX "v_add_u32_dpp v42, v72, v74" DPPX "v_add_u32_dpp v43, v72, v74" DPPX
          "v_add_u32_dpp v44, v72, v74" DPPX "v_add_u32_dpp v45, v72, v74" DPPX "v_add_u32_dpp v46, v72, v74" DPPX "v_add_u32_dpp v47, v72, v74" DPPX
          "v_add_u32_dpp v48, v72, v74" DPPX "v_add_u32_dpp v49, v72, v74" DPPX "v_add_u32_dpp v50, v72, v74" DPPX "v_add_u32_dpp v51, v72, v74" DPPX
          "v_add_u32_dpp v52, v72, v74" DPPX "v_add_u32_dpp v53, v72, v74" DPPX "v_add_u32_dpp v54, v72, v74" DPPX "v_add_u32_dpp v55, v72, v74" DPPX);
  } else if (OP == 4) {  // 16 independent v_fmac_f32_dpp
    RUN16("v_fmac_f32_dpp v40, v72, v74" DPPX "v_fmac_f32_dpp v41, v72, v74" DPPX "v_fmac_f32_dpp v42, v72, v74" DPPX "v_fmac_f32_dpp v43, v72, v74" DPPX
          "v_fmac_f32_dpp v44, v72, v74" DPPX "v_fmac_f32_dpp v45, v72, v74" DPPX "v_fmac_f32_dpp v46, v72, v74" DPPX "v_fmac_f32_dpp v47, v72, v74" DPPX
          "v_fmac_f32_dpp v48, v72, v74" DPPX "v_fmac_f32_dpp v49, v72, v74" DPPX "v_fmac_f32_dpp v50, v72, v74" DPPX "v_fmac_f32_dpp v51, v72, v74" DPPX
          "v_fmac_f32_dpp v52, v72, v74" DPPX "v_fmac_f32_dpp v53, v72, v74" DPPX "v_fmac_f32_dpp v54, v72, v74" DPPX "v_fmac_f32_dpp v55, v72, v74" DPPX);
  } else if (OP == 5) {  // the e-step of stage B as shipped: add_dpp, mov_dpp, 2 pk_fma  (x4 = 16 instructions)
    RUN16("v_add_u32_dpp v40, v72, v74" DPPX "v_mov_b32_dpp v41, v73" DPPX "v_pk_fma_f32 v[44:45], v[56:57], v[74:75], v[44:45]\n v_pk_fma_f32 v[46:47], v[58:59], v[74:75], v[46:47]\n"
          "v_add_u32_dpp v42, v72, v74" DPPX "v_mov_b32_dpp v43, v73" DPPX "v_pk_fma_f32 v[48:49], v[56:57], v[74:75], v[48:49]\n v_pk_fma_f32 v[50:51], v[58:59], v[74:75], v[50:51]\n"
          "v_add_u32_dpp v60, v72, v74" DPPX "v_mov_b32_dpp v61, v73" DPPX "v_pk_fma_f32 v[52:53], v[56:57], v[74:75], v[52:53]\n v_pk_fma_f32 v[54:55], v[58:59], v[74:75], v[54:55]\n"
          "v_add_u32_dpp v62, v72, v74" DPPX "v_mov_b32_dpp v63, v73" DPPX "v_pk_fma_f32 v[64:65], v[56:57], v[74:75], v[64:65]\n v_pk_fma_f32 v[66:67], v[58:59], v[74:75], v[66:67]\n");
  } else if (OP == 6) {  // the proposed e-step: add_dpp + 4 fmac_dpp  (x3 = 15 instructions + 1 add)
    RUN16("v_add_u32_dpp v40, v72, v74" DPPX "v_fmac_f32_dpp v44, v73, v56" DPPX "v_fmac_f32_dpp v45, v73, v57" DPPX "v_fmac_f32_dpp v46, v73, v58" DPPX "v_fmac_f32_dpp v47, v73, v59" DPPX
          "v_add_u32_dpp v41, v72, v74" DPPX "v_fmac_f32_dpp v48, v73, v56" DPPX "v_fmac_f32_dpp v49, v73, v57" DPPX "v_fmac_f32_dpp v50, v73, v58" DPPX "v_fmac_f32_dpp v51, v73, v59" DPPX
          "v_add_u32_dpp v42, v72, v74" DPPX "v_fmac_f32_dpp v52, v73, v56" DPPX "v_fmac_f32_dpp v53, v73, v57" DPPX "v_fmac_f32_dpp v54, v73, v58" DPPX "v_fmac_f32_dpp v55, v73, v59" DPPX
          "v_add_u32_dpp v43, v72, v74" DPPX);
  } else if (OP == 7) {  // 16 independent v_pk_add_f32
    RUN16("v_pk_add_f32 v[40:41], v[72:73], v[40:41]\n v_pk_add_f32 v[42:43], v[72:73], v[42:43]\n"
          "v_pk_add_f32 v[44:45], v[72:73], v[44:45]\n v_pk_add_f32 v[46:47], v[72:73], v[46:47]\n"
          "v_pk_add_f32 v[48:49], v[72:73], v[48:49]\n v_pk_add_f32 v[50:51], v[72:73], v[50:51]\n"
          "v_pk_add_f32 v[52:53], v[72:73], v[52:53]\n v_pk_add_f32 v[54:55], v[72:73], v[54:55]\n"
          "v_pk_add_f32 v[56:57], v[72:73], v[56:57]\n v_pk_add_f32 v[58:59], v[72:73], v[58:59]\n"
          "v_pk_add_f32 v[60:61], v[72:73], v[60:61]\n v_pk_add_f32 v[62:63], v[72:73], v[62:63]\n"
          "v_pk_add_f32 v[64:65], v[72:73], v[64:65]\n v_pk_add_f32 v[66:67], v[72:73], v[66:67]\n"
          "v_pk_add_f32 v[68:69], v[72:73], v[68:69]\n v_pk_add_f32 v[70:71], v[72:73], v[70:71]\n");
  } else if (OP == 9) {  // 16 x ds_read_b128, conflict free (a 16-lane group reads one 256-byte row), drained per rept
    RUN16("ds_read_b128 v[40:43], v76\n ds_read_b128 v[44:47], v76 offset:1024\n ds_read_b128 v[48:51], v76 offset:2048\n ds_read_b128 v[52:55], v76 offset:3072\n"
          "ds_read_b128 v[56:59], v76 offset:4096\n ds_read_b128 v[60:63], v76 offset:5120\n ds_read_b128 v[64:67], v76 offset:6144\n ds_read_b128 v[68:71], v76 offset:7168\n"
          "ds_read_b128 v[40:43], v76 offset:8192\n ds_read_b128 v[44:47], v76 offset:9216\n ds_read_b128 v[48:51], v76 offset:10240\n ds_read_b128 v[52:55], v76 offset:11264\n"
          "ds_read_b128 v[56:59], v76 offset:12288\n ds_read_b128 v[60:63], v76 offset:13312\n ds_read_b128 v[64:67], v76 offset:14336\n ds_read_b128 v[68:71], v76 offset:15360\n"
          "s_waitcnt lgkmcnt(8)\n");
  } else if (OP == 10) {  // stage B as shipped: per e-step 1 ds_read_b128 + add_dpp, mov_dpp, 2 pk_fma (4 e-steps = 16 VALU + 4 DS)
    RUN16("v_add_u32_dpp v40, v72, v74" DPPX "ds_read_b128 v[56:59], v76\n" "v_mov_b32_dpp v41, v73" DPPX "v_pk_fma_f32 v[44:45], v[64:65], v[74:75], v[44:45]\n v_pk_fma_f32 v[46:47], v[66:67], v[74:75], v[46:47]\n"
          "v_add_u32_dpp v42, v72, v74" DPPX "ds_read_b128 v[60:63], v76 offset:1024\n" "v_mov_b32_dpp v43, v73" DPPX "v_pk_fma_f32 v[48:49], v[68:69], v[74:75], v[48:49]\n v_pk_fma_f32 v[50:51], v[70:71], v[74:75], v[50:51]\n"
          "v_add_u32_dpp v40, v72, v74" DPPX "ds_read_b128 v[64:67], v76 offset:2048\n" "v_mov_b32_dpp v41, v73" DPPX "v_pk_fma_f32 v[52:53], v[56:57], v[74:75], v[52:53]\n v_pk_fma_f32 v[54:55], v[58:59], v[74:75], v[54:55]\n"
          "v_add_u32_dpp v42, v72, v74" DPPX "ds_read_b128 v[68:71], v76 offset:3072\n" "v_mov_b32_dpp v43, v73" DPPX "v_pk_fma_f32 v[44:45], v[60:61], v[74:75], v[44:45]\n v_pk_fma_f32 v[46:47], v[62:63], v[74:75], v[46:47]\n"
          "s_waitcnt lgkmcnt(2)\n");
  } else if (OP == 11) {  // the same with 8-byte gathers (a 16-bit Q row): 1 ds_read_b64 + add_dpp, mov_dpp, 2 pk_fma
    RUN16("v_add_u32_dpp v40, v72, v74" DPPX "ds_read_b64 v[56:57], v77\n" "v_mov_b32_dpp v41, v73" DPPX "v_pk_fma_f32 v[44:45], v[64:65], v[74:75], v[44:45]\n v_pk_fma_f32 v[46:47], v[66:67], v[74:75], v[46:47]\n"
          "v_add_u32_dpp v42, v72, v74" DPPX "ds_read_b64 v[60:61], v77 offset:1024\n" "v_mov_b32_dpp v43, v73" DPPX "v_pk_fma_f32 v[48:49], v[68:69], v[74:75], v[48:49]\n v_pk_fma_f32 v[50:51], v[70:71], v[74:75], v[50:51]\n"
          "v_add_u32_dpp v40, v72, v74" DPPX "ds_read_b64 v[64:65], v77 offset:2048\n" "v_mov_b32_dpp v41, v73" DPPX "v_pk_fma_f32 v[52:53], v[56:57], v[74:75], v[52:53]\n v_pk_fma_f32 v[54:55], v[58:59], v[74:75], v[54:55]\n"
          "v_add_u32_dpp v42, v72, v74" DPPX "ds_read_b64 v[68:69], v77 offset:3072\n" "v_mov_b32_dpp v43, v73" DPPX "v_pk_fma_f32 v[44:45], v[60:61], v[74:75], v[44:45]\n v_pk_fma_f32 v[46:47], v[62:63], v[74:75], v[46:47]\n"
          "s_waitcnt lgkmcnt(2)\n");
  } else if (OP == 12) {  // 16 x ds_read_b64, conflict free
    RUN16("ds_read_b64 v[40:41], v77\n ds_read_b64 v[44:45], v77 offset:1024\n ds_read_b64 v[48:49], v77 offset:2048\n ds_read_b64 v[52:53], v77 offset:3072\n"
          "ds_read_b64 v[56:57], v77 offset:4096\n ds_read_b64 v[60:61], v77 offset:5120\n ds_read_b64 v[64:65], v77 offset:6144\n ds_read_b64 v[68:69], v77 offset:7168\n"
          "ds_read_b64 v[40:41], v77 offset:8192\n ds_read_b64 v[44:45], v77 offset:9216\n ds_read_b64 v[48:49], v77 offset:10240\n ds_read_b64 v[52:53], v77 offset:11264\n"
          "ds_read_b64 v[56:57], v77 offset:12288\n ds_read_b64 v[60:61], v77 offset:13312\n ds_read_b64 v[64:65], v77 offset:14336\n ds_read_b64 v[68:69], v77 offset:15360\n"
          "s_waitcnt lgkmcnt(8)\n");
  } else {  // 8: 16 independent v_add_u32 (plain VOP2, the cheapest thing there is)
    RUN16("v_add_u32 v40, v72, v40\n v_add_u32 v41, v72, v41\n v_add_u32 v42, v72, v42\n v_add_u32 v43, v72, v43\n"
          "v_add_u32 v44, v72, v44\n v_add_u32 v45, v72, v45\n v_add_u32 v46, v72, v46\n v_add_u32 v47, v72, v47\n"
          "v_add_u32 v48, v72, v48\n v_add_u32 v49, v72, v49\n v_add_u32 v50, v72, v50\n v_add_u32 v51, v72, v51\n"
          "v_add_u32 v52, v72, v52\n v_add_u32 v53, v72, v53\n v_add_u32 v54, v72, v54\n v_add_u32 v55, v72, v55\n");
  }
  if ((threadIdx.x & 63) == 0)
    out[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, unsigned long long* d_out, int cus) {
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps, waves = cus * 4 * wps;
    hipLaunchKernelGGL(probe<OP>, dim3(cus), dim3(threads), 0, 0, d_out);  // warm
    hipLaunchKernelGGL(probe<OP>, dim3(cus), dim3(threads), 0, 0, d_out);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), d_out, waves * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[waves / 2], instr = 16.0 * REPS;
    printf("%-44s %d wave(s)/SIMD: wave time %8.0f cyc -> %5.2f cyc per instruction per SIMD\n", name, wps, med,
           med / (instr * wps));
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  unsigned long long* d_out;
  hipMalloc(&d_out, (size_t)cus * 16 * 8);
  printf("%s, %d CUs; %d x 16 instructions per wave between two s_memtime reads\n", p.name, cus, REPS);
  run<8>("v_add_u32 (plain VOP2)", d_out, cus);
  run<0>("v_fma_f32", d_out, cus);
  run<1>("v_pk_fma_f32", d_out, cus);
  run<7>("v_pk_add_f32", d_out, cus);
  run<2>("v_mov_b32_dpp row_newbcast", d_out, cus);
  run<3>("v_add_u32_dpp row_newbcast", d_out, cus);
  run<4>("v_fmac_f32_dpp row_newbcast", d_out, cus);
  run<5>("e-step as shipped (add_dpp, mov_dpp, 2 pk_fma)", d_out, cus);
  run<6>("e-step proposed (add_dpp, 4 fmac_dpp)", d_out, cus);
  printf("-- LDS (per 16 'instructions' of a rept: see the source; cycles per rept-instruction per SIMD)\n");
  run<9>("16 x ds_read_b128 alone", d_out, cus);
  run<12>("16 x ds_read_b64 alone", d_out, cus);
  run<10>("4 x (ds_read_b128 + 4 VALU e-step): 20 instr/16", d_out, cus);
  run<11>("4 x (ds_read_b64  + 4 VALU e-step): 20 instr/16", d_out, cus);
  return 0;
}
