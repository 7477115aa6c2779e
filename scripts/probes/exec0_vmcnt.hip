// Does a VMEM load issued with EXEC == 0 take part in vmcnt accounting, in order?
// Wave: cold load L0 -> v1 (HBM miss), then N loads with EXEC = 0, then s_waitcnt vmcnt(N), then
// read v1.  If the EXEC=0 loads are counted and retire in order behind L0, v1 holds the loaded
// value; if they are skipped (not counted) or retire early, the wait falls through and v1 still
// holds the sentinel.  Control arm: same thing with vmcnt(N) but NO extra loads (must be stale
// most of the time, proving the test can see staleness).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ARM>
__global__ void probe(const unsigned* __restrict__ src, unsigned* __restrict__ out, size_t stride) {
  const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned* p = src + w * stride + (threadIdx.x & 63);
  unsigned got;
  unsigned long long save;
  if (ARM == 0) {  // 8 loads with EXEC = 0 behind the real one
    asm volatile(
        "v_mov_b32 %0, 0xdeadbeef\n\t"
        "global_load_dword %0, %2, off\n\t"
        "s_mov_b64 %1, exec\n\t"
        "s_mov_b64 exec, 0\n\t"
        "global_load_dword v200, %2, off\n\t"
        "global_load_dword v200, %2, off\n\t"
        "global_load_dword v200, %2, off\n\t"
        "global_load_dword v200, %2, off\n\t"
        "global_load_dword v200, %2, off\n\t"
        "global_load_dword v200, %2, off\n\t"
        "global_load_dword v200, %2, off\n\t"
        "global_load_dword v200, %2, off\n\t"
        "s_mov_b64 exec, %1\n\t"
        "s_waitcnt vmcnt(8)\n\t"
        "v_mov_b32 %0, %0\n\t"
        "s_nop 4"
        : "=&v"(got), "=&s"(save)
        : "v"(p)
        : "v200", "memory");
  } else {  // control: nothing behind it, same wait
    asm volatile(
        "v_mov_b32 %0, 0xdeadbeef\n\t"
        "global_load_dword %0, %2, off\n\t"
        "s_mov_b64 %1, exec\n\t"
        "s_mov_b64 exec, %1\n\t"
        "s_waitcnt vmcnt(8)\n\t"
        "v_mov_b32 %0, %0\n\t"
        "s_nop 4"
        : "=&v"(got), "=&s"(save)
        : "v"(p)
        : "v200", "memory");
  }
  // copy out BEFORE the load can land late: read `got` into another register right away
  unsigned snap;
  asm volatile("v_mov_b32 %0, %1" : "=&v"(snap) : "v"(got));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[w * 64 + (threadIdx.x & 63)] = snap;
}

int main() {
  const int blocks = 2048, threads = 256, waves = blocks * threads / 64;
  const size_t stride = 1 << 16;  // 256 KiB apart: every wave misses everything
  unsigned *src, *out;
  hipMalloc(&src, waves * stride * 4);
  hipMalloc(&out, waves * 64 * 4);
  hipMemset(src, 0x5a, waves * stride * 4);
  std::vector<unsigned> h(waves * 64);
  for (int arm = 0; arm < 2; ++arm) {
    hipMemset(out, 0, waves * 64 * 4);
    if (arm == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(threads), 0, 0, src, out, stride);
    else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(threads), 0, 0, src, out, stride);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    size_t ok = 0, stale = 0, other = 0;
    for (unsigned v : h) { if (v == 0x5a5a5a5au) ++ok; else if (v == 0xdeadbeefu) ++stale; else ++other; }
    printf("arm %d (%s): loaded=%zu stale=%zu other=%zu\n", arm,
           arm == 0 ? "8 EXEC=0 loads + vmcnt(8)" : "control: no extra loads + vmcnt(8)", ok, stale, other);
  }
  return 0;
}
