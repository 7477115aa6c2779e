#!/usr/bin/env python
"""What a 6.3 GB host array costs besides its bytes on the GPU box: first touch (page faults) and release (munmap)."""
import time
import numpy as np

n = int(1.57e9)
src = np.ones(1 << 24, dtype=np.float32)
for rnd in range(3):
    t = time.perf_counter()
    a = np.empty(n, dtype=np.float32)
    t1 = time.perf_counter()
    for i in range(0, n, 1 << 24):
        a[i:i + (1 << 24)] = src[:min(1 << 24, n - i)]
    t2 = time.perf_counter()
    for i in range(0, n, 1 << 24):
        a[i:i + (1 << 24)] = src[:min(1 << 24, n - i)]
    t3 = time.perf_counter()
    b = a.copy()
    t4 = time.perf_counter()
    del a
    t5 = time.perf_counter()
    del b
    t6 = time.perf_counter()
    print(f"alloc {1e3 * (t1 - t):.0f}  first write {1e3 * (t2 - t1):.0f}  second write {1e3 * (t3 - t2):.0f}  copy {1e3 * (t4 - t3):.0f}  "
          f"free {1e3 * (t5 - t4):.0f}  free of the copy {1e3 * (t6 - t5):.0f} ms", flush=True)
with open("/proc/meminfo") as f:
    print("".join(l for l in f if "Huge" in l or "MemFree" in l or "MemAvailable" in l))
