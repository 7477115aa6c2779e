#!/usr/bin/env python
"""Host-side throughput of the fingerprint hash and of plain copies against the number of threads (what bounds the
API path's fingerprints)."""
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import xxhash

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
a = np.random.default_rng(0).integers(0, 255, 6 << 30, dtype=np.uint8)
CH = 64 << 20
for T in (1, 4, 8, 16, 32, 64):
    cuts = list(range(0, a.size, CH))
    with ThreadPoolExecutor(T) as ex:
        list(ex.map(lambda o: xxhash.xxh3_64_intdigest(a[o:o + CH]), cuts[:T]))
        t = time.perf_counter()
        list(ex.map(lambda o: xxhash.xxh3_64_intdigest(a[o:o + CH]), cuts))
        dt = time.perf_counter() - t
        b = np.empty_like(a)
        t = time.perf_counter()
        list(ex.map(lambda o: np.copyto(b[o:o + CH], a[o:o + CH]), cuts))
        dc = time.perf_counter() - t
        t = time.perf_counter()
        list(ex.map(lambda o: np.copyto(b[o:o + CH], a[o:o + CH]), cuts))
        dc2 = time.perf_counter() - t
        t = time.perf_counter()
        list(ex.map(lambda o: a[o:o + CH].sum(dtype=np.uint64), cuts))
        ds = time.perf_counter() - t
        del b
    print(f"{T:3d} threads: xxh3 {a.size / dt / 1e9:6.1f} GB/s, copy (first touch) {a.size / dc / 1e9:6.1f}, copy again {a.size / dc2 / 1e9:6.1f}, byte sum {a.size / ds / 1e9:6.1f} GB/s", flush=True)
t = time.perf_counter()
del a
print(f"freeing 6 GiB: {1e3 * (time.perf_counter() - t):.0f} ms")
