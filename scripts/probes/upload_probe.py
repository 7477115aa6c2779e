#!/usr/bin/env python
"""Host -> device upload of a big CSR array: pageable `.to(device)` against the pinned, threaded
pipeline of HipBackend.to_device (with and without a dtype conversion on the way)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from muon_amd._backend import HipBackend

be = HipBackend(0)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 4) * (1 << 28)  # GiB of float32
a = np.random.default_rng(0).random(n, dtype=np.float32)
idx = (np.arange(n, dtype=np.int64) % 200000)


def timed(f, reps=2):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) / reps


ref, t = timed(lambda: torch.as_tensor(a).to(be.device))
print(f"pageable .to(device), {a.nbytes / 2**30:.0f} GiB f32: {t:.3f} s  {a.nbytes / t / 1e9:.1f} GB/s", flush=True)
for th in (1, 2, 4, 8):
    be._UPLOAD_THREADS = th
    be.__dict__.pop("_upload_state", None)
    got, t = timed(lambda: be.to_device(a))
    print(f"pinned pipeline, {th} threads: {t:.3f} s  {a.nbytes / t / 1e9:.1f} GB/s  equal={bool(torch.equal(got, ref))}", flush=True)
be._UPLOAD_THREADS = 4
be.__dict__.pop("_upload_state", None)
ref2, t = timed(lambda: torch.as_tensor(idx.astype(np.int32)).to(be.device))
print(f"int64 -> int32 astype + pageable upload: {t:.3f} s", flush=True)
got2, t = timed(lambda: be.to_device(idx, np.int32))
print(f"int64 -> int32 inside the pipeline:      {t:.3f} s  equal={bool(torch.equal(got2, ref2))}", flush=True)
