"""The hard-spectrum workload of bench.py (80 planted topics, n_comps = 50, 1e6 x 200 000) with the f64 continuation
(lsi_device(refine_f64=True)): what it costs at that scale and what it certifies.
usage: python scripts/probes/hard_f64_probe.py [cells]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from muon_amd._atac.preproc import tfidf_device  # noqa: E402
from muon_amd._atac.tools import lsi_device  # noqa: E402
from muon_amd._backend import HipBackend  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, topics, k = 200_000, 80, 50
be = HipBackend(0)
X = be.synth_counts(0, n, d, topics, 0.03, 0)
out = torch.empty_like(X.values)
for refine in (False, True, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T = tfidf_device(be, X, n, 3, 1e4, out=out)
    U, sd, V, info = lsi_device(be, T, n_comps=k, n_obs=n, return_info=True, refine_f64=refine)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    r = info["refine_f64"]
    print(f"refine_f64={refine}: {ms:.0f} ms, {info['spmm']} products, converged {info['converged']}, angle bound "
          f"{info['angle_bound']:.2e}, gap {info['gap_rel']:.1e}, f32 floor {info['f32_floor']:.1e}"
          + (f"; continuation: {r['blocks']} blocks, history " + ", ".join(f"{h['angle_bound']:.1e}" for h in r["history"]) if r else ""),
          flush=True)
    del U, V, T  # (the caching allocator keeps the blocks: re-acquiring ~15 GB from the driver took 3 s a run)
print(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
