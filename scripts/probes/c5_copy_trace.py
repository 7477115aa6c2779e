"""Where the device-to-device copies of a segmented MOFA iteration come from (c5_rank8): torch profiler with stacks on ONE
eager iteration of the rank-of-eight emulation."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

from muon_amd._backend import HipBackend
from muon_amd._core.mofa_engine import MofaEngine

spec = importlib.util.spec_from_file_location("bench_rank8", os.path.join(ROOT, "scripts", "bench_rank8.py"))
r8 = importlib.util.module_from_spec(spec)
spec.loader.exec_module(r8)
spec = importlib.util.spec_from_file_location("bench_mofa", os.path.join(ROOT, "scripts", "bench_mofa.py"))
bm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bm)
be = HipBackend(0)
comm = r8.EightAlike()
rna, atac = bm.make_views(be, 0, 12_500, 100_000, 20_000, 100_000, 0, comm)
eng = MofaEngine(be, [rna, atac], np.zeros(12_500, dtype=np.int64), 10, dtype=torch.float32, seed=1, comm=comm, row_offset=0,
                 n_total=100_000)
eng._seg_ok = False  # stay eager: the profiler sees python stacks
eng._graph_ok = False
for _ in range(3):
    eng.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng._iteration_segments() if hasattr(eng, "_iteration_segments") and getattr(eng, "_seg", False) else eng.step()
    torch.cuda.synchronize()
from collections import Counter

c = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::mul_", "aten::mul"):
        st = [f for f in (ev.stack or []) if "muon_amd" in f or "bench_rank8" in f]
        c[(ev.name, tuple(s.split("/")[-1] for s in st[:2]))] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(v, k)
print("kernels:", Counter(ev.name[:50] for ev in prof.events() if ev.device_type is not None and str(ev.device_type).endswith("CUDA")).most_common(12))
