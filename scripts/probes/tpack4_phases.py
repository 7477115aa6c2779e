import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import torch
from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend
be = HipBackend(0)
cells = 1000000
X = be.synth_counts(0, cells, 200000, 50, 0.03, 0)
out = torch.empty_like(X.values)
T = tfidf_device(be, X, cells, 3, 1e4, out=out)
xs, rd = be._xstream_of(T)
NAMES = ["top wait + header", "phase 1 (bits)", "B1 + scan + phase 2", "B2 + phase 3 + next loads", "B3 + write-out"]
for abl in (0, 2):
    be.tune("tpack4_abl", abl)
    be.tune("tpack_dbg", 1)
    be.lib.mu_tpack4_phase_cycles(None, 1)
    r = be.transpose_stream(T, src=(xs, rd)); del r
    o = (ctypes.c_ulonglong * 6)()
    be.lib.mu_tpack4_phase_cycles(ctypes.cast(o, ctypes.c_void_p), 0)
    be.tune("tpack_dbg", 0)
    tot = sum(o[:5])
    tiles = 1954 * (200000 // 480 + 1)
    print(f"abl {abl}: " + ", ".join(f"{n} {100.0 * v / tot:.1f} % ({v / tiles / 100.0:.0f} x100 cyc/tile)" for n, v in zip(NAMES, o[:5])) + f"; retried {o[5]}", flush=True)
be.tune("tpack4_abl", 0)
