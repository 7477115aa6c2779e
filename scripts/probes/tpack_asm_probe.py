#!/usr/bin/env python
"""The transposition's fill with the batch loads from asm (counted waits, r04) against the compiler's loads
(tune tpack_asm = 1): time, phase shares, and the same bytes out."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
X = be.synth_counts(0, cells, 200000, 50, 0.03, 0)
T = tfidf_device(be, X, cells, 3, 1e4)
NAMES = ["header+scan", "count walk", "prefix", "place walk", "wait others", "write-out"]
ref = None
for mode, split in ((1, 1), (0, 1), (0, 0), (1, 1), (0, 1), (0, 0)):
    be.tune("tpack_asm", mode)
    be.tune("tcount_pipe", split)
    P = be.transpose_stream(T)
    torch.cuda.synchronize()
    if ref is None:
        ref = P
    same = bool(torch.equal(P.ent, ref.ent) and torch.equal(P.sptr, ref.sptr))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        be.transpose_stream(T)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 3
    be.tune("tpack_dbg", 1)
    be.lib.mu_csr_tpack_phase_cycles(None, 1)
    be.transpose_stream(T)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 6)()
    be.lib.mu_csr_tpack_phase_cycles(ctypes.cast(out, ctypes.c_void_p), 0)
    be.tune("tpack_dbg", 0)
    tot = sum(out)
    print(f"count sweep {'r03' if split else 'pipelined'}, tpack_asm {mode} ({'compiler loads' if mode else 'asm loads, counted waits'}): {ms:.2f} ms (count + layout + scan + fill), "
          f"same bytes {same}; " + ", ".join(f"{n} {100.0 * v / tot:.1f} %" for n, v in zip(NAMES, out)), flush=True)
be.tune("tpack_asm", 0)
be.tune("tcount_pipe", 0)
