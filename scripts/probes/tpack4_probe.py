#!/usr/bin/env python
"""r05 operand-building probe on the bench matrix: the TF-IDF scale sweep with / without the row stream, the
transposition's third generation (alone and next to the streaming copy, as lsi ran it in r04) against the fourth
(csrc/tpack4.hip: CSR source, row-stream source), their output bytes compared, the fourth generation's phase cycles
and a sweep of its tile-width parameter.  Usage: tpack4_probe.py [cells] [--sweep]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import HipBackend

be = HipBackend(0)
be.keep_tpack4_work = True
cells = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 125000
peaks = 200000
X = be.synth_counts(0, cells, peaks, 50, 0.03, 0)
print(f"{cells} x {peaks}, {X.nnz} stored entries", flush=True)
out = torch.empty_like(X.values)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = None
    for _ in range(reps):
        r = None  # (at 1e6 cells a result is 50 .. 100 GB: never two alive)
        r = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, r


def checksum(t, n):
    """two 64-bit sums over the first n pairs (wrapping): the plain one and one weighted by position"""
    v = t[:n].view(torch.int64)
    s1, s2 = 0, 0
    step = 1 << 27
    for lo in range(0, n, step):
        c = v[lo:lo + step]
        s1 = (s1 + int(c.sum().item())) & ((1 << 64) - 1)
        w = torch.arange(lo, lo + c.numel(), device=c.device, dtype=torch.int64)
        s2 = (s2 + int((c * (2 * w + 1)).sum().item())) & ((1 << 64) - 1)
    return s1, s2


def scale_only(emit):
    """the scale sweep alone (sums and idf once outside)"""
    rowsum, colsum = be.row_col_sums(X)
    idf = be.idf(colsum, float(cells), 3, X.values.dtype)
    lay = be.stream_layout(X) if emit else None
    keep = be.__dict__.pop("_sweep_work", None)

    def run():
        be._sweep_work = keep
        return be.tfidf_scale(X, rowsum, idf, 1e4, 3, out=out, emit=lay)

    ms, _ = timed(run)
    return ms


print(f"scale sweep, values only:        {scale_only(False):.2f} ms", flush=True)
print(f"scale sweep, values + row stream: {scale_only(True):.2f} ms", flush=True)
torch.cuda.empty_cache()
ms0, T0 = timed(lambda: tfidf_device(be, X, cells, 3, 1e4, out=out, emit_stream=False))
print(f"tfidf_device without the stream (sums + idf + scale):          {ms0:.2f} ms", flush=True)
nnz = T0.nnz

# the reference: streaming copy + the general transposition (r06: the third generation is archived in scripts/probes/tpack_v3.hip)
be.tune("tpack4_off", 1)
ms3, (Xs3, Xt3) = timed(lambda: be.stream_both(T0))
print(f"general: stream_both (copy of X on a second stream + count + layout + fill): {ms3:.2f} ms", flush=True)
ref_t, ref_x = checksum(Xt3.ent, nnz), checksum(Xs3.ent, nnz)
ref_sptr, ref_perm = Xt3.sptr.clone(), Xt3.perm.clone()
del Xs3, Xt3
torch.cuda.empty_cache()
ms3a, _r = timed(lambda: be.transpose_stream(T0))
del _r
print(f"general: transposition alone:                                                {ms3a:.2f} ms", flush=True)
be.tune("tpack4_off", 0)
torch.cuda.empty_cache()

ms4c, Xt4 = timed(lambda: be.transpose_stream(T0))
print(f"v4: transposition from the CSR arrays:    {ms4c:.2f} ms  (error word {be.tpack4_status()})", flush=True)
same = checksum(Xt4.ent, nnz) == ref_t and torch.equal(Xt4.sptr, ref_sptr) and torch.equal(Xt4.perm, ref_perm)
print(f"    same bytes as general: {same}", flush=True)
del Xt4, T0
torch.cuda.empty_cache()

ms, T = timed(lambda: tfidf_device(be, X, cells, 3, 1e4, out=out, emit_stream=True))
print(f"tfidf_device with the stream (layout + sums + idf + scale):    {ms:.2f} ms", flush=True)
assert be._xstream_of(T) is not None
ms4s, (Xs4, Xt4) = timed(lambda: be.stream_both(T))
print(f"v4: stream_both with the sweep's stream:   {ms4s:.2f} ms  (error word {be.tpack4_status()})", flush=True)
same = checksum(Xt4.ent, nnz) == ref_t and torch.equal(Xt4.sptr, ref_sptr) and torch.equal(Xt4.perm, ref_perm)
print(f"    X^T stream same bytes as general: {same};  X stream same bytes as the copy: {checksum(Xs4.ent, nnz) == ref_x}", flush=True)
del Xt4, Xs4
torch.cuda.empty_cache()
# count phase alone
col_nnz = be.empty((peaks,), torch.int64)
wb = int(be.lib.mu_tpack4_worksize(cells, peaks, T.nnz))
work = be.empty((wb,), torch.uint8)
sp = be._slab_ptr_of(T)
msc, _ = timed(lambda: be.lib.mu_tpack4_count(cells, peaks, T.nnz, T.indptr.data_ptr(), T.indices.data_ptr(),
                                               col_nnz.data_ptr(), work.data_ptr(), wb, sp.data_ptr() if sp is not None else None,
                                               be._stream()))
print(f"v4: count + base alone: {msc:.2f} ms", flush=True)
del work

NAMES = ["top wait + header", "phase 1 (bits)", "B1 + phase 2 (prefix)", "B2 + phase 3 (place) + next loads", "B3 + write-out", "retried tiles"]


def phases(label):
    be.tune("tpack_dbg", 1)
    be.lib.mu_tpack4_phase_cycles(None, 1)
    be.stream_both(T)
    o = (ctypes.c_ulonglong * 6)()
    be.lib.mu_tpack4_phase_cycles(ctypes.cast(o, ctypes.c_void_p), 0)
    be.tune("tpack_dbg", 0)
    tot = sum(o[:5])
    print(f"   {label}: " + ", ".join(f"{n} {100.0 * v / max(tot, 1):.1f} %" for n, v in zip(NAMES[:5], o[:5])) +
          f", retried tiles {o[5]}", flush=True)


phases("phase share of the fill (thread 0 of every block)")
be.tune("tpack4_plain", 1)
ms, r_ = timed(lambda: be.stream_both(T))
del r_
print(f"v4 with workgroup = row block (no XCD-aware order): {ms:.2f} ms", flush=True)
phases("plain order")
be.tune("tpack4_plain", 0)
if "--sweep" in sys.argv:
    for m in (14, 15, 17):
        be.tune("tpack4_m", m)
        ms, r_ = timed(lambda: be.stream_both(T))
        del r_
        g = (ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_int(0))
        be.lib.mu_tpack4_geometry(cells, peaks, T.nnz, ctypes.byref(g[0]), ctypes.byref(g[1]), ctypes.byref(g[2]))
        print(f"v4 m = {m} (tile {g[2].value} columns, {g[1].value} blocks of {g[0].value} rows): {ms:.2f} ms", flush=True)
        phases(f"m = {m}")
    be.tune("tpack4_m", 0)
