#!/usr/bin/env python
"""What would the packed SpMM gain from length-sorted rows dealt round-robin to workgroups / waves?
Permutes the rows of each operand on the device (no kernel change) and times the same kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muon_amd._atac.preproc import tfidf_device
from muon_amd._backend import DeviceCSR, HipBackend

be = HipBackend(0)
N, D = 125000, 200000
X = be.synth_counts(0, N, D, 50, 0.03, 0)
T = tfidf_device(be, X, N, 3, 1e4)
Tt = be.transpose(T)


def sorted_layout(M, K):
    n = M.shape[0]
    lens = M.indptr[1:] - M.indptr[:-1]
    sidx = torch.argsort(lens, descending=True, stable=True)
    nwg = (n + 64 * K - 1) // (64 * K)
    npos = nwg * 64 * K
    i = torch.arange(n, device=lens.device)
    q, j = i // 4, i % 4
    b, t = q % nwg, q // nwg
    w, k = t % 16, t // 16
    pos = ((b * 16 + w) * K + k) * 4 + j
    newlens = torch.zeros(npos, dtype=torch.int64, device=lens.device)
    newlens[pos] = lens[sidx]
    src_start = torch.zeros(npos, dtype=torch.int64, device=lens.device)
    src_start[pos] = M.indptr[:-1][sidx]
    new_indptr = torch.zeros(npos + 1, dtype=torch.int64, device=lens.device)
    new_indptr[1:] = torch.cumsum(newlens, 0)
    idx = torch.repeat_interleave(src_start - new_indptr[:-1], newlens) + torch.arange(M.nnz, device=lens.device)
    return DeviceCSR(new_indptr, M.indices[idx], M.values[idx], (npos, M.shape[1]))


def timeit(P, Dn, reps=4):
    be.spmm(P, Dn); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): be.spmm(P, Dn)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def block_sorted_layout(M, K):
    """Sort only inside every workgroup-sized block of 64 K consecutive rows (keeps neighbouring rows in
    neighbouring memory), deal the block's sorted row-sets round robin to its 16 waves."""
    n = M.shape[0]
    per = 64 * K
    nwg = (n + per - 1) // per
    npos = nwg * per
    lens = torch.zeros(npos, dtype=torch.int64, device=M.indptr.device)
    lens[:n] = M.indptr[1:] - M.indptr[:-1]
    starts = torch.zeros(npos, dtype=torch.int64, device=lens.device)
    starts[:n] = M.indptr[:-1]
    order = torch.argsort(lens.view(nwg, per), dim=1, descending=True, stable=True)  # within block
    i = torch.arange(per, device=lens.device)
    q, j = i // 4, i % 4
    w, k = q % 16, q // 16
    pos_in = ((w * K) + k) * 4 + j                     # sorted rank i -> position inside the block
    src = (order + torch.arange(nwg, device=lens.device)[:, None] * per)  # [nwg, per] source rows by rank
    dst = (pos_in[None, :] + torch.arange(nwg, device=lens.device)[:, None] * per)
    newlens = torch.zeros(npos, dtype=torch.int64, device=lens.device)
    newlens[dst.reshape(-1)] = lens[src.reshape(-1)]
    src_start = torch.zeros(npos, dtype=torch.int64, device=lens.device)
    src_start[dst.reshape(-1)] = starts[src.reshape(-1)]
    new_indptr = torch.zeros(npos + 1, dtype=torch.int64, device=lens.device)
    new_indptr[1:] = torch.cumsum(newlens, 0)
    idx = torch.repeat_interleave(src_start - new_indptr[:-1], newlens) + torch.arange(M.nnz, device=lens.device)
    return DeviceCSR(new_indptr, M.indices[idx], M.values[idx], (npos, M.shape[1]))


for name, M, K in (("X*Q ", T, 8), ("Xt*Y", Tt, 7)):
    Dn = be.randn(M.shape[1], 64, 1)
    be.tune("spmm_k", K)
    base = timeit(be.pack(M, sort_rows=False), Dn)
    S = sorted_layout(M, K)
    srt = timeit(be.pack(S, sort_rows=False), Dn)
    del S
    S = block_sorted_layout(M, K)
    bsr = timeit(be.pack(S, sort_rows=False), Dn)
    print(f"{name}: natural {base:.3f} ms, sorted+dealt {srt:.3f} ms ({100 * (srt / base - 1):+.1f} %), "
          f"sorted inside workgroup blocks {bsr:.3f} ms ({100 * (bsr / base - 1):+.1f} %)", flush=True)
    del S
be.tune("spmm_k", 0)
