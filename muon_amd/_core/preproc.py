"""muon.pp.neighbors (weighted nearest neighbours), muon.pp.l2norm - SURVEY 8f.4.

Host side mirrors /root/reference/muon/_core/preproc.py:182-260 (``l2norm``) and :264-640
(``neighbors``: same signature, same slots - ``mdata.obsp[distances / connectivities]``,
``mdata.uns[key]``, per-modality weights in ``mdata.obs["<mod>:mod_weight"]`` or the modalities' own
``.obs``).  The reference searches neighbours with UMAP's NN-descent (approximate, seeded; numba metric
kernels ``_jaccard_euclidean_metric`` :53-77, ``_sparse_csr_fast_knn_`` :112-134); here every search is
EXHAUSTIVE on the device - distance tiles as GEMMs against the whole representation, top-k per tile -
so ``random_state`` / ``low_memory`` are accepted and have nothing to influence; the result is the exact
graph NN-descent approximates (oracle/wnn_oracle.py is the same computation in numpy loops).

Device work: the within- and cross-modality neighbourhood means ``r_i = mean_{j in N(i)} x_j`` are
products of the kNN graph with the n x p representation and run through the row-stream SpMM of the LSI
(csrc/spmm_win.hip); distance tiles, selections, the shared-neighbour (Jaccard) counts and the fuzzy
simplicial set are PyTorch-ROCm tensor operations on HBM-resident arrays (plumbing: no hand-written
kernel is claimed for them).

``knn`` writes what scanpy's ``sc.pp.neighbors`` would (scanpy is outside muon and absent here): the
per-modality input the reference requires (:366-373).
"""
from __future__ import annotations

import math
from itertools import repeat
from typing import Dict, Optional

import numpy as np
import torch
from scipy.sparse import csr_matrix, issparse

from .._containers import is_anndata, is_mudata

_METRICS = ("euclidean", "sqeuclidean", "cosine", "cityblock", "manhattan", "chebyshev")


def _backend(backend):
    if backend is None:
        from .._backend import get_backend

        backend = get_backend()
    return backend


def _choose_representation(adata, use_rep=None, n_pcs=None):
    """scanpy.tools._utils._choose_representation, the part the path needs (:372)."""
    if use_rep is None or use_rep == -1:
        if adata.X.shape[1] > 50 and "X_pca" in adata.obsm:
            X = adata.obsm["X_pca"]
            return X[:, :n_pcs] if n_pcs not in (None, -1, 0) else X
        return adata.X
    if use_rep == "X":
        return adata.X
    if use_rep in adata.obsm:
        X = adata.obsm[use_rep]
        if use_rep == "X_pca" and n_pcs not in (None, -1, 0):
            X = X[:, :n_pcs]
        return X
    raise ValueError(f"Did not find {use_rep} in `.obsm.keys()`.")


# -----------------------------------------------------------------------------------------------------
# l2norm (reference :182-260)
# -----------------------------------------------------------------------------------------------------
# widest representation the filter kernel's LDS tiles hold (csrc/knn.hip: two 64-row operand tiles of p_pad + 1
# doubles in 160 KiB); wider ones take the tiled search (the gate said 1024 and the kernel raised)
_KNN_FILTER_MAX_P = 156


def _l2norm(adata, rep=None, n_pcs=0):
    X = _choose_representation(adata, rep, n_pcs)
    if issparse(X):
        # in place on X's own arrays, whatever the format (reference :194-195 writes `X.data[:]` for csr, csc and
        # coo; r03 normalised a csr COPY of csc / coo input and left X untouched - ADVICE r03).  Like there, the
        # scaled values are cast to X's dtype on assignment.
        fmt = X.format
        if fmt == "csr":
            rows = np.repeat(np.arange(X.shape[0]), np.diff(X.indptr))
        elif fmt == "csc":
            rows = np.asarray(X.indices)
        elif fmt == "coo":
            rows = np.asarray(X.row)
        else:
            raise TypeError(f"l2norm: sparse format '{fmt}' is not supported (csr, csc, coo)")
        data = np.asarray(X.data, dtype=np.float64)
        nrm = np.sqrt(np.bincount(rows, weights=data * data, minlength=X.shape[0]))
        with np.errstate(divide="ignore", invalid="ignore"):
            d = data / nrm[rows]
        d[~np.isfinite(d)] = 0
        X.data[:] = d
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            norm = X / np.linalg.norm(X, ord=2, axis=1, keepdims=True)
        norm[~np.isfinite(norm)] = 0
        X[:] = norm


def l2norm(mdata, mod=None, rep=None, n_pcs=0, copy: bool = False):
    """Normalize observations to unit L2 norm (reference :205-260, same arguments)."""
    if is_anndata(mdata):
        for name, v in (("rep", rep), ("n_pcs", n_pcs)):
            if v is not None and not isinstance(v, (str, int)):
                v = list(v)
                if len(v) != 1:
                    raise RuntimeError(f"If '{name}' is an Iterable, it must have length 1")
                if name == "rep":
                    rep = v[0]
                else:
                    n_pcs = v[0]
        if copy:
            mdata = mdata.copy()
        _l2norm(mdata, rep, n_pcs)
    else:
        if mod is None:
            mod = mdata.mod.keys()
        elif isinstance(mod, str):
            mod = [mod]
        if rep is None or isinstance(rep, str):
            rep = repeat(rep)
        if n_pcs is None or isinstance(n_pcs, int):
            n_pcs = repeat(n_pcs)
        if copy:
            mdata = mdata.copy()
        for m, r, n in zip(mod, rep, n_pcs):
            _l2norm(mdata.mod[m], r, n)
    return mdata if copy else None


# -----------------------------------------------------------------------------------------------------
# exhaustive k nearest neighbours on the device
# -----------------------------------------------------------------------------------------------------
# the `metric` values of the reference's signature (preproc.py:270-294) that scipy's cdist evaluates pair by pair;
# "mahalanobis" / "seuclidean" take their (co)variances from the rows of each cdist call - in the reference one call per
# cell over that cell's candidates: `_cell_dist`, in the final step of `neighbors` only - and "wminkowski" needs weights
# the signature cannot pass (and is gone from scipy): not offered
_PAIR_METRICS = ("euclidean", "sqeuclidean", "minkowski", "cityblock", "manhattan", "chebyshev", "cosine", "correlation",
                 "braycurtis", "canberra", "jensenshannon", "hamming", "matching", "jaccard", "dice", "kulsinski",
                 "rogerstanimoto", "russellrao", "sokalmichener", "sokalsneath", "yule")


def _pair_dist(A: torch.Tensor, B: torch.Tensor, metric: str) -> torch.Tensor:
    """Row-wise distances d(A_i, B_i) (same shapes [..., p]), computed directly (no expansion), with the definitions
    of scipy.spatial.distance (what the reference's final step calls through cdist, preproc.py:596-606)."""
    if metric in ("euclidean", "sqeuclidean", "minkowski"):  # (minkowski: scipy's default p = 2)
        d = ((A - B) ** 2).sum(dim=-1)
        return d if metric == "sqeuclidean" else torch.sqrt(d)
    if metric in ("cityblock", "manhattan"):
        return (A - B).abs().sum(dim=-1)
    if metric == "chebyshev":
        return (A - B).abs().amax(dim=-1)
    if metric in ("cosine", "correlation"):
        if metric == "correlation":
            A, B = A - A.mean(dim=-1, keepdim=True), B - B.mean(dim=-1, keepdim=True)
        num = (A * B).sum(dim=-1)
        den = torch.sqrt((A * A).sum(dim=-1) * (B * B).sum(dim=-1))
        return 1.0 - num / den
    if metric == "braycurtis":
        return (A - B).abs().sum(dim=-1) / (A + B).abs().sum(dim=-1)
    if metric == "canberra":
        den = A.abs() + B.abs()
        t = (A - B).abs() / torch.where(den > 0, den, torch.ones_like(den))
        return torch.where(den > 0, t, torch.zeros_like(t)).sum(dim=-1)
    if metric == "jensenshannon":
        P, Q = A / A.sum(dim=-1, keepdim=True), B / B.sum(dim=-1, keepdim=True)
        Mm = 0.5 * (P + Q)
        kl = torch.xlogy(P, P / Mm).nan_to_num(0.0).sum(dim=-1) + torch.xlogy(Q, Q / Mm).nan_to_num(0.0).sum(dim=-1)
        return torch.sqrt(torch.clamp(0.5 * kl, min=0.0))
    if metric in ("hamming", "matching"):
        return (A != B).to(A.dtype).mean(dim=-1)
    if metric == "jaccard":  # (on the non-zero patterns, as scipy >= 1.15's cdist; older ones compared real values - the
        a, b = A != 0, B != 0  # two agree on 0/1 data, what the metric is meant for)
        den = (a | b).to(A.dtype).sum(dim=-1)
        num = (a != b).to(A.dtype).sum(dim=-1)
        return torch.where(den > 0, num / torch.where(den > 0, den, torch.ones_like(den)), torch.zeros_like(den))
    if metric in ("dice", "kulsinski", "rogerstanimoto", "russellrao", "sokalmichener", "sokalsneath", "yule"):
        a, b = A != 0, B != 0  # (scipy casts to bool)
        f = A.dtype
        n = float(A.shape[-1])
        ntt = (a & b).to(f).sum(dim=-1)
        ntf = (a & ~b).to(f).sum(dim=-1)
        nft = (~a & b).to(f).sum(dim=-1)
        nff = n - ntt - ntf - nft
        if metric == "dice":
            return (ntf + nft) / (2.0 * ntt + ntf + nft)
        if metric == "kulsinski":
            return (ntf + nft - ntt + n) / (ntf + nft + n)
        if metric == "russellrao":
            return (n - ntt) / n
        if metric == "yule":
            R = 2.0 * ntf * nft
            den = ntt * nff + 0.5 * R
            return torch.where(R > 0, R / torch.where(den > 0, den, torch.ones_like(den)), torch.zeros_like(R))
        R = 2.0 * (ntf + nft)
        if metric == "sokalsneath":
            return R / (ntt + R)
        return R / (ntt + nff + R)  # rogerstanimoto == sokalmichener
    raise NotImplementedError(f"metric '{metric}' (implemented pair by pair: {_PAIR_METRICS}; 'mahalanobis' and 'seuclidean' "
                              "depend on the rows of each cdist call: offered for the final step of `neighbors` only; "
                              "'wminkowski' needs weights)")


def _cell_dist(X: torch.Tensor, ri: torch.Tensor, ci: torch.Tensor, ok: torch.Tensor, present: torch.Tensor,
               metric: str, step: int = 1 << 20) -> torch.Tensor:
    """`seuclidean` / `mahalanobis` distances of the candidate pairs (ri, ci) (sorted by ri), as the reference gets
    them: ONE cdist call per cell, `cdist(rep[None, cell], rep[nz])` (preproc.py:596-606), so scipy takes the
    variances V (ddof = 1) / the covariance of THAT call's rows - the cell and its candidates.  Here: segment
    statistics over the pairs of a cell (two passes: means, then centred sums).  `ok`: both ends present in this
    modality (other pairs get 0 and do not count).  mahalanobis needs more rows than dimensions in every call, like
    scipy (ValueError)."""
    n, p = X.shape
    a, b = ri[ok], ci[ok]
    one = torch.ones(a.numel(), dtype=X.dtype, device=X.device)
    cnt = present.to(X.dtype).clone()  # (the cell's own row)
    cnt.index_add_(0, a, one)
    S1 = X * present[:, None].to(X.dtype)
    for lo in range(0, a.numel(), step):
        S1.index_add_(0, a[lo:lo + step], X[b[lo:lo + step]])
    mean = S1 / cnt.clamp(min=1.0)[:, None]
    out = torch.zeros(ri.numel(), dtype=X.dtype, device=X.device)
    sel = torch.nonzero(ok).reshape(-1)
    if metric == "seuclidean":
        SS = (X - mean) ** 2 * present[:, None].to(X.dtype)
        for lo in range(0, a.numel(), step):
            aa, bb = a[lo:lo + step], b[lo:lo + step]
            SS.index_add_(0, aa, (X[bb] - mean[aa]) ** 2)
        V = SS / (cnt - 1.0).clamp(min=1.0)[:, None]
        for lo in range(0, a.numel(), step):
            aa, bb = a[lo:lo + step], b[lo:lo + step]
            out[sel[lo:lo + step]] = torch.sqrt((((X[aa] - X[bb]) ** 2) / V[aa]).sum(dim=1))
        return out
    if bool(((cnt <= p) & present).any()):
        raise ValueError("The number of observations (m) is too small; the covariance matrix is singular. For observations "
                         f"with {p} dimensions, at least {p + 1} observations are required.")  # (scipy's cdist)
    Xc = (X - mean) * present[:, None].to(X.dtype)
    C = Xc[:, :, None] * Xc[:, None, :]
    sub = max(1, min(step, (1 << 27) // max(p * p, 1)))
    for lo in range(0, a.numel(), sub):
        aa, bb = a[lo:lo + sub], b[lo:lo + sub]
        d = X[bb] - mean[aa]
        C.index_add_(0, aa, d[:, :, None] * d[:, None, :])
    C = C / (cnt - 1.0).clamp(min=1.0)[:, None, None]
    eye = torch.eye(p, dtype=X.dtype, device=X.device)
    C = torch.where(present[:, None, None], C, eye)  # (absent cells: no pairs, any invertible matrix)
    VI = torch.linalg.inv(C)
    for lo in range(0, a.numel(), sub):
        aa, bb = a[lo:lo + sub], b[lo:lo + sub]
        d = X[aa] - X[bb]
        out[sel[lo:lo + sub]] = torch.sqrt(torch.clamp(torch.einsum("np,npq,nq->n", d, VI[aa], d), min=0.0))
    return out


def _candidates_filtered(be, Xn: torch.Tensor, sq: torch.Tensor, kc: int, chunk_elems: int, cap: Optional[int] = None) -> torch.Tensor:
    """kc nearest candidates per row (GEMM-form squared distances) WITHOUT the n x n distance panels: the
    rows are walked as candidates in a random order, in panels of doubling size.  The first panel is searched
    densely and leaves every query its kc-th smallest distance so far as a threshold; for each later panel the
    HIP filter kernel (csrc/knn.hip: distances on the f64 matrix cores, compared in registers) appends the
    candidates that beat the threshold - about kc per query and panel, because a panel doubles what the query
    has seen - and a top-k over [list ++ buffer] (4 kc wide instead of n) updates list and threshold.  A
    buffer that overflowed (ties, duplicated rows) has its rows redone densely."""
    n, p = Xn.shape
    dev = Xn.device
    p_pad = (p + 3) // 4 * 4
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    perm = torch.randperm(n, device=dev, generator=g)  # candidate position -> row
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device=dev)            # row -> candidate position
    Xq = torch.zeros((n, p_pad), dtype=torch.float64, device=dev)
    Xq[:, :p] = Xn
    Xc = Xq[perm].contiguous()
    sq = sq.contiguous()
    sqc = sq[perm].contiguous()
    self_pos = inv.to(torch.int32).contiguous()
    ar = torch.arange(n, device=dev)

    def dense(rows, c0, c1):
        """squared distances of `rows` (tensor of row numbers, or None: all) to positions [c0, c1), self = inf"""
        q = Xq if rows is None else Xq[rows]
        D = (sq if rows is None else sq[rows])[:, None] + sqc[None, c0:c1] - 2.0 * (q @ Xc[c0:c1].T)
        sp = inv if rows is None else inv[rows]
        hit = (sp >= c0) & (sp < c1)
        r = torch.nonzero(hit)[:, 0]
        D[r, sp[r] - c0] = float("inf")
        return D

    p0 = min(n, max(2048, 4 * kc))
    cur_d = torch.empty((n, kc), dtype=torch.float64, device=dev)
    cur_p = torch.empty((n, kc), dtype=torch.int64, device=dev)
    rows = max(1, min(n, chunk_elems // p0))
    for lo in range(0, n, rows):
        hi = min(n, lo + rows)
        t = torch.topk(dense(ar[lo:hi], 0, p0), kc, dim=1, largest=False)
        cur_d[lo:hi], cur_p[lo:hi] = t.values, t.indices
    cap = int(cap) if cap else 3 * kc + 64
    buf_pos = torch.empty((n, cap), dtype=torch.int32, device=dev)
    buf_d = torch.empty((n, cap), dtype=torch.float64, device=dev)
    cnt = torch.empty((n,), dtype=torch.int32, device=dev)
    slot = torch.arange(cap, device=dev)[None, :]
    c_lo = p0
    fused = hasattr(be, "knn_merge") and kc + cap <= 1024
    thr = cur_d.amax(dim=1).contiguous()
    while c_lo < n:
        c_hi = min(n, 2 * c_lo)
        be.knn_filter(Xq, Xc, sq, sqc, thr, self_pos, c_lo, c_hi, buf_pos, buf_d, cnt)
        if fused:  # list ++ buffer -> the kc smallest and the new threshold, a wave per query (csrc/knn.hip)
            new_d, new_p, thr = be.knn_merge(cur_d.contiguous(), cur_p.contiguous(), buf_d, buf_pos, cnt)
        else:
            valid = slot < cnt[:, None]
            d_all = torch.cat([cur_d, torch.where(valid, buf_d, torch.full_like(buf_d, float("inf")))], dim=1)
            p_all = torch.cat([cur_p, buf_pos.long()], dim=1)
            t = torch.topk(d_all, kc, dim=1, largest=False)
            new_d, new_p = t.values, torch.gather(p_all, 1, t.indices)
        over = torch.nonzero(cnt > cap)[:, 0]
        if over.numel():  # (rare) more candidates than the buffer holds: those rows again, densely, over all seen
            for lo in range(0, over.numel(), max(1, chunk_elems // c_hi)):
                r = over[lo:lo + max(1, chunk_elems // c_hi)]
                t = torch.topk(dense(r, 0, c_hi), kc, dim=1, largest=False)
                new_d[r], new_p[r] = t.values, t.indices
            thr = new_d.amax(dim=1).contiguous()
        elif not fused:
            thr = new_d.amax(dim=1).contiguous()
        cur_d, cur_p = new_d, new_p
        c_lo = c_hi
    return perm[cur_p], cur_d


def device_knn(X: torch.Tensor, k: int, metric: str = "euclidean", chunk_elems: int = 1 << 28, backend=None,
               indices_only: bool = False):
    """The k nearest OTHER rows of every row of X [n, p] (f64 on the device): (indices [n, k] int64,
    distances [n, k]) ascending, ties by index.  k + 8 candidates per query from squared distances in GEMM
    form (cosine: normalised rows) - with a backend that has the filter kernel and enough rows through
    ``_candidates_filtered``, else tiles of queries against all rows and a top-k per tile -, then the exact
    distances of the candidates and the final selection: the cancellation of the GEMM form never decides
    the order.  ``indices_only``: the caller wants the neighbour SET (the candidate union of ``neighbors``): with
    the filter path the k smallest GEMM-form distances decide and the exact re-evaluation - a gather of
    n x (k + 8) x p values - is skipped; the distances returned are then the GEMM-form ones."""
    if metric not in _PAIR_METRICS:
        _pair_dist(X[:1], X[:1], metric)  # (raises, naming what is offered)
    n, p = X.shape
    k = min(int(k), n - 1)
    kc = min(k + 8, n - 1)
    idx = torch.empty((n, k), dtype=torch.int64, device=X.device)
    dst = torch.empty((n, k), dtype=X.dtype, device=X.device)
    gemm = metric in ("euclidean", "sqeuclidean", "cosine")
    Xn = X / torch.sqrt((X * X).sum(dim=1, keepdim=True)) if metric == "cosine" else X
    sq = (Xn * Xn).sum(dim=1)
    ar = torch.arange(n, device=X.device)
    cand_all = None
    if (gemm and backend is not None and hasattr(backend, "knn_filter") and X.dtype == torch.float64
            and n >= 8192 and 4 * kc <= n // 2 and -(-p // 4) * 4 <= _KNN_FILTER_MAX_P):
        cand_all, cand_d = _candidates_filtered(backend, Xn, sq, kc, chunk_elems)
        if indices_only:
            t = torch.topk(cand_d, k, dim=1, largest=False, sorted=True)
            d = t.values.clamp_min(0.0)
            return torch.gather(cand_all, 1, t.indices), (d if metric == "sqeuclidean" else
                                                            torch.sqrt(d) if metric == "euclidean" else d / 2.0)
    tiled = gemm or metric in _METRICS  # (else: every pair through `_pair_dist`, a tile of queries x all rows x p)
    rows = max(1, min(n, chunk_elems // max((n if tiled else n * p) if cand_all is None else kc * p, 1)))
    for lo in range(0, n, rows):
        hi = min(n, lo + rows)
        if cand_all is not None:
            cand = cand_all[lo:hi]
        else:
            if gemm:
                D = sq[lo:hi, None] + sq[None, :] - 2.0 * (Xn[lo:hi] @ Xn.T)
            elif tiled:
                D = torch.cdist(X[lo:hi], X, p=1.0 if metric in ("cityblock", "manhattan") else float("inf"))
            else:
                D = _pair_dist(X[lo:hi, None, :].expand(hi - lo, n, p), X[None, :, :].expand(hi - lo, n, p), metric)
                D = torch.where(torch.isnan(D), torch.full_like(D, float("inf")), D)
            D[ar[lo:hi] - lo, ar[lo:hi]] = float("inf")  # not the row itself
            cand = torch.topk(D, kc, dim=1, largest=False).indices
            del D
        exact = _pair_dist(X[lo:hi, None, :].expand(hi - lo, kc, p), X[cand], metric)
        # ascending by (distance, index): stable sort of the index-sorted candidates
        o = torch.argsort(cand, dim=1)
        cand, exact = torch.gather(cand, 1, o), torch.gather(exact, 1, o)
        o = torch.argsort(exact, dim=1, stable=True)[:, :k]
        idx[lo:hi], dst[lo:hi] = torch.gather(cand, 1, o), torch.gather(exact, 1, o)
    return idx, dst


# -----------------------------------------------------------------------------------------------------
# UMAP connectivities (scanpy's `umap` connectivity: fuzzy_simplicial_set with set_op_mix_ratio = 1,
# local_connectivity = 1; umap/umap_.py smooth_knn_dist + compute_membership_strengths)
# -----------------------------------------------------------------------------------------------------
def fuzzy_simplicial_set(knn_idx: torch.Tensor, knn_dist: torch.Tensor, n_obs: int, n_neighbors: int, backend=None):
    d = knn_dist.to(torch.float32).to(torch.float64)  # umap works on float32 distances
    n, k = d.shape
    target = math.log2(n_neighbors)
    if backend is not None and hasattr(backend, "umap_strengths") and n > 0:
        # rho, the 64 bisection steps for sigma and the strengths: a thread per row (csrc/wnn.hip)
        val = backend.umap_strengths(d.contiguous(), knn_idx.to(torch.int64).contiguous(), target, float(d.mean()))
        return _symmetrise(knn_idx, val, n_obs)
    pos = torch.where(d > 0, d, torch.full_like(d, float("inf")))
    has = torch.isfinite(pos).any(dim=1)
    # rho: the first positive distance in storage order (local_connectivity = 1)
    first = torch.argmax((d > 0).to(torch.int8), dim=1)
    rho = torch.where(has, torch.gather(d, 1, first[:, None]).squeeze(1), torch.zeros(n, dtype=d.dtype, device=d.device))
    lo = torch.zeros(n, dtype=d.dtype, device=d.device)
    hi = torch.full((n,), float("inf"), dtype=d.dtype, device=d.device)
    mid = torch.ones(n, dtype=d.dtype, device=d.device)
    done = torch.zeros(n, dtype=torch.bool, device=d.device)
    x = d[:, 1:] - rho[:, None]  # (the first stored neighbour is skipped, as in umap)
    for _ in range(64):
        ps = torch.where(x > 0, torch.exp(-x / mid[:, None]), torch.ones_like(x)).sum(dim=1)
        done = done | ((ps - target).abs() < 1e-5)
        up = ps > target
        new_hi = torch.where(up & ~done, mid, hi)
        new_lo = torch.where(~up & ~done, mid, lo)
        new_mid = torch.where(up, (lo + mid) / 2.0, torch.where(torch.isinf(hi), mid * 2.0, (mid + hi) / 2.0))
        mid = torch.where(done, mid, new_mid)
        lo, hi = new_lo, new_hi
        if bool(done.all()):
            break
    sigma = mid
    mean_i = d.mean(dim=1)
    mean_all = d.mean()
    sigma = torch.where(rho > 0, torch.maximum(sigma, 1e-3 * mean_i), torch.maximum(sigma, 1e-3 * mean_all))
    rows = torch.arange(n, device=d.device)[:, None].expand(n, k)
    val = torch.where(knn_idx == rows, torch.zeros_like(d),
                      torch.where((d - rho[:, None] <= 0) | (sigma[:, None] == 0), torch.ones_like(d),
                                  torch.exp(-(d - rho[:, None]) / sigma[:, None])))
    return _symmetrise(knn_idx, val, n_obs)


def _symmetrise(knn_idx: torch.Tensor, val: torch.Tensor, n_obs: int):
    """P + P^T - P o P^T of the membership strengths (set_op_mix_ratio = 1), as a CSR for .obsp.  On the tensors'
    device (r04; was scipy on the host: transpose, multiply, add, subtract = 86 ms per graph of 100 000 x 20): the
    entries of P and P^T as keys row * n + column, one `unique`, and per key the sum s1 and the sum of squares s2
    of its one or two values - a + b - a b = s1 - (s1^2 - s2) / 2, and a single value a gives a exactly."""
    n, k = val.shape
    dev = val.device
    r = torch.arange(n, device=dev, dtype=torch.int64).repeat_interleave(k)
    c = knn_idx.reshape(-1).to(torch.int64)
    v = val.reshape(-1).to(torch.float64)
    keep = v != 0
    r, c, v = r[keep], c[keep], v[keep]
    key = torch.cat([r * n_obs + c, c * n_obs + r])
    vv = torch.cat([v, v])
    uk, inv = torch.unique(key, return_inverse=True)  # sorted: row-major, columns ascending
    s1 = torch.zeros(uk.numel(), dtype=torch.float64, device=dev).scatter_add_(0, inv, vv)
    s2 = torch.zeros(uk.numel(), dtype=torch.float64, device=dev).scatter_add_(0, inv, vv * vv)
    out = s1 - (s1 * s1 - s2) / 2.0
    nz = out != 0
    uk, out = uk[nz], out[nz]
    rows = torch.div(uk, n_obs, rounding_mode="floor")
    cols = (uk - rows * n_obs).to(torch.int32)
    indptr = torch.zeros(n_obs + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(rows, minlength=n_obs), 0, out=indptr[1:])
    m = csr_matrix((out.cpu().numpy(), cols.cpu().numpy(), indptr.cpu().numpy()), shape=(n_obs, n_obs))
    m.has_sorted_indices = True
    return m


def knn(adata, n_neighbors: int = 15, use_rep: Optional[str] = None, n_pcs: Optional[int] = None,
        metric: str = "euclidean", key_added: Optional[str] = None, backend=None):
    """Exact k-nearest-neighbour graph of one modality in scanpy's slots (``.obsp["distances"]`` with
    n_neighbors - 1 entries per row, ``.obsp["connectivities"]``, ``.uns["neighbors"]``): the input
    ``neighbors`` expects per modality (reference :366-373 "Run `sc.pp.neighbors` on all modalities first")."""
    be = _backend(backend)
    X = _choose_representation(adata, use_rep, n_pcs)
    X = X.toarray() if issparse(X) else np.asarray(X)
    Xd = be.to_device(np.ascontiguousarray(X, dtype=np.float64))
    n = Xd.shape[0]
    idx, dst = device_knn(Xd, n_neighbors - 1, metric, backend=be)
    self_i = torch.arange(n, device=idx.device)[:, None]
    idx_s = torch.cat([self_i, idx], dim=1)
    dst_s = torch.cat([torch.zeros((n, 1), dtype=dst.dtype, device=dst.device), dst], dim=1)
    conn = fuzzy_simplicial_set(idx_s, dst_s, n, n_neighbors, backend=be)
    k1 = idx.shape[1]
    distances = csr_matrix((be.to_host(dst).reshape(-1), be.to_host(idx).reshape(-1),
                            np.arange(0, n * k1 + 1, k1)), shape=(n, n))
    if key_added is None:
        key_added, ck, dk = "neighbors", "connectivities", "distances"
    else:
        ck, dk = f"{key_added}_connectivities", f"{key_added}_distances"
    adata.obsp[dk] = distances
    adata.obsp[ck] = conn
    params = {"n_neighbors": n_neighbors, "method": "umap", "metric": metric, "random_state": 0}
    if use_rep is not None:
        params["use_rep"] = use_rep
    if n_pcs is not None:
        params["n_pcs"] = n_pcs
    adata.uns[key_added] = {"connectivities_key": ck, "distances_key": dk, "params": params}
    return None


# -----------------------------------------------------------------------------------------------------
# weighted nearest neighbours (reference :264-640)
# -----------------------------------------------------------------------------------------------------
def _mean_operator(be, G: csr_matrix, cols_present=None):
    """The row-normalised pattern of G as the operand of the row-stream SpMM: r_i = mean of X over the neighbours
    of cell i (reference :493-497: ``X[neighbordistances[cell, :].nonzero()[1]]`` - the NON-ZERO stored distances,
    so a duplicate cell at distance 0 does not count), restricted to the columns in ``cols_present``
    (cells the representation exists for).  Built once per (graph, modality) pair and reused for every block."""
    from .io import canonicalize

    G = G.tocsr()
    keep = G.data != 0
    if cols_present is not None:
        keep &= cols_present[G.indices]
    rows = np.repeat(np.arange(G.shape[0]), np.diff(G.indptr))[keep]
    cnt = np.bincount(rows, minlength=G.shape[0])
    indptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    with np.errstate(divide="ignore"):
        vals = np.repeat(1.0 / cnt, cnt).astype(np.float32)
    # (kNN rows are stored by ascending distance: column order is restored on the device)
    Gd = canonicalize(be, be.upload_csr(indptr, G.indices[keep], vals, G.shape, values_dtype=np.float32))
    can = hasattr(be, "can_stream")
    return be.stream(Gd) if can and be.can_stream(Gd, 64) else Gd


def _graph_mean(be, S, X32: torch.Tensor) -> torch.Tensor:
    """``S`` (a ``_mean_operator``) applied to the representation X32 [n, p], 64 columns at a time - the row-stream
    SpMM (csrc/spmm_win.hip), X padded to the block widths it serves."""
    n, p = X32.shape
    out = torch.zeros((n, p), dtype=torch.float32, device=X32.device)
    for c0 in range(0, p, 64):
        w = min(64, p - c0)
        B = 16 if w <= 16 else (32 if w <= 32 else 64)
        Q = torch.zeros((n, B), dtype=torch.float32, device=X32.device)
        Q[:, :w] = X32[:, c0:c0 + w]
        out[:, c0:c0 + w] = be.spmm(S, Q)[:, :w]
    return out


def _bandwidths(be, X: torch.Tensor, G: csr_matrix, n_bandwidth_neighbors: int) -> torch.Tensor:
    """csigma_i (reference :400-472): the mean Euclidean distance from cell i to the
    n_bandwidth_neighbors cells whose kNN sets overlap its own least (but do), ties towards the larger
    distance - the reference's `_jaccard_euclidean_metric` searched exhaustively.  Candidates are the
    cells that share a neighbour (A A^T pattern of the binary kNN graph A), the overlap counts are its
    values."""
    n = X.shape[0]
    G = G.tocsr()
    dev = X.device
    if (hasattr(be, "wnn_bandwidth") and X.dtype == torch.float64 and X.shape[1] <= 256 and X.is_contiguous()
            and n_bandwidth_neighbors <= 64):
        # one wave per cell (csrc/wnn.hip): candidates from the reverse graph, sorted and counted in LDS
        R = G.T.tocsr()
        bbox = float(torch.linalg.norm(X.amax(dim=0) - X.amin(dim=0)))
        cs, over = be.wnn_bandwidth(X, be.to_device(G.indptr, np.int64), be.to_device(G.indices, np.int32),
                                    be.to_device(R.indptr, np.int64), be.to_device(R.indices, np.int32),
                                    n_bandwidth_neighbors, bbox)
        if not over:
            return cs
    rows = torch.as_tensor(np.repeat(np.arange(n), np.diff(G.indptr)), device=dev)
    cols = torch.as_tensor(G.indices.astype(np.int64), device=dev)
    deg = torch.as_tensor(np.diff(G.indptr).astype(np.float64), device=dev)
    A = torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.ones(rows.numel(), dtype=torch.float64, device=dev),
                                (n, n)).coalesce()
    I = torch.sparse.mm(A, A.transpose(0, 1)).coalesce()  # |N(i) & N(j)| for every pair that shares a neighbour
    i, j = I.indices()
    inter = I.values()
    keep = i != j
    i, j, inter = i[keep], j[keep], inter[keep]
    jac_dist = 1.0 - inter / (deg[i] + deg[j] - inter)
    bbox = torch.linalg.norm(X.amax(dim=0) - X.amin(dim=0))
    e = torch.empty(i.numel(), dtype=X.dtype, device=dev)
    step = 1 << 22
    for lo in range(0, i.numel(), step):
        e[lo:lo + step] = _pair_dist(X[i[lo:lo + step]], X[j[lo:lo + step]], "euclidean")
    key = (n - jac_dist * n) + (bbox - e) / bbox
    ok = jac_dist < 1.0
    i, j, e, key = i[ok], j[ok], e[ok], key[ok]
    # per row the n_bandwidth_neighbors smallest keys (ties by column): sort by (row, key, column)
    o = torch.argsort(j, stable=True)
    i, e, key = i[o], e[o], key[o]
    o = torch.argsort(key, stable=True)
    i, e = i[o], e[o]
    o = torch.argsort(i, stable=True)
    i, e = i[o], e[o]
    start = torch.searchsorted(i, torch.arange(n, device=dev))
    rank = torch.arange(i.numel(), device=dev) - start[i]
    take = rank < n_bandwidth_neighbors
    s = torch.zeros(n, dtype=X.dtype, device=dev).index_add_(0, i[take], e[take])
    c = torch.zeros(n, dtype=X.dtype, device=dev).index_add_(0, i[take], torch.ones_like(e[take]))
    return s / c


def neighbors(mdata, n_neighbors: Optional[int] = None, n_bandwidth_neighbors: int = 20,
              n_multineighbors: int = 200, neighbor_keys: Optional[Dict[str, Optional[str]]] = None,
              metric: str = "euclidean", low_memory: Optional[bool] = None, key_added: Optional[str] = None,
              weight_key: Optional[str] = "mod_weight", add_weights_to_modalities: bool = False,
              eps: float = 1e-4, copy: bool = False, random_state=42, *, backend=None):
    """
    Multimodal nearest neighbor search (weighted nearest neighbours of Hao et al. / Swanson et al.).

    Same arguments and slots as the reference (:264-340).  ``low_memory`` and ``random_state`` belong to
    NN-descent and are recorded only: every search here is exhaustive.  Modalities may list the observations in any
    order and may lack cells (r04): a modality then contributes nothing to those cells and gets weight 0 for them.
    """
    if not is_mudata(mdata):
        raise TypeError("Expected a MuData object")
    if metric not in _PAIR_METRICS + ("seuclidean", "mahalanobis"):
        # (at entry, not after the bandwidth and weight work - ADVICE r04; the message names what is offered)
        _pair_dist(torch.zeros((1, 1)), torch.zeros((1, 1)), metric)
    be = _backend(backend)
    mdata = mdata.copy() if copy else mdata
    if neighbor_keys is None:
        modalities = list(mdata.mod.keys())
        neighbor_keys = {}
    else:
        modalities = list(neighbor_keys.keys())
    if len(modalities) < 2:
        raise ValueError("weighted nearest neighbours need at least two modalities")
    observations = mdata.obs.index
    n = len(observations)
    params, reps, mod_reps, mod_n_pcs, mod_k, pos = {}, {}, {}, {}, [], {}
    for mod in modalities:
        nkey = neighbor_keys.get(mod) or "neighbors"
        try:
            nparams = mdata.mod[mod].uns[nkey]
        except KeyError:
            raise ValueError(
                f'Did not find .uns["{nkey}"] for modality "{mod}". Run `sc.pp.neighbors` on all modalities first.'
            )
        use_rep = nparams["params"].get("use_rep", None)
        n_pcs = nparams["params"].get("n_pcs", None)
        mod_k.append(nparams["params"].get("n_neighbors", 0))
        params[mod] = nparams
        X = _choose_representation(mdata.mod[mod], use_rep, n_pcs)
        reps[mod] = X.toarray() if issparse(X) else np.asarray(X)
        mod_reps[mod] = use_rep if use_rep is not None else -1
        mod_n_pcs[mod] = n_pcs if n_pcs is not None else -1
        # where the modality's cells sit among the observations of the MuData object: any order, and a modality may
        # lack cells (reference :381-384, :546-575).  r03 raised unless every modality listed all cells in the same order.
        p = np.asarray(observations.get_indexer(mdata.mod[mod].obs.index))
        if (p < 0).any() or len(np.unique(p)) != len(p):
            raise ValueError(f"modality '{mod}' has observations that the MuData object does not list (or lists twice): "
                             "call `mdata.update()` first")
        pos[mod] = p
    if n_neighbors is None:
        ks = np.asarray([k for k in mod_k if k > 0])
        n_neighbors = int(round(np.mean(ks), 0))
    M = len(modalities)
    # Everything is laid out over the n observations of the MuData object: rows of cells a modality lacks are zero
    # and masked (`pres`).  A modality contributes to a cell's weights, to the neighbourhood means and to the
    # affinity of a pair only where it has the cells involved; its weight for a cell it lacks is 0 (ratio -inf, like
    # the reference's initial value :451).  (The reference walks the joint graph with modality-local row numbers
    # there - `neighbordistances.indptr[cell]`, `weights[cell, i]`, :586-593 - i.e. the rows of other cells as soon as
    # a modality lacks one: the intent, not that indexing, is what is implemented.)
    Xl = {m: be.to_device(np.ascontiguousarray(reps[m], dtype=np.float64)) for m in modalities}  # local order
    dev = Xl[modalities[0]].device
    pos_d = {m: torch.as_tensor(pos[m], device=dev, dtype=torch.int64) for m in modalities}
    pres = {m: np.zeros(n, dtype=bool) for m in modalities}
    Xd, X32, graphs, graphs_local = {}, {}, {}, {}
    for m in modalities:
        pres[m][pos[m]] = True
        full = torch.zeros((n, Xl[m].shape[1]), dtype=torch.float64, device=dev)
        full[pos_d[m]] = Xl[m]
        Xd[m] = full
        X32[m] = full.to(torch.float32).contiguous()
        g = mdata.mod[m].obsp[params[m]["distances_key"]].tocsr()
        cnt = np.diff(g.indptr)
        if (cnt == 0).any():
            i = int(np.nonzero(cnt == 0)[0][0])
            raise ValueError(
                f"Cell {i} in modality {m} does not have any neighbors. "
                "This could be due to subsetting after nearest neighbors calculation. "
                "Make sure to subset before calculating nearest neighbors."
            )
        graphs_local[m] = g
        if len(pos[m]) == n and np.array_equal(pos[m], np.arange(n)):
            graphs[m] = g  # the modality has every cell in mdata's order (the usual case): nothing to remap
        else:
            coo = g.tocoo()
            graphs[m] = csr_matrix((coo.data, (pos[m][coo.row], pos[m][coo.col])), shape=(n, n))
    pres_d = {m: torch.as_tensor(pres[m], device=dev) for m in modalities}
    ratios = torch.full((n, M), -float("inf"), dtype=torch.float64, device=dev)
    sigmas, mean_ops = {}, {}
    ninf = torch.full((n,), -float("inf"), dtype=torch.float64, device=dev)
    for i1, m1 in enumerate(modalities):
        G1 = graphs_local[m1]
        nnd = torch.zeros(n, dtype=torch.float64, device=dev)
        nnd[pos_d[m1]] = torch.as_tensor(np.minimum.reduceat(G1.data, G1.indptr[:-1]).astype(np.float64), device=dev)  # :389-398
        cs_local = _bandwidths(be, Xl[m1], G1, n_bandwidth_neighbors)
        bad = ~torch.isfinite(cs_local)
        if bool(bad.any()):
            # a cell none of whose neighbours' neighbour lists overlaps its own (reference: a mean over an empty
            # selection, NaN, silently carried into the weights - ADVICE r03): say so
            raise ValueError(f"modality '{m1}': {int(bad.sum())} cells (the first: {int(torch.nonzero(bad)[0])}) share no "
                             "neighbour with any other cell - the kernel bandwidth is undefined; increase n_neighbors of "
                             "the modality's graph")
        csig = torch.ones(n, dtype=torch.float64, device=dev)
        csig[pos_d[m1]] = cs_local
        # (a bandwidth equal to the nearest-neighbour distance - duplicated cells - is not an error by itself: the
        #  reference's exp(-x / 0) is 0 for x > 0, a valid ratio; only 0 / 0 is undefined, caught where it happens)
        thetas, cur = [], None
        for i2, m2 in enumerate(modalities):  # :484-506
            # (the operator depends on the graph and on which cells HAVE modality m1: one per graph when every
            #  modality has every cell)
            op_key = (m2, None if pres[m1].all() else m1)
            if op_key not in mean_ops:
                mean_ops[op_key] = _mean_operator(be, graphs[m2], None if pres[m1].all() else pres[m1])
            S = mean_ops[op_key]
            r = _graph_mean(be, S, X32[m1]).to(torch.float64)
            th = torch.exp(-torch.clamp(torch.linalg.norm(Xd[m1] - r, dim=1) - nnd, min=0) / (csig - nnd))
            both = pres_d[m1] & pres_d[m2]
            undefined = both & torch.isnan(th)
            if bool(undefined.any()):
                raise ValueError(f"modality '{m1}': the affinity ratio of cell {int(torch.nonzero(undefined)[0])} is 0 / 0 (its "
                                 "kernel bandwidth equals its nearest-neighbour distance and the neighbourhood mean "
                                 f"in '{m2}' sits on the cell itself: duplicated cells?)")
            th = torch.where(both, th, ninf)
            if i1 == i2:
                cur = th
            else:
                thetas.append(th)
        ratio = cur / (torch.stack(thetas, dim=1).amax(dim=1) + eps)  # :507
        ratios[:, i1] = torch.where(pres_d[m1], ratio, ninf)
        sigmas[m1] = csig
    weights = torch.softmax(ratios, dim=1)  # :510
    # candidates: the union of every modality's n_multineighbors nearest neighbours (:517-575)
    keys = []
    for m in modalities:
        idx, _ = device_knn(Xl[m], n_multineighbors, params[m].get("metric", "euclidean"), backend=be, indices_only=True)  # (:520: the top-level key)
        keys.append((pos_d[m][:, None] * n + pos_d[m][idx]).reshape(-1))
    key = torch.unique(torch.cat(keys))  # sorted: row-major
    ri = torch.div(key, n, rounding_mode="floor")
    ci = key - ri * n
    aff = torch.zeros(key.numel(), dtype=torch.float64, device=dev)
    step = 1 << 22
    for i, m in enumerate(modalities):  # :579-609
        X = Xd[m]
        if metric in ("seuclidean", "mahalanobis"):  # (their (co)variances are those of each cell's own cdist call)
            both = pres_d[m][ri] & pres_d[m][ci]
            d = _cell_dist(X, ri, ci, both, pres_d[m], metric)
            term = torch.exp(-d / sigmas[m][ri]) * weights[ri, i]
            aff += torch.where(both, term, torch.zeros_like(term))
            continue
        for lo in range(0, key.numel(), step):
            a, b = ri[lo:lo + step], ci[lo:lo + step]
            term = torch.exp(-_pair_dist(X[a], X[b], metric) / sigmas[m][a]) * weights[a, i]
            aff[lo:lo + step] += torch.where(pres_d[m][a] & pres_d[m][b], term, torch.zeros_like(term))
    dist = torch.sqrt(torch.clamp(0.5 * (1.0 - aff), min=0.0))  # :610
    # the n_neighbors + 1 smallest per row (:612 `_sparse_csr_fast_knn`): sort by (row, distance, column)
    o = torch.argsort(dist, stable=True)
    ri_s, ci_s, d_s = ri[o], ci[o], dist[o]
    o = torch.argsort(ri_s, stable=True)
    ri_s, ci_s, d_s = ri_s[o], ci_s[o], d_s[o]
    start = torch.searchsorted(ri_s, torch.arange(n, device=dev))
    rank = torch.arange(ri_s.numel(), device=dev) - start[ri_s]
    k1 = n_neighbors + 1
    take = rank < k1
    cnt = torch.bincount(ri_s[take], minlength=n)
    if int(cnt.min()) < k1:
        raise ValueError(f"fewer than n_neighbors + 1 = {k1} candidate neighbours for some cells: raise n_multineighbors")
    knn_idx = ci_s[take].reshape(n, k1)
    knn_d = d_s[take].reshape(n, k1)
    distances = csr_matrix((be.to_host(knn_d).reshape(-1), be.to_host(knn_idx).reshape(-1),
                            np.arange(0, n * k1 + 1, k1)), shape=(n, n))
    connectivities = fuzzy_simplicial_set(knn_idx, knn_d, n, k1, backend=be)  # :615-622

    w_host = be.to_host(weights)
    for i, m in enumerate(modalities):  # :583-588
        if weight_key:
            if add_weights_to_modalities:
                mdata.mod[m].obs[weight_key] = w_host[pos[m], i]
            else:
                mdata.obs[":".join([m, weight_key])] = w_host[:, i]
    if key_added is None:
        key_added, conns_key, dists_key = "neighbors", "connectivities", "distances"
    else:
        conns_key, dists_key = f"{key_added}_connectivities", f"{key_added}_distances"
    mdata.obsp[dists_key] = distances
    mdata.obsp[conns_key] = connectivities
    mdata.uns[key_added] = {
        "connectivities_key": conns_key, "distances_key": dists_key,
        "params": {"n_neighbors": n_neighbors, "n_multineighbors": n_multineighbors, "metric": metric, "eps": eps,
                   "random_state": random_state, "use_rep": mod_reps, "n_pcs": mod_n_pcs, "method": "umap"},
    }
    if hasattr(mdata, "update_obs"):
        mdata.update_obs()
    return mdata if copy else None
