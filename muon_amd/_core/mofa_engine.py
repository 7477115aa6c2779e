"""MOFA+ variational inference engine on MI355X (Gaussian likelihood).

Replaces the `ent.build(); ent.run()` section of the reference
(/root/reference/muon/_core/tools.py:583-585, arithmetic in the third-party mofapy2) with a
sufficient-statistics formulation that never densifies a sparse view
(the reference densifies every modality, tools.py:117-141):

  per iteration and view   B_g = Y_g^T <Z_g>   (D x K)   one pass over Y   [W, tau, ELBO]
                           A   = Y (tau_g o <W>) (N x K)  one pass over Y   [Z]

Sparse views go through the CSR SpMM kernel in both directions (X and its device transpose),
group blocks stacked along the dense dimension so that one pass serves all groups; centring is
applied implicitly as a rank-one correction ((Y - 1 mu^T) W = Y W - 1 (mu^T W)), so the view
stays CSR.  Dense views go through the tall-skinny MFMA kernels mu_skinny_nn / mu_skinny_tn
(n_factors <= 16; PyTorch-ROCm GEMMs otherwise and for the K x K blocks).  The Gauss-Seidel
sweeps over factors are the fused HIP kernels mu_mofa_update_w / mu_mofa_update_z; the
remaining O(D K + G K^2) node updates (tau, alpha, theta, ELBO) are torch element-wise ops.
When samples are sharded over ranks, the statistics B, <Z>^T<Z>, sum <z^2> and the sample
part of the ELBO are all-reduced (RCCL) and the W / tau / alpha / theta updates are replicated.

Update equations: oracle/mofa_oracle.py (same schedule and initialisation, so the two can be
compared iteration by iteration).
"""
from __future__ import annotations

import math
import os
import time
import warnings
from typing import List, Optional

import numpy as np
import torch
from scipy.sparse import issparse

from .._comm import default_comm

A0 = 1e-14
B0 = 1e-14
TH_A0 = 1.0
TH_B0 = 1.0
TOL = {"fast": 5e-4, "medium": 5e-5, "slow": 5e-6}


def _pad_block(w: int) -> int:
    for b in (16, 32, 64):
        if w <= b:
            return b
    raise NotImplementedError("n_groups * n_factors must be <= 64 for sparse views")


def _can_ell16(be, X, wide: bool) -> bool:
    """The sliced-ELL kernels address both dense operands with 32-bit byte offsets (64-byte rows of an f32 block, 128-byte
    rows of an f64 one): X and X^T must both stay under 4 GiB of dense rows, else the row-stream path is taken."""
    fn = getattr(be, "can_ell16", None)
    if fn is not None:
        return bool(fn(X, wide))
    row = 128 if wide else 64
    return max(X.shape) * row < (1 << 32)


class _View:
    pass


class MofaEngine:
    def __init__(self, backend, views: List, groups: np.ndarray, n_factors: int, *,
                 dtype=torch.float64, center_groups=True, scale_views=False, scale_groups=False,
                 ard_weights=True, ard_factors=True, spikeslab_weights=True, seed=1, comm=None,
                 row_offset: int = 0, n_total: Optional[int] = None):
        """``views``: list of (N x D_m) scipy CSR or ndarray; rows that are entirely NaN (dense)
        or flagged in ``present`` are samples missing from that view.  ``groups``: int [N]."""
        self.be = backend
        self.comm = default_comm(comm)
        self.T = dtype
        # MUON_AMD_MOFA_PROFILE=1: wall time of the set-up's stages (synchronising between them) in `setup_profile`
        self.setup_profile = [] if os.environ.get("MUON_AMD_MOFA_PROFILE", "0") == "1" else None
        self._t_last = time.perf_counter()
        self.K = int(n_factors)
        if not (1 <= self.K <= 32):
            raise NotImplementedError("1 <= n_factors <= 32")
        self.opts = dict(ard_weights=ard_weights, ard_factors=ard_factors,
                         spikeslab_weights=spikeslab_weights)
        groups = np.asarray(groups, dtype=np.int64)
        self.N = N = len(groups)
        # the seeded host draw of the factors' initial expectations (8 ns per normal, the GIL released): on a host
        # thread while the device works through the views
        import threading

        z0 = {}
        nt = N if n_total is None else int(n_total)
        th = threading.Thread(target=lambda: z0.setdefault("z", np.random.default_rng(seed).standard_normal((nt, self.K))),
                              daemon=True)
        th.start()
        self._z0 = (th, z0)
        self.G = G = self._global_max(groups) + 1
        # samples sorted by group (stable) so that every group is a contiguous row range
        self.perm = np.argsort(groups, kind="stable")
        gs = groups[self.perm]
        self.gslice = [(int(np.searchsorted(gs, g, "left")), int(np.searchsorted(gs, g, "right")))
                       for g in range(G)]
        self.grp = backend.to_device(gs.astype(np.int32))
        self.Ng = self._allreduce(torch.tensor([b - a for a, b in self.gslice], dtype=torch.float64))
        self.M = len(views)
        self._mark("groups")
        self.views = [self._prepare_view(v, center_groups, scale_views, scale_groups) for v in views]
        self.Ds = [v.D for v in self.views]
        self._init_state(seed, row_offset, n_total)
        self._mark("state")
        self._stats = {}
        self._stat_buf = {}
        self.elbo = []
        # one iteration is ~150 short launches (the K x K algebra of the tau / alpha / theta / ELBO
        # terms): on a GPU, and without collectives inside the step, it is captured once into a HIP
        # graph and replayed, so the host only launches the graph and reads the ELBO back
        self._graph = None
        self._graph_elbo = None
        self._graph_ok = (getattr(backend, "name", "") == "hip" and self.comm.world_size == 1
                          and os.environ.get("MUON_AMD_MOFA_GRAPH", "1") != "0")
        self._eager_steps = 0
        # several ranks: the iteration as two captured segments around one packed all-reduce (see _seg_a); on operator
        # sets without graphs (CPU tests) the segments run eagerly - same schedule, one collective per iteration
        self._loc = {}
        self._seg_graphs = None
        self._seg = bool(self.comm.world_size > 1 and getattr(self, "_fused", False)
                         and os.environ.get("MUON_AMD_MOFA_SEGMENTS", "1") != "0")
        self._seg_ok = (self._seg and getattr(backend, "name", "") == "hip"
                        and os.environ.get("MUON_AMD_MOFA_GRAPH", "1") != "0")

    # -- helpers -------------------------------------------------------------------------
    def _mark(self, label):
        if self.setup_profile is not None:
            if getattr(self.be, "name", "") == "hip":
                torch.cuda.synchronize(self.be.device)
            now = time.perf_counter()
            self.setup_profile.append((label, (now - self._t_last) * 1e3))
            self._t_last = now

    def _global_max(self, groups):
        m = int(groups.max()) if groups.size else 0
        if self.comm.world_size > 1:
            t = torch.tensor([m], dtype=torch.int64)
            if getattr(self.be, "name", "") == "hip":
                t = t.to(self.be.device)
            m = int(self.comm.all_reduce_max(t).item())
        return m

    def _allreduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.comm.world_size > 1:
            dev = t.device
            if getattr(self.be, "name", "") == "hip" and not t.is_cuda:
                t = t.to(self.be.device)
            self.comm.all_reduce_sum(t)
            t = t.to(dev)
        return t

    def _all_ranks(self, flag: bool) -> bool:
        """True iff ``flag`` holds on EVERY rank (a per-shard property that selects a code path with collectives)."""
        if self.comm.world_size == 1:
            return bool(flag)
        bad = self._allreduce(torch.tensor([0.0 if flag else 1.0], dtype=torch.float64))
        return float(bad.item()) == 0.0

    def _pad16(self, A: torch.Tensor) -> torch.Tensor:
        out = torch.zeros((A.shape[0], 16), dtype=A.dtype, device=A.device)
        out[:, : A.shape[1]] = A
        return out

    def _tn(self, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
        """A^T B for tall blocks with <= 16 columns (K x K Grams of the factor / weight blocks).
        rocBLAS' f64 GEMM takes 4-11 ms for these 10 x 10 outputs with a 1e5-long inner dimension
        (Tensile 128x128 macro tiles); the tall-skinny MFMA kernel streams the blocks once."""
        if A.shape[1] <= 16 and B.shape[1] <= 16 and A.shape[0] >= 1024 and hasattr(self.be, "skinny_tn"):
            return self.be.skinny_tn(self._pad16(A), self._pad16(B))[: A.shape[1], : B.shape[1]]
        return A.T @ B

    def _dev(self, arr, dtype=None):
        t = self.be.to_device(np.ascontiguousarray(arr))
        return t.to(dtype or self.T)

    def _f32_storage_ok(self, scale_views, scale_groups):
        """f64 fit, dense view whose values are exact in f32 (AnnData's default dtype; tools.py:308 widens it for the
        default use_float32=False): the view stays in f32 in HBM - half the bytes of the two passes over it per
        iteration, the bound of the f64 fit - and is widened in registers (mu_skinny_*_f64_f32); products and sums
        are the f64 ones.  The centred values are not exact in f32, so the centring moves into the products, as for
        the sparse views: Y (tau o W) - pres (mu^T (tau o W)) and Y^T Z - mu (pres^T Z).  Against centring first the
        result differs by rounding only (a few ulp of the uncentred products: |mu| / std digits at most).  Not with
        view / group scaling (the scaled values are not exact in f32 either)."""
        be = self.be
        return (self.T == torch.float64 and not scale_views and not scale_groups and self.K <= 16
                and getattr(be, "skinny_mixed", False) and hasattr(be, "skinny_nn") and hasattr(be, "mofa_rowstats")
                and os.environ.get("MUON_AMD_MOFA_F32_STORAGE", "1") != "0")

    def _prepare_view(self, v, center_groups, scale_views, scale_groups):
        be, T, G = self.be, self.T, self.G
        V = _View()
        N = self.N
        V.D = D = v.shape[1]
        from .._backend import DeviceCSR

        if isinstance(v, (DeviceCSR, torch.Tensor)):
            # device-resident view (bench / pipelines): rows must already be in group order
            if not np.array_equal(self.perm, np.arange(N)):
                raise NotImplementedError("device-resident views need samples sorted by group")
            pres = np.ones(N, dtype=bool)
            if isinstance(v, DeviceCSR):
                V.kind = "sparse"
                if v.nnz >= 2 and hasattr(be, "lib"):
                    # the operand layouts below rank an entry inside its (row, slab) by position: rows must be sorted
                    # by column (a caller's device-resident CSR is not canonicalised anywhere else - ADVICE r04)
                    from .io import canonicalize

                    v = canonicalize(be, v)
                V.X = v.with_values(v.values.to(T, copy=True))  # centring / scaling work in place:
            else:                                               # never on the caller's tensors
                V.kind = "dense"
                if self._all_ranks(self._f32_storage_ok(scale_views, scale_groups) and v.dtype == torch.float32):
                    V.Y = v.contiguous()  # read-only from here on: no copy (see _f32_storage_ok)
                    V.implicit = True
                elif v.dtype == T and v.is_contiguous() and not scale_views and not scale_groups:
                    V.Y = v            # the caller's tensor, read only: the centred copy is written in ONE pass below
                    V.alias = True     # (r04: copy, then centre in place - two more passes over the view)
                else:
                    V.Y = v.to(T, copy=True)
        elif issparse(v):
            m = v.tocsr()[self.perm]
            m.sort_indices()
            V.kind = "sparse"
            pres = np.ones(N, dtype=bool)
            if getattr(v, "_missing_rows", None) is not None:
                pres = ~np.asarray(v._missing_rows)[self.perm]
            host_vals = m.data.astype(np.float64)
            if np.isnan(host_vals).any():
                raise NotImplementedError("element-wise missing values are not supported")
            X = be.upload_csr(m.indptr, m.indices, host_vals, m.shape)
            V.X = X.with_values(X.values.to(T))
        else:
            a = np.asarray(v, dtype=np.float64)[self.perm]
            nanrow = np.isnan(a).any(axis=1)
            if np.isnan(a[~nanrow]).any() or (nanrow & ~np.isnan(a).all(axis=1)).any():
                raise NotImplementedError("element-wise missing values are not supported")
            pres = ~nanrow
            V.kind = "dense"
            a = np.where(pres[:, None], a, 0.0)
            a32 = a.astype(np.float32)
            # (the storage mode is ONE decision for all ranks: a shard that is exact in f32 next to one that is not would
            #  all-reduce uncentred Y^T Z contributions with centred ones - ADVICE r04)
            if self._all_ranks(self._f32_storage_ok(scale_views, scale_groups) and np.array_equal(a32.astype(np.float64), a)):
                V.Y = self._dev(a32, torch.float32)
                V.implicit = True
            else:
                V.Y = self._dev(a)
        self._mark(f"{V.kind}: values")
        V.pres = self._dev(pres.astype(np.float64))
        V.Ngm = self._allreduce(torch.tensor([float(pres[a:b].sum()) for a, b in self.gslice],
                                             dtype=torch.float64))
        # per (group, feature) first and second moments over the observed samples
        s1 = torch.zeros((G, D), dtype=T, device=V.pres.device)
        s2 = torch.zeros((G, D), dtype=T, device=V.pres.device)
        for g, (a, b) in enumerate(self.gslice):
            if V.kind == "dense" and hasattr(be, "col_moments") and V.Y.is_contiguous():
                # one pass, f64 sums (csrc/mofa_stats.hip) - also over the f32-stored block of an f64 fit
                if b > a:
                    m1, m2 = be.col_moments(V.Y, a, b)
                    s1[g], s2[g] = m1.to(T), m2.to(T)
            elif V.kind == "dense" and getattr(V, "implicit", False):
                for c0 in range(a, b, 16384):  # (f64 moments of the f32-stored block, a slab of rows at a time)
                    blk = V.Y[c0:min(b, c0 + 16384)].to(T)
                    s1[g] += blk.sum(dim=0)
                    s2[g] += (blk * blk).sum(dim=0)
            elif V.kind == "dense":
                s1[g] = V.Y[a:b].sum(dim=0)
                s2[g] = (V.Y[a:b] ** 2).sum(dim=0)
            elif hasattr(be, "lib") and hasattr(be, "row_col_sums"):
                # the per-feature moments of a group's rows are column sums of the values and of their squares: the
                # TF-IDF sum sweep (f64 sums in LDS bins) instead of two index_add_ over every entry (r04: 2 x 12.8 ms)
                from .._backend import DeviceCSR

                lo, hi = int(V.X.indptr[a].item()), int(V.X.indptr[b].item())
                sub = DeviceCSR((V.X.indptr[a:b + 1] - lo).contiguous(), V.X.indices[lo:hi], V.X.values[lo:hi], (b - a, D))
                if b > a and hi > lo:
                    s1[g] = be.row_col_sums(sub)[1].to(T)
                    be.__dict__.pop("_sweep_work", None)
                    s2[g] = be.row_col_sums(sub.with_values(V.X.values[lo:hi] ** 2))[1].to(T)
                    be.__dict__.pop("_sweep_work", None)
            else:
                lo, hi = int(V.X.indptr[a].item()), int(V.X.indptr[b].item())
                idx = V.X.indices[lo:hi].long()
                s1[g].index_add_(0, idx, V.X.values[lo:hi])
                s2[g].index_add_(0, idx, V.X.values[lo:hi] ** 2)
        self._mark(f"{V.kind}: moments")
        s1 = self._allreduce(s1)
        s2 = self._allreduce(s2)
        n = V.Ngm.to(s1.device).to(T).clamp(min=1.0)[:, None]
        mu = s1 / n
        V.intercepts = mu.clone()  # tools.py:283-286: nanmean per (view, group)
        if not center_groups:
            # mofapy2 process_data: without group centring every feature still loses its mean over
            # ALL observed samples (r01 left such data uncentred)
            mu = (s1.sum(dim=0) / V.Ngm.to(s1.device).to(T).sum().clamp(min=1.0))[None, :].expand(G, D).contiguous()
        c1 = s1 - n * mu                     # sum (y - mu)   over the observed samples of a group
        yy = s2 - 2 * mu * s1 + n * mu * mu  # sum (y - mu)^2
        # scale_views: the centred view / its nanstd; scale_groups (applied after it): every group's
        # block / its nanstd, which cancels the view's scalar.  nanstd subtracts the scalar mean of
        # the block (zero under group centring, not otherwise).
        scale = torch.ones((G,), dtype=T, device=s1.device)
        if scale_groups:
            for g in range(G):
                cnt = V.Ngm[g].item() * D
                if cnt > 0:
                    var = yy[g].sum() / cnt - (c1[g].sum() / cnt) ** 2
                    if var > 0:
                        scale[g] = 1.0 / math.sqrt(float(var))
        elif scale_views:
            cnt = float(V.Ngm.sum().item()) * D
            if cnt > 0:
                var = float(yy.sum().item()) / cnt - (float(c1.sum().item()) / cnt) ** 2
                if var > 0:
                    scale = scale / math.sqrt(var)
        if scale_groups or scale_views:
            for g, (a, b) in enumerate(self.gslice):
                if V.kind == "dense":
                    V.Y[a:b] *= scale[g]
                else:
                    lo, hi = int(V.X.indptr[a].item()), int(V.X.indptr[b].item())
                    V.X.values[lo:hi] *= scale[g]
            mu = mu * scale[:, None]
            yy = yy * scale[:, None] ** 2
        V.yy = yy
        V.mu = mu
        self._mark(f"{V.kind}: scaling")
        try:
            return self._layout_view(V)
        finally:
            self._mark(f"{V.kind}: centring / layouts")

    def _layout_view(self, V):
        be, mu = self.be, V.mu
        if V.kind == "dense" and getattr(V, "implicit", False):
            V.Yt = None  # centred in the products, like the sparse views: Y stays what it was (exact in f32)
        elif V.kind == "dense":
            # explicit centring of the dense block, one pass: y - pres mu (pres is 0 / 1: the product is exact)
            src = V.Y
            if getattr(V, "alias", False):
                V.Y = torch.empty_like(src)
                V.alias = False
            for g, (a, b) in enumerate(self.gslice):
                if b > a:
                    torch.addcmul(src[a:b], V.pres[a:b, None], mu[g][None, :], value=-1.0, out=V.Y[a:b])
            V.Yt = None
        elif (self.T == torch.float32 and hasattr(be, "ell16") and _pad_block(self.G * self.K) == 16
              and V.X.shape[0] > 0 and V.X.shape[1] > 0 and _can_ell16(be, V.X, wide=False)):
            # f32, factor blocks of <= 16 columns: the operand of both directions never changes during a fit - laid
            # out once as sliced ELL (csrc/spmm_ell.hip): no per-row protocol is left in the product
            # (the transposed operand through the tile-staged transposition, csrc/tpack4.hip: 3 ms where the general
            #  kernel behind `transpose` took 14)
            V.Xt = be.ell16(be.transpose_csr(V.X) if hasattr(be, "transpose_csr") else be.transpose(V.X))
            V.Xs = be.ell16(V.X)
        elif self.T == torch.float32 and hasattr(be, "can_stream") and be.can_stream(V.X, 16):
            # f32: both directions read row streams, built once
            V.Xt = be.transpose_stream(V.X)
            V.Xs = be.stream(V.X)
        elif (self.T == torch.float64 and hasattr(be, "ell16") and _pad_block(self.G * self.K) == 16
              and V.X.shape[0] > 0 and V.X.shape[1] > 0 and _can_ell16(be, V.X, wide=True)):
            # f64 (the reference's default precision), factor blocks of <= 16 columns: the same static layout with
            # f32 stored values - one operand when the data is exact in f32, hi + lo otherwise - against f64 blocks
            if hasattr(be, "ell16_pair"):
                V.Xs, V.Xt = be.ell16_pair(V.X, wide=True)
            else:
                V.Xt = be.ell16(be.transpose(V.X), wide=True)
                V.Xs = be.ell16(V.X, wide=True)
        elif (self.T == torch.float64 and hasattr(be, "split_streams") and _pad_block(self.G * self.K) <= 32
              and V.X.shape[0] > 0 and V.X.shape[1] > 0):
            # f64 (the reference's default precision): the same row streams with f32 stored values -
            # one stream when the data is exact in f32, hi + lo otherwise - against f64 blocks
            V.Xs, V.Xt = be.split_streams(V.X)
        else:
            V.Xt = be.transpose(V.X)
            V.Xs = V.X
        return V

    def _init_state(self, seed, row_offset, n_total):
        K, G, T = self.K, self.G, self.T
        n_total = self.N if n_total is None else int(n_total)
        th, box = self._z0
        th.join()
        del self._z0
        z0 = box["z"][row_offset:row_offset + self.N]
        self.EZ = self._dev(z0[self.perm])
        self.EZ2 = self.EZ ** 2 + 1.0
        self.sig2z = torch.ones_like(self.EZ)
        dev = self.EZ.device
        c = float(torch.digamma(torch.tensor(1.0, dtype=torch.float64)) - torch.digamma(torch.tensor(2.0, dtype=torch.float64)))
        self.W = []
        for D in self.Ds:
            w = _View()
            w.EW = torch.zeros((D, K), dtype=T, device=dev)
            w.EW2 = torch.ones((D, K), dtype=T, device=dev)
            w.gamma = torch.ones((D, K), dtype=T, device=dev)
            w.EWh2 = torch.ones((D, K), dtype=T, device=dev)
            w.sig2 = torch.ones((D, K), dtype=T, device=dev)
            w.tau = torch.ones((G, D), dtype=T, device=dev)
            w.ltau = torch.zeros((G, D), dtype=T, device=dev)
            w.alpha = torch.ones((K,), dtype=T, device=dev)
            w.lalpha = torch.zeros((K,), dtype=T, device=dev)
            w.lth = torch.full((K,), c, dtype=T, device=dev)
            w.l1mth = torch.full((K,), c, dtype=T, device=dev)
            self.W.append(w)
        self.alpha_z = torch.ones((G, K), dtype=T, device=dev)
        self.lalpha_z = torch.zeros((G, K), dtype=T, device=dev)
        # constants of the Gamma updates, resident so that an iteration has no host -> device copies
        self.Ng_d = self.Ng.to(dev).to(T)
        self._Ng64 = self.Ng.to(dev).to(torch.float64).contiguous()
        self._zs = torch.zeros((G, 2, K), dtype=torch.float64, device=dev)
        self._elbo_work = self.be.mofa_elbo_work(K)
        for V, w in zip(self.views, self.W):
            V.Ngm_d = V.Ngm.to(dev).to(T).contiguous()
            V.yy = V.yy.contiguous()
            w.a_alpha = torch.tensor(A0 + 0.5 * V.D, dtype=T, device=dev)
        # fused statistics (csrc/mofa_stats.hip): operands and moments at fixed addresses, the zero
        # padding of the operand blocks written once
        self._fused = hasattr(self.be, "mofa_rowstats")
        if self._fused:
            M = self.M
            self._rs_work = self.be.mofa_rowstats_work(K)
            self._A = torch.zeros((M, self.N, K), dtype=T, device=dev)
            self._Gw = torch.zeros((M, G, K, K), dtype=T, device=dev)
            self._dw2 = torch.zeros((M, G, K), dtype=T, device=dev)
            self._corr = torch.zeros((M, G, K), dtype=T, device=dev)
            self._pres = torch.stack([v.pres for v in self.views]).contiguous()
            self._zpad = {}
            for V in self.views:
                V.full = bool((V.pres == 1).all().item())
                if V.kind == "sparse":
                    V.ld = _pad_block(G * K)
                    V.TWs = torch.zeros((V.D, V.ld), dtype=T, device=dev)
                elif K <= 16 and hasattr(self.be, "skinny_tn"):
                    V.ld = 16
                    # f32: the library GEMM for A = Y (tau o W) had been 3 % ahead of mu_skinny_nn; with the branch-free,
                    # asm-prefetching build of r04 (views of whole 128-byte tiles) the kernel is 2 % ahead of it
                    # (c4: 0.465 against 0.475 s with the library GEMM)
                    own_f32 = V.D % 32 == 0
                    if (T == torch.float64 or own_f32) and hasattr(self.be, "skinny_nn"):
                        V.T16 = [torch.zeros((V.D, 16), dtype=T, device=dev) for _ in range(G)]
                    else:
                        V.TWt = [torch.zeros((K, V.D), dtype=T, device=dev) for _ in range(G)]
                else:
                    V.ld = 0
                    V.TWt = [torch.zeros((K, V.D), dtype=T, device=dev) for _ in range(G)]
                if V.ld and (V.ld, V.kind == "sparse") not in self._zpad:
                    self._zpad[(V.ld, V.kind == "sparse")] = torch.zeros((self.N, V.ld), dtype=T, device=dev)
            self._zmom = {}
        # r05: the views' shares of an iteration are independent between the joins W updates | products A | Z update |
        # statistics, tau, alpha / theta per view | factor node - each view's share runs on its own stream (fork / join by
        # events; captured into the HIP graph as parallel branches), so the ~25 small kernels of one view hide under the
        # other view's product and the HBM-bound dense product overlaps the sliced-ELL one.  Same kernels, same operands:
        # the same numbers; the ELBO adds the views' terms at the end (per-view scalars: the kernels add to a scalar in
        # place).  One process, fused path only.
        # (several ranks: the first, eager iteration has collectives inside the views' shares and stays on one stream;
        #  the segmented iterations that follow - _seg_a / _seg_b, no collective inside - take the views' streams too)
        self._par_ok = bool(self._fused and self.M > 1 and getattr(self.be, "name", "") == "hip")
        self._par = self._par_ok and self.comm.world_size == 1
        self._side = [torch.cuda.Stream(self.be.device) for _ in range(self.M - 1)] if self._par_ok else []
        if self._fused:
            self._rs_work_v = [self._rs_work] + [self.be.mofa_rowstats_work(K) for _ in range(self.M - 1)]
        self._elbo_work_v = [self._elbo_work] + [self.be.mofa_elbo_work(K) for _ in range(self.M - 1)]

    # -- the views' streams -------------------------------------------------------------------
    def _fork(self):
        if self._par:
            main = torch.cuda.current_stream(self.be.device)
            for s in self._side:
                s.wait_stream(main)

    def _on(self, m):
        """Context in which view m's share of a phase is issued (view 0: the current stream)."""
        import contextlib

        if self._par and m > 0:
            return torch.cuda.stream(self._side[m - 1])
        return contextlib.nullcontext()

    def _join(self):
        if self._par:
            main = torch.cuda.current_stream(self.be.device)
            for s in self._side:
                main.wait_stream(s)

    # -- sufficient statistics --------------------------------------------------------------
    def _zstats(self, m):
        # B computed after the Z update serves tau / ELBO of this iteration AND the W update of
        # the next one (Z does not change in between): two passes over Y per iteration, not three
        cached = self._stats.get(m)
        if cached is not None:
            return cached
        st = self._zstats_compute(m)
        self._stats[m] = st
        return st

    def _zstats_compute(self, m):
        st = self._zstats_fresh(m)
        buf = self._stat_buf.get(m)
        if buf is None:
            self._stat_buf[m] = buf = tuple(t.clone() for t in st)
        else:
            # fixed addresses: a captured iteration reads the statistics its predecessor wrote
            for dst, src in zip(buf, st):
                dst.copy_(src)
        return buf

    def _z_moments(self, m):
        """(Gz, Z2, Zs) of view m's presence mask, one pass per group over the factor block
        (mu_mofa_rowstats); views that observe every sample share one set.  The first call after a Z
        update also refreshes the padded <Z> operands of the products B = Y^T <Z>."""
        V, K, G, T, dev = self.views[m], self.K, self.G, self.T, self.EZ.device
        key = "all" if V.full else m
        got = self._zmom.get(key)
        if got is None:
            fixed = self.__dict__.get("_zmom_out", {}).get(key)  # (segmented iterations: views of one flat block)
            if fixed is not None:
                Gz, Z2, Zs = fixed
            else:
                Gz = torch.empty((G, K, K), dtype=T, device=dev)
                Z2 = torch.empty((G, K), dtype=T, device=dev)
                Zs = torch.empty((G, K), dtype=T, device=dev)
            pads = list(self._zpad.items()) if not self._zmom else []
            for g, (a, b) in enumerate(self.gslice):
                (ld0, st0), pad0 = pads[0] if pads else ((0, False), None)
                self.be.mofa_rowstats(self.EZ, self.EZ2, a, b, self._rs_work_v[m], wgt=None if V.full else V.pres,
                                      out_pad=pad0, col0=g * K if st0 else 0, gram=Gz[g], s2=Z2[g], s1=Zs[g])
                for (ld, st), pad in pads[1:]:
                    c0 = g * K if st else 0
                    pad[a:b, c0:c0 + K] = self.EZ[a:b]
            got = self._zmom[key] = (Gz, Z2, Zs)
        return got

    def _stats_local(self, m):
        """This rank's share of view m's statistics - (Gz, Z2, Zs) of its presence mask and B_g = Y_g^T <Z_g> - before any
        sum over the ranks and before the implicit centring (which needs the summed Zs)."""
        V, K, G = self.views[m], self.K, self.G
        Gz, Z2, Zs = self._z_moments(m)
        if V.kind == "sparse":
            out = self.be.spmm(V.Xt, self._zpad[(V.ld, True)])  # D x (G K): X_g^T <Z_g> for every group in one pass
            B = torch.stack([out[:, g * K:(g + 1) * K] for g in range(G)]) if G > 1 else out[None, :, :K]
        elif V.ld == 16:
            Z16 = self._zpad[(16, False)]
            B = torch.stack([self.be.skinny_tn(V.Y[a:b], Z16[a:b])[:, :K] for a, b in self.gslice])
        else:
            B = torch.stack([V.Y[a:b].T @ self.EZ[a:b] for a, b in self.gslice])
        return Gz, Z2, Zs, B

    def _zstats_fused(self, m):
        V = self.views[m]
        Gz, Z2, Zs, B = self._stats_local(m)
        if self.comm.world_size > 1:
            B = B.contiguous()
            Gz, Z2, Zs = Gz.clone(), Z2.clone(), Zs.clone()  # (shared between views: reduce copies)
            self.comm.all_reduce_sum(Gz, Z2, Zs, B)
        if V.kind == "sparse" or getattr(V, "implicit", False):
            B = B - V.mu[:, :, None] * Zs[:, None, :]  # implicit centring
        return Gz, Z2, B.contiguous()

    def _zstats_fresh(self, m):
        if getattr(self, "_fused", False):
            return self._zstats_fused(m)
        V, K, G = self.views[m], self.K, self.G
        Gz = torch.zeros((G, K, K), dtype=self.T, device=self.EZ.device)
        Z2 = torch.zeros((G, K), dtype=self.T, device=self.EZ.device)
        Zs = torch.zeros((G, K), dtype=self.T, device=self.EZ.device)
        for g, (a, b) in enumerate(self.gslice):
            zp = self.EZ[a:b] * V.pres[a:b, None]
            Gz[g] = self._tn(zp, zp)
            Z2[g] = (self.EZ2[a:b] * V.pres[a:b, None]).sum(dim=0)
            Zs[g] = zp.sum(dim=0)
        if V.kind == "dense" and K <= 16 and hasattr(self.be, "skinny_tn"):
            # B_g = Y_g^T <Z_g> on the matrix cores, one pass over Y (csrc/skinny.hip)
            Bl = []
            for a, b in self.gslice:
                Z16 = torch.zeros((b - a, 16), dtype=self.T, device=self.EZ.device)
                Z16[:, :K] = self.EZ[a:b]
                Bl.append(self.be.skinny_tn(V.Y[a:b], Z16)[:, :K])
            B = torch.stack(Bl)
        elif V.kind == "dense":
            B = torch.stack([V.Y[a:b].T @ self.EZ[a:b] for a, b in self.gslice])
        else:
            Bp = _pad_block(G * K)
            Zst = torch.zeros((self.N, Bp), dtype=self.T, device=self.EZ.device)
            for g, (a, b) in enumerate(self.gslice):
                Zst[a:b, g * K:(g + 1) * K] = self.EZ[a:b]
            out = self.be.spmm(V.Xt, Zst)  # D x (G K): X_g^T <Z_g> for every group in one pass
            B = torch.stack([out[:, g * K:(g + 1) * K] for g in range(G)]).contiguous()
        if self.comm.world_size > 1:
            self.comm.all_reduce_sum(Gz, Z2, Zs, B)
        if V.kind == "sparse":
            B = B - V.mu[:, :, None] * Zs[:, None, :]  # implicit centring
        return Gz, Z2, B.contiguous()

    def _update_w(self, m):
        V, Wm = self.views[m], self.W[m]
        Gz, Z2, B = self._zstats(m)
        alpha = Wm.alpha if self.opts["ard_weights"] else torch.ones_like(Wm.alpha)
        self.be.mofa_update_w(B, Wm.tau.contiguous(), Gz.contiguous(), Z2.contiguous(), alpha,
                              Wm.lth, Wm.l1mth, self.opts["spikeslab_weights"], Wm.EW, Wm.EW2,
                              Wm.gamma, Wm.EWh2, Wm.sig2)

    def _view_product_a(self, m):
        """View m's share of the Z update: tau o W with its K x K statistics (one pass over the weight block per group)
        and A[m] = Y_m (tau o W_m)."""
        K, G, be, A = self.K, self.G, self.be, self._A
        V, Wm, rs_work = self.views[m], self.W[m], self._rs_work_v[m]
        for g, (a, b) in enumerate(self.gslice):
            kw = dict(wgt=Wm.tau[g], scale_out=True, gram=self._Gw[m, g], s2=self._dw2[m, g])
            if V.kind == "sparse":
                be.mofa_rowstats(Wm.EW, Wm.EW2, 0, V.D, rs_work, aux=V.mu[g], out_pad=V.TWs, col0=g * K,
                                 s1=self._corr[m, g], **kw)
            elif hasattr(V, "T16") and getattr(V, "implicit", False):
                be.mofa_rowstats(Wm.EW, Wm.EW2, 0, V.D, rs_work, aux=V.mu[g], out_pad=V.T16[g],
                                 s1=self._corr[m, g], **kw)  # (the centring term goes into the sweep: corr)
                A[m, a:b] = be.skinny_nn(V.Y[a:b], V.T16[g])[:, :K]
            elif hasattr(V, "T16"):
                be.mofa_rowstats(Wm.EW, Wm.EW2, 0, V.D, rs_work, out_pad=V.T16[g], **kw)
                A[m, a:b] = be.skinny_nn(V.Y[a:b], V.T16[g])[:, :K]
            else:
                # (f32: hipBLASLt streams Y at 5.1 TB/s when the K x D operand is the transposed one,
                #  scripts/probes/skinny_nn_probe.py: the kernel writes tau o W as K x D)
                be.mofa_rowstats(Wm.EW, Wm.EW2, 0, V.D, rs_work, out_t=V.TWt[g], **kw)
                torch.matmul(V.Y[a:b], V.TWt[g].T, out=A[m, a:b])
        if V.kind == "sparse":
            out = be.spmm(V.Xs, V.TWs)  # N x (G K); the centring term goes into the sweep (corr)
            if G == 1:
                A[m].copy_(out[:, :K])
            else:
                for g, (a, b) in enumerate(self.gslice):
                    A[m, a:b] = out[a:b, g * K:(g + 1) * K]

    def _update_z_fused(self):
        """W-side statistics in one pass per (view, group) over the weight block (mu_mofa_rowstats: tau o W
        as the dense operand, Gw, dw2 and the centring correction), the products A = Y (tau o W), the
        sample sweep."""
        K, G, M = self.K, self.G, self.M
        be, A = self.be, self._A
        self._fork()
        for m in range(M):
            with self._on(m):
                self._view_product_a(m)
        self._join()
        az = self.alpha_z if self.opts["ard_factors"] else torch.ones_like(self.alpha_z)
        be.mofa_update_z(A, self._pres, self.grp, self._Gw, self._dw2, az.contiguous(), self.EZ, self.EZ2,
                         self.sig2z, corr=self._corr)
        self._stats = {}  # <Z> changed: statistics are stale
        self._zmom = {}

    def _update_z(self):
        if getattr(self, "_fused", False):
            return self._update_z_fused()
        K, G, M, N = self.K, self.G, self.M, self.N
        dev = self.EZ.device
        A = torch.zeros((M, N, K), dtype=self.T, device=dev)
        Gw = torch.zeros((M, G, K, K), dtype=self.T, device=dev)
        dw2 = torch.zeros((M, G, K), dtype=self.T, device=dev)
        pres = torch.stack([v.pres for v in self.views]).contiguous()
        for m, (V, Wm) in enumerate(zip(self.views, self.W)):
            TW = Wm.tau[:, :, None] * Wm.EW[None, :, :]  # G x D x K
            for g in range(G):
                Gw[m, g] = self._tn(Wm.EW, TW[g])
                dw2[m, g] = (Wm.tau[g][:, None] * Wm.EW2).sum(dim=0)
            # (f32: hipBLASLt streams Y at 5.1 TB/s when the K x D operand is the transposed one -
            #  1.57 ms at 1e5 x 2e4 against 2.27 ms with tau o W stored D x K and 3.1 ms for
            #  mu_skinny_nn, scripts/probes/skinny_nn_probe.py; f64: rocBLAS needs 26 ms, mu_skinny_nn 4.6)
            if V.kind == "dense" and K <= 16 and self.T == torch.float64 and hasattr(self.be, "skinny_nn"):
                for g, (a, b) in enumerate(self.gslice):
                    T16 = torch.zeros((V.D, 16), dtype=self.T, device=dev)
                    T16[:, :K] = TW[g]
                    A[m, a:b] = self.be.skinny_nn(V.Y[a:b], T16)[:, :K]
            elif V.kind == "dense":
                for g, (a, b) in enumerate(self.gslice):
                    A[m, a:b] = V.Y[a:b] @ TW[g].T.contiguous().T
            else:
                Bp = _pad_block(G * K)
                TWs = torch.zeros((V.D, Bp), dtype=self.T, device=dev)
                for g in range(G):
                    TWs[:, g * K:(g + 1) * K] = TW[g]
                out = self.be.spmm(V.Xs, TWs)  # N x (G K)
                for g, (a, b) in enumerate(self.gslice):
                    # K: mu^T (tau o W), implicit centring (as a reduction: rocBLAS' f64 path takes
                    # 10.7 ms for this 1 x 1e5 by 1e5 x 10 product)
                    corr = (V.mu[g][:, None] * TW[g]).sum(dim=0)
                    A[m, a:b] = out[a:b, g * K:(g + 1) * K] - V.pres[a:b, None] * corr[None, :]
        az = self.alpha_z if self.opts["ard_factors"] else torch.ones_like(self.alpha_z)
        self.be.mofa_update_z(A.contiguous(), pres, self.grp, Gw.contiguous(), dw2.contiguous(),
                              az.contiguous(), self.EZ, self.EZ2, self.sig2z)
        self._stats = {}  # <Z> changed: statistics are stale

    def _update_rest_and_elbo(self):
        """tau, alpha_w, theta, alpha_z and the ELBO: csrc/mofa_elbo.hip (one pass per array and a
        one-workgroup finish per node; the same equations as tensor operations - ~250 launches per
        iteration - are tests/cpu_backend.py's versions of these entry points)."""
        o, be = self.opts, self.be
        elbo = torch.zeros((), dtype=torch.float64, device=self.EZ.device)
        work = self._elbo_work
        if self._par:
            # the factors' moments are shared between the views that observe every sample (and refresh the padded <Z>
            # operands of every product): made once, before the views part
            if getattr(self, "_fused", False):
                for m in range(self.M):
                    if self._stats.get(m) is None:
                        self._z_moments(m)
            parts = [elbo] + [torch.zeros((), dtype=torch.float64, device=self.EZ.device) for _ in range(self.M - 1)]
        else:
            parts = [elbo] * self.M
        self._fork()
        for m, (V, Wm) in enumerate(zip(self.views, self.W)):
            with self._on(m):
                wk = self._elbo_work_v[m] if self._par else work
                Gz, Z2, B = self._zstats(m)
                be.mofa_tau_elbo(V.yy, V.Ngm_d, Wm.EW, Wm.EW2, B, Gz, Z2, A0, B0, Wm.tau, Wm.ltau, parts[m], wk)
                be.mofa_w_elbo(Wm.EWh2, Wm.gamma, Wm.sig2, o["ard_weights"], o["spikeslab_weights"],
                               A0 + 0.5 * V.D, A0, B0, TH_A0, TH_B0, Wm.alpha, Wm.lalpha, Wm.lth, Wm.l1mth,
                               parts[m], wk)
        self._join()
        # factors: per-group column sums of <z^2> and ln sig2 over this rank's samples, added up over
        # the ranks in one collective; the ARD update and the ELBO terms follow from the global sums
        zs = self._zs
        for g, (a_, b_) in enumerate(self.gslice):
            be.mofa_z_sums(self.EZ2, self.sig2z, a_, b_, zs[g], work)
        if self.comm.world_size > 1:
            self.comm.all_reduce_sum(zs)
        be.mofa_z_elbo(zs, self._Ng64, o["ard_factors"], A0, B0, self.alpha_z, self.lalpha_z, elbo)
        if self._par:
            for extra in parts[1:]:
                elbo = elbo + extra  # (the views' terms, in view order)
        return elbo

    # -- several ranks: an iteration as TWO captured segments around ONE packed all-reduce (r06) -------------------------
    # With collectives inside, r02 - r05 launched a multi-rank iteration eagerly (~150 launches: 1.1 ms at the 12 500-cell
    # shard of configs[4], host-bound, against 4.2 ms / 8 = 0.5 ms of device work) and spent three collectives per
    # iteration (one per view's statistics, one for the factors' sums).  The iteration splits where the sums over the
    # ranks are needed: segment A = W updates | Z update | every view's LOCAL statistics | the factors' local sums, one
    # all-reduce of all of them (fixed buffers, packed into one message per dtype by the communicator), segment B = the
    # implicit centring | tau, alpha, theta per view | the factor node | the ELBO.  Each segment is a HIP graph; the host
    # replays A, issues the collective, replays B, reads the ELBO.  Same kernels on the same operands as the eager
    # iteration: the same numbers (tests/test_distributed_gloo.py runs the segments eagerly on CPU with two ranks,
    # tests/test_gpu_mofa.py the captured ones against the eager trace).
    def _seg_alloc(self):
        """The fixed buffers of the segmented iteration as ONE flat block: the factors' moments (one set per distinct
        presence mask), every view's B, and - in an f64 fit - the factors' sums.  What a rank contributes to the sums over
        the ranks is then one contiguous tensor: the collective needs no packing copy and no copy back."""
        K, G, T, dev = self.K, self.G, self.T, self.EZ.device
        keys = []
        for m, V in enumerate(self.views):
            key = "all" if V.full else m
            if key not in keys:
                keys.append(key)
        n_mom = G * K * K + 2 * G * K
        n_b = [G * V.D * K for V in self.views]
        n_zs = G * 2 * K if T == torch.float64 else 0
        flat = torch.zeros((len(keys) * n_mom + sum(n_b) + n_zs,), dtype=T, device=dev)
        off = 0
        self._zmom_out = {}
        for key in keys:
            Gz = flat[off:off + G * K * K].view(G, K, K)
            Z2 = flat[off + G * K * K:off + G * K * K + G * K].view(G, K)
            Zs = flat[off + G * K * K + G * K:off + n_mom].view(G, K)
            self._zmom_out[key] = (Gz, Z2, Zs)
            off += n_mom
        self._locB = []
        for V, nb in zip(self.views, n_b):
            self._locB.append(flat[off:off + nb].view(G, V.D, K))
            off += nb
        if n_zs:
            self._zs = flat[off:off + n_zs].view(G, 2, K)  # (f64 fit: the factors' sums travel in the same message)
        self._loc_flat = flat
        # B with the implicit centring applied (sparse / implicitly centred views); the others read their B in place
        self._statB = [torch.empty((G, V.D, K), dtype=T, device=dev)
                       if (V.kind == "sparse" or getattr(V, "implicit", False)) else None for V in self.views]
        self._zmom = {}  # (the next moments are written into the fixed tensors)

    def _seg_a(self):
        if self.__dict__.get("_loc_flat") is None:
            self._seg_alloc()
        self._fork()
        for m in range(self.M):
            with self._on(m):
                self._update_w(m)
        self._join()
        self._update_z()
        if self._par:  # the factors' moments and the padded <Z> operands are shared by the views: before they part
            for m in range(self.M):
                self._z_moments(m)
        self._fork()
        for m in range(self.M):
            with self._on(m):
                self._locB[m].copy_(self._stats_local(m)[3])  # (the moments are in the flat block already)
        self._join()
        for g, (a_, b_) in enumerate(self.gslice):
            self.be.mofa_z_sums(self.EZ2, self.sig2z, a_, b_, self._zs[g], self._elbo_work)

    def _seg_reduce(self):
        if self._zs.dtype == self._loc_flat.dtype:
            self.comm.all_reduce_sum(self._loc_flat)  # (an f64 fit: one message, the factors' sums inside)
        else:
            self.comm.all_reduce_sum(self._loc_flat)
            self.comm.all_reduce_sum(self._zs)

    def _seg_b(self) -> torch.Tensor:
        o, be = self.opts, self.be
        elbo = torch.zeros((), dtype=torch.float64, device=self.EZ.device)
        parts = [elbo] + ([torch.zeros((), dtype=torch.float64, device=self.EZ.device) for _ in range(self.M - 1)]
                          if self._par else [elbo] * (self.M - 1))
        self._fork()
        for m, (V, Wm) in enumerate(zip(self.views, self.W)):
            with self._on(m):
                wk = self._elbo_work_v[m] if self._par else self._elbo_work
                Gz, Z2, Zs = self._zmom_out["all" if V.full else m]
                B = self._locB[m]
                if self._statB[m] is not None:
                    # implicit centring with the sums over all ranks: B - mu (x) Zs in one kernel
                    B = torch.addcmul(B, V.mu[:, :, None], Zs[:, None, :], value=-1.0, out=self._statB[m])
                self._stats[m] = (Gz, Z2, B)
                be.mofa_tau_elbo(V.yy, V.Ngm_d, Wm.EW, Wm.EW2, B, Gz, Z2, A0, B0, Wm.tau, Wm.ltau, parts[m], wk)
                be.mofa_w_elbo(Wm.EWh2, Wm.gamma, Wm.sig2, o["ard_weights"], o["spikeslab_weights"], A0 + 0.5 * V.D, A0,
                               B0, TH_A0, TH_B0, Wm.alpha, Wm.lalpha, Wm.lth, Wm.l1mth, parts[m], wk)
        self._join()
        be.mofa_z_elbo(self._zs, self._Ng64, o["ard_factors"], A0, B0, self.alpha_z, self.lalpha_z, elbo)
        if self._par:
            for extra in parts[1:]:
                elbo = elbo + extra  # (the views' terms, in view order)
        return elbo

    def _iteration_segments(self) -> torch.Tensor:
        self._seg_a()
        self._seg_reduce()
        return self._seg_b()

    def _capture_segments(self):
        dev = self.be.device
        torch.cuda.synchronize(dev)
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(ga):
                self._seg_a()
            ga.replay()  # (capturing records, it does not run: this replay IS the iteration's first half)
            self._seg_reduce()
            with torch.cuda.graph(gb):
                out = self._seg_b()
            gb.replay()
        except Exception as e:  # capture refused: stay eager (the state may be half an iteration ahead: finish it)
            warnings.warn(f"MOFA iteration not captured into HIP graph segments ({e}); running eagerly")
            self._seg_ok = False
            self._stats = {}
            return None
        self._seg_graphs, self._graph_elbo = (ga, gb), out
        return out

    # -- driver --------------------------------------------------------------------------------
    def _iteration(self) -> torch.Tensor:
        """One coordinate-ascent sweep (W per view, Z, tau / alpha / theta, ELBO); device work only,
        returns the ELBO as a device scalar."""
        if self._par and getattr(self, "_fused", False):
            #  the factors' moments and the padded <Z> operands are SHARED by the views: when no statistics
            # are cached (first eager iteration, the one after a refused capture) view 0's W update would compute them
            # on the main stream after the fork while view 1's reads them on its side stream - made here, before the
            # views part, exactly as _update_rest_and_elbo does
            for m in range(self.M):
                if self._stats.get(m) is None:
                    self._z_moments(m)
        self._fork()
        for m in range(self.M):
            with self._on(m):
                self._update_w(m)
        self._join()
        self._update_z()
        return self._update_rest_and_elbo()

    def _capture(self):
        # the statistics the first W update reads must already sit in their fixed buffers
        for m in range(self.M):
            self._zstats(m)
        torch.cuda.synchronize(self.be.device)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                out = self._iteration()
        except Exception as e:  # capture refused (e.g. a library call that allocates): stay eager
            warnings.warn(f"MOFA iteration not captured into a HIP graph ({e}); running eagerly")
            self._graph_ok = False
            self._stats = {}
            return
        self._graph, self._graph_elbo = g, out

    def step(self):
        if self._graph is None and self._graph_ok and self._eager_steps >= 2:
            self._capture()
        if self._graph is not None:
            self._graph.replay()
            e = float(self._graph_elbo.item())
        elif self._seg_graphs is not None:
            self._seg_graphs[0].replay()
            self._seg_reduce()
            self._seg_graphs[1].replay()
            e = float(self._graph_elbo.item())
        elif self._seg and self._eager_steps >= 1:
            self._par = self._par_ok  # (no collective inside a segment: the views' shares on their own streams)
            out = None
            if self._seg_ok and self._eager_steps >= 3:
                out = self._capture_segments()  # (runs the iteration it captures)
            if out is None:
                out = self._iteration_segments()
            e = float(out.item())
            self._eager_steps += 1
        else:
            e = float(self._iteration().item())
            self._eager_steps += 1
        self.elbo.append(e)
        return e

    def run(self, n_iterations=1000, convergence_mode="fast", min_iterations=2, callback=None):
        tol = TOL[convergence_mode]
        for it in range(n_iterations):
            self.step()
            if callback is not None:
                callback(it, self)
            if it >= min_iterations and len(self.elbo) >= 2:
                # (rank 0 decides: a rank that leaves alone strands the others in their next collective)
                if self.comm.agree(100.0 * abs((self.elbo[-1] - self.elbo[-2]) / self.elbo[0]) < tol):
                    break
        return len(self.elbo)

    def variance_explained(self):
        """R2 (%) of every factor alone per (view, group): 1 - SS(y - z_k w_k^T) / SS(y), from the
        same statistics (no extra pass over the data beyond B)."""
        K, G = self.K, self.G
        r2 = torch.zeros((self.M, G, K), dtype=torch.float64)
        for m, (V, Wm) in enumerate(zip(self.views, self.W)):
            Gz, Z2, B = self._zstats(m)
            for g in range(G):
                ss = V.yy[g].sum()
                for k in range(K):
                    w = Wm.EW[:, k]
                    res = ss - 2.0 * (w * B[g][:, k]).sum() + (w * w).sum() * Gz[g][k, k]
                    r2[m, g, k] = float((100.0 * (1.0 - res / ss)).item()) if float(ss) > 0 else 0.0
        return r2.numpy()

    def results(self, sort_factors=True):
        """Factors / weights on the host in the caller's sample order."""
        inv = np.empty_like(self.perm)
        inv[self.perm] = np.arange(self.N)
        Z = self.be.to_host(self.EZ)[inv].astype(np.float64)
        W = [self.be.to_host(w.EW).astype(np.float64) for w in self.W]
        r2 = self.variance_explained()
        order = np.arange(self.K)
        if sort_factors:
            order = np.argsort(-r2.sum(axis=(0, 1)), kind="stable")
        return {"Z": Z[:, order], "W": [w[:, order] for w in W], "r2": r2[:, :, order],
                "elbo": list(self.elbo), "order": order,
                "intercepts": [self.be.to_host(v.intercepts) for v in self.views]}
