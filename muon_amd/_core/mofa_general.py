"""MOFA+ with non-gaussian likelihoods and element-wise missing values.

The reference reaches this through ``likelihoods`` (/root/reference/muon/_core/tools.py:296, guessed
by mofapy2's ``guess_likelihoods`` when None, :272-280) and through NaN entries of a modality
(:144-169); the arithmetic is mofapy2's pseudo-data nodes (Seeger bound for poisson counts, Jaakkola
bound for bernoulli data), which need the dense N x D prediction in every iteration - the reference
densifies every modality for it (:117-141).

Here every view is a gaussian model on pseudo-data with an ELEMENT-WISE precision (equations:
oracle/mofa_oracle.py ``run_general``; same schedule and initialisation, so engine and oracle agree
iteration by iteration), which breaks the per-feature-tau sufficient statistics of ``MofaEngine``:

    W update of view m needs   T_d = sum_n Omega_nd <z_n z_n^T>   (D x K x K)  and  b = R^T <Z>
    Z update needs             S_n = sum_d Omega_nd <w_d w_d^T>   (N x K x K)  and  a = R <W>

Nothing of size N x D is ever stored: the samples are walked in row chunks (``chunk_elems`` dense
elements at a time); a sparse modality stays CSR in HBM and only the chunk in flight is densified -
zeros are data for a count likelihood.  Per chunk the work is dense GEMMs against K- and K^2-column
blocks plus element-wise transforms: PyTorch-ROCm tensor operations (north_star: "PyTorch-ROCm only
for the MOFA dense factor blocks").  Three chunk passes per iteration (W statistics, Z update,
tau / ELBO).  Gaussian models without missing entries keep the two-pass HIP engine (mofa_engine.py).
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import numpy as np
import torch
from scipy.sparse import issparse

from .._comm import default_comm
from .mofa_engine import A0, B0, TH_A0, TH_B0, TOL

LIKELIHOODS = ("gaussian", "poisson", "bernoulli")


class _GView:
    pass


def _lambda_jj(x: torch.Tensor) -> torch.Tensor:
    x = x.abs().clamp(min=1e-8)
    return torch.tanh(0.5 * x) / (4.0 * x)


def _gamma_kl(a0, b0, a, b, ex, elx):
    lp = a0 * math.log(b0) - math.lgamma(a0) + (a0 - 1.0) * elx - b0 * ex
    lq = a * torch.log(b) - torch.lgamma(a) + (a - 1.0) * elx - b * ex
    return lp - lq


def _beta_kl(a0, b0, a, b, elx, el1mx):
    lb = lambda p, q: torch.lgamma(p) + torch.lgamma(q) - torch.lgamma(p + q)  # noqa: E731
    lb0 = math.lgamma(a0) + math.lgamma(b0) - math.lgamma(a0 + b0)
    return (lb(a, b) - lb0) + (a0 - a) * elx + (b0 - b) * el1mx


class GeneralMofaEngine:
    """Same driver interface as MofaEngine (step / run / results / elbo)."""

    def __init__(self, backend, views: List, likelihoods: List[str], groups: np.ndarray, n_factors: int, *,
                 dtype=torch.float64, center_groups=True, scale_views=False, scale_groups=False,
                 ard_weights=True, ard_factors=True, spikeslab_weights=True, seed=1, comm=None,
                 row_offset: int = 0, n_total: Optional[int] = None, chunk_elems: int = 1 << 27,
                 spikeslab_factors: bool = False):
        assert len(likelihoods) == len(views) and set(likelihoods).issubset(LIKELIHOODS)
        self.be = backend
        self.comm = default_comm(comm)
        self.T = dtype
        self.K = K = int(n_factors)
        self.lik = list(likelihoods)
        # spikeslab_factors (/root/reference/muon/_core/tools.py:305,486): the W node's spike-and-slab update with samples in
        # the place of features, one (alpha, theta) pair per (group, factor) - oracle/mofa_oracle.py run_general
        self.opts = dict(ard_weights=ard_weights, ard_factors=ard_factors, spikeslab_weights=spikeslab_weights,
                         spikeslab_factors=bool(spikeslab_factors))
        groups = np.asarray(groups, dtype=np.int64)
        self.N = N = len(groups)
        gmax = int(groups.max()) if groups.size else 0
        if self.comm.world_size > 1:
            gmax = int(self._allreduce_max(torch.tensor([gmax], dtype=torch.int64)).item())
        self.G = G = gmax + 1
        self.perm = np.argsort(groups, kind="stable")
        gs = groups[self.perm]
        self.gslice = [(int(np.searchsorted(gs, g, "left")), int(np.searchsorted(gs, g, "right"))) for g in range(G)]
        self.Ng = self._allreduce(torch.tensor([b - a for a, b in self.gslice], dtype=torch.float64))
        self.M = len(views)
        self.chunk_elems = int(chunk_elems)
        self.views = [self._prepare_view(v, lk, center_groups, scale_views, scale_groups)
                      for v, lk in zip(views, self.lik)]
        self.Ds = [v.D for v in self.views]
        self.dev = self.views[0].dev
        # initialisation shared with the oracle (oracle/mofa_oracle.py init_state)
        n_total = N if n_total is None else int(n_total)
        z0 = np.random.default_rng(seed).standard_normal((n_total, K))[row_offset:row_offset + N]
        self.EZ = backend.to_device(np.ascontiguousarray(z0[self.perm])).to(dtype)
        self.EZ2 = self.EZ ** 2 + 1.0
        self.sig2z = torch.ones_like(self.EZ)
        c = float(torch.digamma(torch.tensor(1.0, dtype=torch.float64)) - torch.digamma(torch.tensor(2.0, dtype=torch.float64)))
        # the alpha / theta / factor-ARD nodes and their ELBO terms: the fused kernels of MofaEngine (csrc/mofa_elbo.hip -
        # the same equations; ~170 tensor launches per iteration otherwise) keep these nodes in the fit's type
        # (the fused factor nodes have no spike: with spikeslab_factors the small nodes run as tensor operations)
        self._fused_small = (hasattr(backend, "mofa_w_elbo") and hasattr(backend, "mofa_z_elbo") and K <= 32
                             and not spikeslab_factors)
        ST = dtype if self._fused_small else torch.float64
        self.W = []
        for D in self.Ds:
            w = _GView()
            w.EW = torch.zeros((D, K), dtype=dtype, device=self.dev)
            w.EW2 = torch.ones((D, K), dtype=dtype, device=self.dev)
            w.gamma = torch.ones((D, K), dtype=dtype, device=self.dev)
            w.EWh2 = torch.ones((D, K), dtype=dtype, device=self.dev)
            w.sig2 = torch.ones((D, K), dtype=dtype, device=self.dev)
            w.tau = torch.ones((G, D), dtype=dtype, device=self.dev)
            w.ltau = torch.zeros((G, D), dtype=dtype, device=self.dev)
            w.alpha = torch.ones((K,), dtype=ST, device=self.dev)
            w.lalpha = torch.zeros((K,), dtype=ST, device=self.dev)
            w.lth = torch.full((K,), c, dtype=ST, device=self.dev)
            w.l1mth = torch.full((K,), c, dtype=ST, device=self.dev)
            self.W.append(w)
        self.alpha_z = torch.ones((G, K), dtype=ST, device=self.dev)
        self.lalpha_z = torch.zeros((G, K), dtype=ST, device=self.dev)
        self.gamma_z = torch.ones_like(self.EZ)
        self.EZh2 = self.EZ2.clone()
        self.lthz = torch.full((G, K), c, dtype=torch.float64, device=self.dev)
        self.l1mthz = torch.full((G, K), c, dtype=torch.float64, device=self.dev)
        self._elbo_work = backend.mofa_elbo_work(K) if hasattr(backend, "mofa_elbo_work") and K <= 32 else None
        if self._fused_small:
            self._zs = torch.zeros((G, 2, K), dtype=torch.float64, device=self.dev)
        self.elbo = []
        self._Ng_dev = self.Ng.to(self.dev)  # (resident: an iteration has no host -> device copies)
        # One iteration is ~400 short launches (chunk passes of the masked / bernoulli views, the K x K algebra of the
        # tau / alpha / theta / ELBO terms).  Every expectation is updated in place (r04 rebound alpha / theta), so the
        # iteration CAN be captured into a HIP graph and replayed - MUON_AMD_MOFA_NG_GRAPH=1, bit-identical to eager
        # launches (tests/test_gpu_mofa.py) - but it is off by default: measured at 20 000 x 22 000 the replay takes
        # 5.29 ms against 5.33 ms eager (scripts/probes/mofa_ng_graph_probe.py) and the capture costs 16 ms once.  The
        # 3 ms next to the poisson passes' 2 ms are not host launches: ~400 kernels of 2-20 us with the device's own
        # dispatch gap between them, replayed or not.  Fewer kernels (the masked gaussian view's chunk passes and the
        # tau / alpha / theta algebra fused as in MofaEngine) is what would move it.
        self._graph = None
        self._graph_elbo = None
        self._graph_ok = (getattr(backend, "name", "") == "hip" and self.comm.world_size == 1
                          and os.environ.get("MUON_AMD_MOFA_NG_GRAPH", "0") == "1")
        self._eager_steps = 0
        self._zver = 0      # state counter of the factors (the cached statistics of _gauss_stats belong to one state)
        self._gstats = {}
        self._wver = [0] * self.M  # ... and of every view's weights: b = R^T <Z> of a fused poisson view made by the ELBO
        self._pois_pads = {}       # padded factor blocks of the poisson passes (HipBackend.mofa_poisson_pass)
        self._bnext = {}           # pass (mode 3 of mofa_poisson_pass) serves the next W update if neither has changed

    # -- collectives ------------------------------------------------------------------------------
    def _on_comm_device(self, t):
        if getattr(self.be, "name", "") == "hip" and not t.is_cuda:
            return t.to(self.be.device)
        return t

    def _allreduce(self, *ts):
        if self.comm.world_size > 1:
            moved = [self._on_comm_device(t) for t in ts]
            self.comm.all_reduce_sum(*moved)
            for t, m in zip(ts, moved):
                if m is not t:
                    t.copy_(m)
        return ts[0] if len(ts) == 1 else ts

    def _allreduce_max(self, t):
        if self.comm.world_size > 1:
            m = self._on_comm_device(t)
            self.comm.all_reduce_max(m)
            t = m.to(t.device)
        return t

    # -- data ---------------------------------------------------------------------------------------
    def _rows_per_chunk(self, D):
        return max(64, self.chunk_elems // max(D, 1))

    def _prepare_view(self, v, lik, center_groups, scale_views, scale_groups):
        be, T, G, N = self.be, self.T, self.G, self.N
        V = _GView()
        V.lik = lik
        V.D = D = v.shape[1]
        if issparse(v):
            m = v.tocsr()[self.perm]
            m.sort_indices()
            if np.isnan(m.data).any():
                raise NotImplementedError("NaN among the stored entries of a sparse modality: densify it "
                                          "(stored zeros of a sparse matrix are observations)")
            V.kind = "sparse"
            V.X = be.upload_csr(m.indptr, m.indices, m.data.astype(np.float64), m.shape)
            V.X = V.X.with_values(V.X.values.to(T))
            pres = np.ones(N, dtype=bool)
            if getattr(v, "_missing_rows", None) is not None:
                pres = ~np.asarray(v._missing_rows)[self.perm]
            V.rowmask = be.to_device(pres.astype(np.float64)).to(T)
            V.mask = None
            V.dev = V.X.values.device
        else:
            a = np.asarray(v, dtype=np.float64)[self.perm]
            nan = np.isnan(a)
            V.kind = "dense"
            V.Y = be.to_device(np.where(nan, 0.0, a)).to(T)
            V.mask = be.to_device((~nan).astype(np.float64)).to(T) if nan.any() else None
            V.rowmask = None
            V.dev = V.Y.device
        # first / second moments per (group, feature) over the observed entries
        s1 = torch.zeros((G, D), dtype=torch.float64, device=V.dev)
        s2 = torch.zeros((G, D), dtype=torch.float64, device=V.dev)
        cnt = torch.zeros((G, D), dtype=torch.float64, device=V.dev)
        mx = torch.zeros((D,), dtype=torch.float64, device=V.dev)
        V.mu = torch.zeros((G, D), dtype=T, device=V.dev)
        V.scale = torch.ones((G,), dtype=T, device=V.dev)
        for g, (a0, b0) in enumerate(self.gslice):
            for lo, hi, Yc, Mc in self._chunks(V, a0, b0, raw=True):
                Yd = Yc.double()
                Md = Mc.double() if Mc is not None else None
                s1[g] += (Yd * Md).sum(dim=0) if Md is not None else Yd.sum(dim=0)
                s2[g] += (Yd * Yd * Md).sum(dim=0) if Md is not None else (Yd * Yd).sum(dim=0)
                cnt[g] += Md.sum(dim=0) if Md is not None else float(hi - lo)
                if hi > lo:
                    mx = torch.maximum(mx, (Yd * Md if Md is not None else Yd).max(dim=0).values)
        s1, s2, cnt = self._allreduce(s1, s2, cnt)
        n = cnt.clamp(min=1.0)
        V.intercepts = (s1 / n).to(T)  # tools.py:283-286: nanmean per (view, group), whatever the likelihood
        V.kappa = None
        if lik == "poisson":
            V.kappa = (0.25 + 0.17 * self._allreduce_max(mx)).to(T)
        if lik == "gaussian":
            mu = s1 / n
            if not center_groups:
                mu = (s1.sum(dim=0) / cnt.sum(dim=0).clamp(min=1.0))[None, :].expand(G, D).contiguous()
            c1 = s1 - cnt * mu
            yy = s2 - 2 * mu * s1 + cnt * mu * mu
            scale = torch.ones((G,), dtype=torch.float64, device=V.dev)
            if scale_groups:
                for g in range(G):
                    c = float(cnt[g].sum())
                    if c > 0:
                        var = float(yy[g].sum()) / c - (float(c1[g].sum()) / c) ** 2
                        if var > 0:
                            scale[g] = 1.0 / math.sqrt(var)
            elif scale_views:
                c = float(cnt.sum())
                if c > 0:
                    var = float(yy.sum()) / c - (float(c1.sum()) / c) ** 2
                    if var > 0:
                        scale = scale / math.sqrt(var)
            V.mu = mu.to(T)
            V.scale = scale.to(T)
        # a poisson view stored sparse with every sample present never needs a dense chunk: for y = 0 the pseudo-data
        # depend on (z_n, w_d) only - dense sweeps over the two factor blocks + corrections over the stored entries
        # (csrc/mofa_poisson.hip, r04)
        V.fused = bool(lik == "poisson" and V.kind == "sparse" and hasattr(be, "mofa_poisson_pass") and self.K <= 32
                       and V.X.values.dtype == T and pres.all())
        # a bernoulli view stored sparse with every sample present needs none either (r06, csrc/mofa_bernoulli.hip): the
        # data enter through R = y - 1/2 and the likelihood only - sparse products - and the Jaakkola precision depends on
        # the two factor blocks alone: one dense sweep per update
        V.fusedb = bool(lik == "bernoulli" and V.kind == "sparse" and hasattr(be, "mofa_jaakkola_sweep")
                        and hasattr(be, "mofa_softplus_sweep") and self.K <= 16 and V.X.values.dtype == T and pres.all())
        if V.fused or V.fusedb:
            V.Xt = be.transpose(V.X)
        # the three sparse products of a fused bernoulli view multiply by one 16-column block each: the sliced-ELL layout
        # of MofaEngine's sparse views (csrc/spmm_ell.hip: 0.24 -> ~0.08 ms per product at 3e7 entries), laid out once
        V.Xe = V.Xte = None
        if V.fusedb and hasattr(be, "ell16"):
            from .mofa_engine import _can_ell16

            wide = T == torch.float64
            if min(V.X.shape) > 0 and _can_ell16(be, V.X, wide):
                # (f64 values go in as hi + lo parts, the second only when some value is not exact in f32)
                V.Xe, V.Xte = be.ell16(V.X, wide=wide), be.ell16(V.Xt, wide=wide)
        V.centred = False
        V.stats = False
        if lik == "gaussian" and V.kind == "dense":
            # the centred / scaled / masked values are a constant of the fit: made once, in place (r04 recomputed them in
            # every chunk pass - three tensor passes over the view, six times per iteration).  Same operations in the
            # same order as _chunks: the same values.
            for g, (a0, b0) in enumerate(self.gslice):
                if b0 > a0:
                    blk = V.Y[a0:b0]
                    blk.sub_(V.mu[g][None, :]).mul_(V.scale[g])
                    if V.mask is not None:
                        blk.mul_(V.mask[a0:b0])
            V.centred = True
            # ... and with them the view needs nothing of size N x D per pass: Omega_nd = tau_gd M_nd, so
            #   T_d = tau_gd (M^T P)_d,  b = tau_gd (Y^T <Z>)_d,  S_n = M_n (tau o <w w^T>),  a_n = Y_n (tau o <W>),
            #   sum_n M_nd <(y - z w)^2> = sum_n y^2 - 2 <w_d> . (Y^T <Z>)_d + <w_d w_d^T> : (M^T P)_d
            # (P_n = <z_n z_n^T>): products of the two constant matrices Y (centred, masked) and M with K- and
            # K^2-column blocks - the sufficient statistics of MofaEngine with the mask inside (r05; r04 made Omega, R,
            # the prediction and the residuals as N x D tensors in every pass: ~25 kernels per iteration).
            V.stats = True
            if V.stats:
                V.yyM = torch.stack([(V.Y[a0:b0].double() ** 2).sum(dim=0) for a0, b0 in self.gslice])
                V.Ngd = (torch.stack([V.mask[a0:b0].double().sum(dim=0) for a0, b0 in self.gslice]) if V.mask is not None
                         else torch.tensor([[float(b0 - a0)] for a0, b0 in self.gslice], dtype=torch.float64,
                                           device=V.dev).expand(G, D).contiguous())
        return V

    def _chunks(self, V, a, b, raw=False):
        """Row chunks [lo, hi) of the samples a..b of a view: (lo, hi, Y, M) with Y the dense chunk
        (centred / scaled unless ``raw``; unobserved entries 0) and M its 0/1 mask (None: all observed)."""
        step = self._rows_per_chunk(V.D)
        for lo in range(a, b, step):
            hi = min(b, lo + step)
            if V.kind == "dense":
                Y = V.Y[lo:hi]
                M = V.mask[lo:hi] if V.mask is not None else None
            else:
                X = V.X
                if hasattr(self.be, "densify_rows") and X.values.dtype == self.T:
                    Y = self.be.densify_rows(X, lo, hi)
                    p0 = p1 = 0
                else:
                    p0, p1 = int(X.indptr[lo].item()), int(X.indptr[hi].item())
                    Y = torch.zeros((hi - lo, V.D), dtype=self.T, device=V.dev)
                if p1 > p0:
                    rows = torch.repeat_interleave(torch.arange(hi - lo, device=V.dev),
                                                   (X.indptr[lo + 1:hi + 1] - X.indptr[lo:hi]))
                    Y[rows, X.indices[p0:p1].long()] = X.values[p0:p1]
                rm = V.rowmask[lo:hi]
                # (whether a chunk has every sample is a property of the fit, not of the iteration: asked once -
                #  a device -> host question per chunk and pass otherwise: 16.5 -> 14.8 ms per iteration at 20k x 22k)
                full = V.__dict__.setdefault("_chunk_full", {})
                if (lo, hi) not in full:
                    full[(lo, hi)] = bool((rm == 1).all())
                M = None if full[(lo, hi)] else rm[:, None].expand(hi - lo, V.D)
            if not raw and V.lik == "gaussian" and not V.centred:
                g = self._group_of(lo)
                Y = (Y - V.mu[g][None, :]) * V.scale[g]
                if M is not None:
                    Y = Y * M
            yield lo, hi, Y, M

    def _group_of(self, row):
        for g, (a, b) in enumerate(self.gslice):
            if a <= row < b:
                return g
        raise IndexError(row)

    def _omega_r(self, V, Wm, g, Y, M, Zc, Z2c):
        """(Omega, R, zeta, omega_vec) of a chunk: the element-wise precision, precision x pseudo-data, prediction.
        Where the precision does not depend on the sample (gaussian / poisson without missing entries) Omega is
        None and ``omega_vec`` [D] says it all: the K x K statistics then factorise (sum_n Omega_nd <z z^T> =
        omega_d sum_n <z z^T>) and nothing of size N x D x K^2 is multiplied."""
        if V.lik == "gaussian":  # (no caller needs the prediction of a gaussian chunk: its N x D x K product is not made)
            if M is None:
                tau = Wm.tau[g]
                return None, tau[None, :] * Y, None, tau
            Om = Wm.tau[g][None, :] * M
            return Om, Om * Y, None, None
        zeta = Zc @ Wm.EW.T
        if V.lik == "poisson":
            if hasattr(self.be, "mofa_poisson_pseudo") and Y.is_contiguous():
                R = self.be.mofa_poisson_pseudo(zeta, Y, V.kappa.contiguous(), 0)  # one pass, in place of zeta
                zeta = None  # (no caller needs the prediction of a poisson chunk)
            else:
                rate = torch.nn.functional.softplus(zeta).clamp(min=1e-300 if self.T == torch.float64 else 1e-30)
                R = V.kappa[None, :] * zeta - torch.sigmoid(zeta) * (1.0 - Y / rate)
            if M is not None:
                return V.kappa[None, :] * M, R * M, zeta, None
            return None, R, zeta, V.kappa
        if hasattr(self.be, "mofa_jaakkola"):
            Om = self.be.mofa_jaakkola(zeta, Z2c @ Wm.EW2.T, (Zc ** 2) @ (Wm.EW ** 2).T)  # one pass
        else:
            xi2 = zeta ** 2 + Z2c @ Wm.EW2.T - (Zc ** 2) @ (Wm.EW ** 2).T
            Om = 2.0 * _lambda_jj(torch.sqrt(xi2.clamp(min=0.0)))
        R = Y - 0.5
        if M is not None:
            Om, R = Om * M, R * M
        return Om, R, zeta, None

    @staticmethod
    def _outer_moments(E, E2):
        """rows -> K^2 columns: <e e^T> with the diagonal replaced by the second moments."""
        n, K = E.shape
        P = E[:, :, None] * E[:, None, :]
        P.diagonal(dim1=1, dim2=2).copy_(E2)  # (one strided copy: no index tensor, no gather / scatter kernels)
        return P.reshape(n, K * K)

    def _times_block(self, X, E):
        """X E for a CSR operand and a K <= 16 column block: the row-wave SpMM multiplies by 16 columns"""
        P = torch.zeros((E.shape[0], 16), dtype=E.dtype, device=E.device)
        P[:, :self.K] = E
        return self.be.spmm(X, P)[:, :self.K]

    def _z_outer(self):
        """<z z^T> rows of ALL local samples [N, K^2] for the current factors: the W update of a fused poisson view (its
        column sums) and the statistics of every dense gaussian view (rows times the mask) read the same block"""
        hit = getattr(self, "_zouter", None)
        if hit is None or hit[0] != self._zver:
            if hit is None or not self._graph_ok:
                hit = self._zouter = [self._zver, self._outer_moments(self.EZ, self.EZ2)]
            else:  # (a fixed buffer, written in place: see _gauss_stats)
                P = hit[1].view(-1, self.K, self.K)
                torch.mul(self.EZ[:, :, None], self.EZ[:, None, :], out=P)
                P.diagonal(dim1=1, dim2=2).copy_(self.EZ2)
                hit[0] = self._zver
        return hit[1]

    def _gauss_stats(self, m):
        """(B, Q) of a dense gaussian view for the CURRENT factors: B[g] = Y_g^T <Z_g> [D, K], Q[g] = M_g^T P_g [D, K^2]
        (P = <z z^T> rows; without a mask every row of Q[g] is sum_n P_n).  Made once per state of the factors: the
        tau / ELBO pass of an iteration and the W update of the next read the same ones."""
        V = self.views[m]
        hit = self._gstats.get(m)
        if hit is None:  # (fixed buffers, written in place: a captured iteration finds the previous replay's statistics)
            Kk = self.K * self.K
            hit = self._gstats[m] = [-1, [torch.empty((V.D, self.K), dtype=self.T, device=self.dev) for _ in self.gslice],
                                     [torch.empty((V.D if V.mask is not None else 1, Kk), dtype=self.T, device=self.dev)
                                      for _ in self.gslice]]
        if hit[0] == self._zver:
            return hit[1], hit[2]
        for g, (a0, b0) in enumerate(self.gslice):
            Zg = self.EZ[a0:b0]
            P = self._z_outer()[a0:b0]
            torch.matmul(V.Y[a0:b0].T, Zg, out=hit[1][g])
            if V.mask is not None:
                torch.matmul(V.mask[a0:b0].T, P, out=hit[2][g])
            else:
                hit[2][g].copy_(P.sum(dim=0)[None, :])
        hit[0] = self._zver
        return hit[1], hit[2]

    # -- one coordinate-ascent sweep -------------------------------------------------------------------
    def _update_w(self, m):
        V, Wm, K = self.views[m], self.W[m], self.K
        # (the statistics are built from their first term, not added to zeros: an iteration of the poisson benchmark is
        #  ~130 small tensor kernels at ~5 us each next to 1 ms of passes - every fill and "+=" that is not needed counts)
        if getattr(V, "fused", False):
            # Omega does not depend on the sample: T_d = kappa_d sum_n <z_n z_n^T>; b = R^T <Z> without R
            Tm = V.kappa[:, None] * self._z_outer().sum(dim=0)[None, :]
            hit = self._bnext.get(m)
            if hit is not None and hit[0] == (self._zver, self._wver[m]):
                # (made by the tau / ELBO pass of the iteration before, in the same sweep as its likelihood term; only
                #  read below, unless the ranks add theirs up in place)
                b = hit[1].clone() if getattr(self.comm, "world_size", 1) > 1 else hit[1]
            else:
                b = self.be.mofa_poisson_pass(1, Wm.EW.contiguous(), self.EZ.contiguous(), V.kappa.contiguous(), V.Xt, pads=self._pois_pads)
        elif getattr(V, "fusedb", False):
            # T_d = sum_n Omega_nd <z z^T>_n in one sweep over the factor blocks; b = (Y - 1/2)^T <Z> without Y dense
            Tm = self.be.mofa_jaakkola_sweep(Wm.EW, Wm.EW2, self.EZ, self.EZ2).reshape(V.D, K * K)
            b = self._times_block(V.Xte if V.Xte is not None else V.Xt, self.EZ) - 0.5 * self.EZ.sum(dim=0)[None, :]
        elif V.stats:
            Bs, Qs = self._gauss_stats(m)
            Tm = b = None
            for g in range(self.G):
                tau = Wm.tau[g][:, None]
                Tm = tau * Qs[g] if Tm is None else Tm.addcmul_(tau, Qs[g])
                b = tau * Bs[g] if b is None else b.addcmul_(tau, Bs[g])
        else:
            Tm = torch.zeros((V.D, K * K), dtype=self.T, device=self.dev)
            b = torch.zeros((V.D, K), dtype=self.T, device=self.dev)
            for g, (a0, b0) in enumerate(self.gslice):
                for lo, hi, Y, M in self._chunks(V, a0, b0):
                    Zc, Z2c = self.EZ[lo:hi], self.EZ2[lo:hi]
                    Om, R, _, om_vec = self._omega_r(V, Wm, g, Y, M, Zc, Z2c)
                    P = self._outer_moments(Zc, Z2c)
                    if Om is None:
                        Tm += om_vec[:, None] * P.sum(dim=0)[None, :]
                    else:
                        Tm += Om.T @ P
                    b += R.T @ Zc
        Tm, b = self._allreduce(Tm, b)
        Tm = Tm.reshape(V.D, K, K)
        aw64 = (Wm.alpha if self.opts["ard_weights"] else torch.ones_like(Wm.alpha)).to(torch.float64).contiguous()
        if hasattr(self.be, "mofa_gs_update") and K <= 32:
            # one Gauss-Seidel sweep over the factors per feature, a thread per feature (csrc/mofa_stats.hip)
            self.be.mofa_gs_update(Tm.contiguous(), b.contiguous(), aw64, Wm.lth.to(torch.float64).contiguous(),
                                   Wm.l1mth.to(torch.float64).contiguous(), self.opts["spikeslab_weights"], Wm.EW,
                                   Wm.EW2, Wm.gamma, Wm.EWh2, Wm.sig2)
            return
        aw = aw64.to(self.T)
        lth, l1mth = Wm.lth.to(self.T), Wm.l1mth.to(self.T)
        EW = Wm.EW
        for k in range(K):
            t = b[:, k] - (EW * Tm[:, k, :]).sum(dim=1) + EW[:, k] * Tm[:, k, k]
            prec = Tm[:, k, k] + aw[k]
            s2 = 1.0 / prec
            mu = t * s2
            if self.opts["spikeslab_weights"]:
                lam = lth[k] - l1mth[k] + 0.5 * torch.log(aw[k]) - 0.5 * torch.log(prec) + 0.5 * t * t * s2
                gam = torch.sigmoid(lam)
            else:
                gam = torch.ones_like(mu)
            EW[:, k] = gam * mu
            Wm.EW2[:, k] = gam * (mu * mu + s2)
            Wm.gamma[:, k] = gam
            Wm.EWh2[:, k] = gam * (mu * mu + s2) + (1.0 - gam) / aw[k]
            Wm.sig2[:, k] = s2

    def _update_z(self):
        K = self.K
        WW = [self._outer_moments(w.EW, w.EW2) for w in self.W]
        az = (self.alpha_z if self.opts["ard_factors"] else torch.ones_like(self.alpha_z)).to(self.T)
        # (chunks of samples sized by the views that are walked in dense chunks: a fused poisson view needs none, and
        #  the [rows, K^2] statistics themselves bound the rest)
        step = min([self._rows_per_chunk(v.D) for v in self.views
                    if not (getattr(v, "fused", False) or getattr(v, "fusedb", False))]
                   + [self._rows_per_chunk(K * K)])
        # fused poisson views: a = R <W> for ALL samples at once (a sample's row depends on its own <z_n> only, which
        # changes in its own chunk, after use) and the sample-independent S
        fused = {m: (self.be.mofa_poisson_pass(0, self.EZ.contiguous(), self.W[m].EW.contiguous(), V.kappa.contiguous(), V.X, pads=self._pois_pads),
                     V.kappa @ WW[m])
                 for m, V in enumerate(self.views) if getattr(V, "fused", False)}
        # fused bernoulli views: S_n = sum_d Omega_nd <w w^T>_d for ALL samples in one sweep (a sample's row depends on
        # its own moments only), a = (Y - 1/2) <W>
        fusedb = {m: (self._times_block(V.Xe if V.Xe is not None else V.X, self.W[m].EW) - 0.5 * self.W[m].EW.sum(dim=0)[None, :],
                      self.be.mofa_jaakkola_sweep(self.EZ, self.EZ2, self.W[m].EW, self.W[m].EW2).reshape(self.N, K * K))
                  for m, V in enumerate(self.views) if getattr(V, "fusedb", False)}
        for g, (a0, b0) in enumerate(self.gslice):
            for lo in range(a0, b0, step):
                hi = min(b0, lo + step)
                Zc, Z2c = self.EZ[lo:hi], self.EZ2[lo:hi]
                # (views whose share is a product come first and START the sums - see _update_w; row-constant shares
                #  and chunk-walked views add to them)
                S = a = None
                order = sorted(range(self.M), key=lambda m: 0 if (self.views[m].stats and m not in fused) else 1)
                for m in order:
                    V = self.views[m]
                    if S is None and not (V.stats and m not in fused and V.mask is not None):
                        S = torch.zeros((hi - lo, K * K), dtype=self.T, device=self.dev)
                        a = torch.zeros((hi - lo, K), dtype=self.T, device=self.dev)
                    if m in fused:
                        S += fused[m][1][None, :]
                        a += fused[m][0][lo:hi]
                        continue
                    if m in fusedb:
                        S += fusedb[m][1][lo:hi]
                        a += fusedb[m][0][lo:hi]
                        continue
                    if V.stats:
                        tau = self.W[m].tau[g][:, None]
                        if V.mask is not None:
                            if S is None:
                                S = V.mask[lo:hi] @ (tau * WW[m])
                                a = V.Y[lo:hi] @ (tau * self.W[m].EW)
                                continue
                            S += V.mask[lo:hi] @ (tau * WW[m])
                        else:
                            S += (tau * WW[m]).sum(dim=0)[None, :]
                        a += V.Y[lo:hi] @ (tau * self.W[m].EW)
                        continue
                    for l2, h2, Y, M in self._chunks_range(V, lo, hi):
                        Om, R, _, om_vec = self._omega_r(V, self.W[m], g, Y, M, self.EZ[l2:h2], self.EZ2[l2:h2])
                        if Om is None:
                            S[l2 - lo:h2 - lo] += (om_vec @ WW[m])[None, :]
                        else:
                            S[l2 - lo:h2 - lo] += Om @ WW[m]
                        a[l2 - lo:h2 - lo] += R @ self.W[m].EW
                S = S.reshape(hi - lo, K, K)
                ssf = self.opts["spikeslab_factors"]
                if hasattr(self.be, "mofa_gs_update") and K <= 32:
                    # (row slices of contiguous [N, K] tensors are contiguous: updated in place)
                    if ssf:  # the W form of the sweep: (alpha, ln theta, ln(1 - theta)) of this group
                        self.be.mofa_gs_update(S.contiguous(), a.contiguous(), az[g].to(torch.float64).contiguous(),
                                               self.lthz[g].contiguous(), self.l1mthz[g].contiguous(), True, Zc, Z2c,
                                               self.gamma_z[lo:hi], self.EZh2[lo:hi], self.sig2z[lo:hi])
                    else:
                        self.be.mofa_gs_update(S.contiguous(), a.contiguous(), az[g].to(torch.float64).contiguous(), None,
                                               None, False, Zc, Z2c, None, None, self.sig2z[lo:hi])
                    continue
                for k in range(K):
                    num = a[:, k] - (Zc * S[:, k, :]).sum(dim=1) + Zc[:, k] * S[:, k, k]
                    prec = az[g, k] + S[:, k, k]
                    s2 = 1.0 / prec
                    mu = num * s2
                    if ssf:
                        lam = ((self.lthz[g, k] - self.l1mthz[g, k]).to(self.T) + 0.5 * torch.log(az[g, k]) - 0.5 * torch.log(prec)
                               + 0.5 * num * num * s2)
                        gz = torch.sigmoid(lam)
                        Zc[:, k] = gz * mu
                        Z2c[:, k] = gz * (mu * mu + s2)
                        self.gamma_z[lo:hi, k] = gz
                        self.EZh2[lo:hi, k] = gz * (mu * mu + s2) + (1.0 - gz) / az[g, k]
                    else:
                        Zc[:, k] = mu
                        Z2c[:, k] = mu * mu + s2
                    self.sig2z[lo:hi, k] = s2

    def _bump_z(self):
        self._zver += 1

    def _chunks_range(self, V, lo, hi):
        """_chunks restricted to [lo, hi) (a Z-update chunk may span several chunks of a wide view)."""
        return self._chunks(V, lo, hi)

    def _update_rest_and_elbo(self):
        o, K, G = self.opts, self.K, self.G
        f64 = torch.float64
        lik = torch.zeros((), dtype=f64, device=self.dev)
        for m, (V, Wm) in enumerate(zip(self.views, self.W)):
            # a stats view's expected squared residuals in one kernel per group, its node in one more (csrc/mofa_elbo.hip:
            # ~35 tensor launches per view and iteration otherwise); the tensor forms below remain for the CPU operator set
            fast_stats = bool(V.stats and self._elbo_work is not None and hasattr(self.be, "mofa_stats_resid")
                              and Wm.EW.dtype == self.T)
            if not fast_stats:
                S = torch.zeros((G, V.D), dtype=f64, device=self.dev)
                Ngd = torch.zeros((G, V.D), dtype=f64, device=self.dev)
            part = torch.zeros((), dtype=f64, device=self.dev) if V.lik != "gaussian" else None
            chunked = not (getattr(V, "fused", False) or getattr(V, "fusedb", False) or V.stats)
            W2, Wsq = (Wm.EW2, Wm.EW ** 2) if chunked else (None, None)
            if getattr(V, "fused", False) and getattr(self.be, "mofa_poisson_lik_with_b", False):
                # the likelihood term and the NEXT W update's b = R^T <Z> read the same predictions: one sweep (r05)
                out = self.be.mofa_poisson_pass(3, Wm.EW.contiguous(), self.EZ.contiguous(), V.kappa.contiguous(), V.Xt, pads=self._pois_pads)
                hit = self._bnext.get(m)
                if hit is None:  # (a fixed buffer, written in place: a captured iteration finds the previous replay's b)
                    hit = self._bnext[m] = [None, torch.empty((V.D, K), dtype=self.T, device=self.dev)]
                hit[1].copy_(out[:, :K])
                hit[0] = (self._zver, self._wver[m])
                part += out[:, K].sum(dtype=f64)
            elif getattr(V, "fused", False):
                part += self.be.mofa_poisson_pass(2, self.EZ.contiguous(), Wm.EW.contiguous(), None, V.X, pads=self._pois_pads).sum(dtype=f64)
            elif getattr(V, "fusedb", False):
                # sum y zeta - ln(1 + e^zeta): the stored entries through Y <W>, the rest as the poisson view's sweep
                part += (self.EZ * self._times_block(V.Xe if V.Xe is not None else V.X, Wm.EW)).sum(dtype=f64)
                part += self.be.mofa_softplus_sweep(self.EZ.contiguous(), Wm.EW.contiguous(), pads=self._pois_pads).sum(dtype=f64)
            if V.stats and fast_stats:
                Bs, Qs = self._gauss_stats(m)
                S = torch.empty((G, V.D), dtype=f64, device=self.dev)
                for g in range(G):
                    self.be.mofa_stats_resid(V.yyM[g], Wm.EW, Wm.EW2, Bs[g], Qs[g], S[g])
                Ngd = V.Ngd.clone() if self.comm.world_size > 1 else V.Ngd  # (the ranks' sum is taken in place)
            elif V.stats:
                Bs, Qs = self._gauss_stats(m)
                WWm = self._outer_moments(Wm.EW, Wm.EW2)
                for g in range(G):
                    S[g] = V.yyM[g] - 2.0 * (Wm.EW * Bs[g]).sum(dim=1).to(f64) + (Qs[g] * WWm).sum(dim=1).to(f64)
                Ngd = V.Ngd.clone()
            for g, (a0, b0) in enumerate(self.gslice if chunked else []):
                for lo, hi, Y, M in self._chunks(V, a0, b0):
                    Zc, Z2c = self.EZ[lo:hi], self.EZ2[lo:hi]
                    zeta = Zc @ Wm.EW.T
                    if V.lik == "gaussian":
                        res = (Y - zeta) ** 2 + (Z2c @ W2.T - (Zc ** 2) @ Wsq.T)
                        if M is not None:
                            res = res * M
                            Ngd[g] += M.sum(dim=0).to(f64)
                        else:
                            Ngd[g] += float(hi - lo)
                        S[g] += res.sum(dim=0).to(f64)
                    elif V.lik == "poisson":
                        if hasattr(self.be, "mofa_poisson_pseudo") and Y.is_contiguous():
                            t = self.be.mofa_poisson_pseudo(zeta, Y, None, 1)
                        else:
                            rate = torch.nn.functional.softplus(zeta).clamp(min=1e-300 if self.T == f64 else 1e-30)
                            t = Y * torch.log(rate) - rate
                        part += ((t * M) if M is not None else t).sum(dtype=f64)
                    else:
                        t = Y * zeta - torch.nn.functional.softplus(zeta)
                        part += ((t * M) if M is not None else t).sum().to(f64)
            if V.lik == "gaussian" and fast_stats:
                S, Ngd = self._allreduce(S, Ngd)
                self.be.mofa_tau_finish(S, Ngd, A0, B0, Wm.tau, Wm.ltau, lik, self._elbo_work)  # (lik += the view's terms)
            elif V.lik == "gaussian":
                S, Ngd = self._allreduce(S, Ngd)
                a = A0 + 0.5 * Ngd
                b = B0 + 0.5 * S
                tau, ltau = a / b, torch.digamma(a) - torch.log(b)
                Wm.tau.copy_(tau.to(self.T))
                Wm.ltau.copy_(ltau.to(self.T))
                lik = lik + (0.5 * Ngd * (ltau - math.log(2 * math.pi)) - 0.5 * tau * S).sum()
                lik = lik + _gamma_kl(A0, B0, a, b, tau, ltau).sum()
            else:
                lik = lik + self._allreduce(part)
            if self._fused_small:
                continue  # (alpha, theta and their ELBO terms: one kernel per view below)
            EWh2, gam = Wm.EWh2.to(f64), Wm.gamma.to(f64)
            if o["ard_weights"]:
                a = torch.full((K,), A0 + 0.5 * V.D, dtype=f64, device=self.dev)
                b = B0 + 0.5 * EWh2.sum(dim=0)
                Wm.alpha.copy_(a / b)  # (in place, like every expectation: the iteration replays as a HIP graph)
                Wm.lalpha.copy_(torch.digamma(a) - torch.log(b))
            if o["spikeslab_weights"]:
                sg = gam.sum(dim=0)
                a, b = TH_A0 + sg, TH_B0 + V.D - sg
                Wm.lth.copy_(torch.digamma(a) - torch.digamma(a + b))
                Wm.l1mth.copy_(torch.digamma(b) - torch.digamma(a + b))
        if self._fused_small:
            be, work = self.be, self._elbo_work
            elbo = torch.zeros((), dtype=f64, device=self.dev)
            for V, Wm in zip(self.views, self.W):
                be.mofa_w_elbo(Wm.EWh2, Wm.gamma, Wm.sig2, o["ard_weights"], o["spikeslab_weights"], A0 + 0.5 * V.D, A0, B0,
                               TH_A0, TH_B0, Wm.alpha, Wm.lalpha, Wm.lth, Wm.l1mth, elbo, work)
            zs = self._zs
            for g, (a0, b0) in enumerate(self.gslice):
                be.mofa_z_sums(self.EZ2, self.sig2z, a0, b0, zs[g], work)
            self._allreduce(zs)
            be.mofa_z_elbo(zs, self._Ng_dev, o["ard_factors"], A0, B0, self.alpha_z, self.lalpha_z, elbo)
            return elbo + lik
        # factors: per-group sums over this rank's samples, added up over the ranks
        ssf = o["spikeslab_factors"]
        zs = torch.zeros((G, 5 if ssf else 2, K), dtype=f64, device=self.dev)
        for g, (a0, b0) in enumerate(self.gslice):
            ls = torch.log(self.sig2z[a0:b0].to(f64))
            if ssf:  # sums of <zhat^2>, gamma ln sig2, gamma, the entropy of the switches (per group, over the ranks)
                gz = self.gamma_z[a0:b0].to(f64)
                zs[g, 0] = self.EZh2[a0:b0].to(f64).sum(dim=0)
                zs[g, 1] = (gz * ls).sum(dim=0)
                zs[g, 2] = gz.sum(dim=0)
                zs[g, 3] = torch.nan_to_num(-(torch.xlogy(gz, gz) + torch.xlogy(1 - gz, 1 - gz))).sum(dim=0)
            else:
                zs[g, 0] = self.EZ2[a0:b0].to(f64).sum(dim=0)
                zs[g, 1] = ls.sum(dim=0)
        zs = self._allreduce(zs)
        Ng = self._Ng_dev
        if o["ard_factors"]:
            a = (A0 + 0.5 * Ng)[:, None].expand(G, K)
            b = B0 + 0.5 * zs[:, 0]
            self.alpha_z.copy_(a / b)
            self.lalpha_z.copy_(torch.digamma(a) - torch.log(b))
        if ssf:
            a, b = TH_A0 + zs[:, 2], TH_B0 + Ng[:, None] - zs[:, 2]
            self.lthz.copy_(torch.digamma(a) - torch.digamma(a + b))
            self.l1mthz.copy_(torch.digamma(b) - torch.digamma(a + b))
        # ---- prior / entropy terms (the same expressions as oracle run()) ----------------------------------
        elbo = lik
        for m, (V, Wm) in enumerate(zip(self.views, self.W)):
            aw = Wm.alpha if o["ard_weights"] else torch.ones((K,), dtype=f64, device=self.dev)
            law = Wm.lalpha if o["ard_weights"] else torch.zeros((K,), dtype=f64, device=self.dev)
            gam, EWh2, sig2 = Wm.gamma.to(f64), Wm.EWh2.to(f64), Wm.sig2.to(f64)
            elbo = elbo + (0.5 * law - 0.5 * aw * EWh2).sum()
            elbo = elbo + (gam * 0.5 * torch.log(sig2) + (1 - gam) * 0.5 * torch.log(1.0 / aw) + 0.5).sum()
            if o["spikeslab_weights"]:
                elbo = elbo + (gam * Wm.lth + (1 - gam) * Wm.l1mth).sum()
                ent = -(torch.xlogy(gam, gam) + torch.xlogy(1 - gam, 1 - gam))
                elbo = elbo + torch.nan_to_num(ent).sum()
                sg = gam.sum(dim=0)
                a, b = TH_A0 + sg, TH_B0 + V.D - sg
                elbo = elbo + _beta_kl(TH_A0, TH_B0, a, b, Wm.lth, Wm.l1mth).sum()
            if o["ard_weights"]:
                a = torch.full((K,), A0 + 0.5 * V.D, dtype=f64, device=self.dev)
                b = B0 + 0.5 * EWh2.sum(dim=0)
                elbo = elbo + _gamma_kl(A0, B0, a, b, aw, law).sum()
        az = self.alpha_z if o["ard_factors"] else torch.ones((G, K), dtype=f64, device=self.dev)
        laz = self.lalpha_z if o["ard_factors"] else torch.zeros((G, K), dtype=f64, device=self.dev)
        for g in range(G):
            elbo = elbo + (0.5 * laz[g] * Ng[g] - 0.5 * az[g] * zs[g, 0] + 0.5 * zs[g, 1] + 0.5 * Ng[g]).sum()
            if ssf:
                sg = zs[g, 2]
                elbo = elbo + ((Ng[g] - sg) * 0.5 * torch.log(1.0 / az[g])).sum()
                elbo = elbo + (sg * self.lthz[g] + (Ng[g] - sg) * self.l1mthz[g]).sum() + zs[g, 3].sum()
                a, b = TH_A0 + sg, TH_B0 + Ng[g] - sg
                elbo = elbo + _beta_kl(TH_A0, TH_B0, a, b, self.lthz[g], self.l1mthz[g]).sum()
            if o["ard_factors"]:
                a = (A0 + 0.5 * Ng[g]).expand(K)
                b = B0 + 0.5 * zs[g, 0]
                elbo = elbo + _gamma_kl(A0, B0, a, b, az[g], laz[g]).sum()
        return elbo

    # -- driver ----------------------------------------------------------------------------------------
    def _iteration(self) -> torch.Tensor:
        for m in range(self.M):
            self._update_w(m)
            self._wver[m] += 1
        self._update_z()
        self._bump_z()
        return self._update_rest_and_elbo()

    def _capture(self):
        torch.cuda.synchronize(self.be.device)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                out = self._iteration()
        except Exception as e:  # capture refused (a path that asks the device a question): stay eager
            import warnings

            warnings.warn(f"MOFA iteration not captured into a HIP graph ({e}); running eagerly")
            self._graph_ok = False
            torch.cuda.synchronize(self.be.device)
            return
        self._graph, self._graph_elbo = g, out

    def step(self):
        # (two eager iterations first: they answer the once-per-fit questions - which chunks hold every sample - and
        #  warm the allocator)
        if self._graph is None and self._graph_ok and self._eager_steps >= 2:
            self._capture()
        if self._graph is not None:
            self._graph.replay()
            e = float(self._graph_elbo.item())
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
                if self.comm.agree(100.0 * abs((self.elbo[-1] - self.elbo[-2]) / self.elbo[0]) < tol):
                    break
        return len(self.elbo)

    def variance_explained(self):
        """R2 (%) of every factor alone per (view, group) on the (pseudo-)data of the last sweep."""
        K, G = self.K, self.G
        ss = torch.zeros((self.M, G), dtype=torch.float64, device=self.dev)
        rs = torch.zeros((self.M, G, K), dtype=torch.float64, device=self.dev)
        for m, (V, Wm) in enumerate(zip(self.views, self.W)):
            for g, (a0, b0) in enumerate(self.gslice):
                for lo, hi, Y, M in self._chunks(V, a0, b0):
                    Zc = self.EZ[lo:hi]
                    Om, R, _, om_vec = self._omega_r(V, Wm, g, Y, M, Zc, self.EZ2[lo:hi])
                    if Om is None:
                        Om = om_vec[None, :].expand_as(R)
                    Yh = torch.where(Om > 0, R / torch.where(Om > 0, Om, torch.ones_like(Om)), torch.zeros_like(R))
                    obs = (Om > 0).to(self.T) if M is None else M
                    ss[m, g] += (obs * Yh * Yh).sum().double()
                    for k in range(K):
                        res = obs * (Yh - Zc[:, k:k + 1] * Wm.EW[:, k][None, :])
                        rs[m, g, k] += (res * res).sum().double()
        ss, rs = self._allreduce(ss, rs)
        r2 = torch.where(ss[:, :, None] > 0, 100.0 * (1.0 - rs / ss[:, :, None].clamp(min=1e-300)), torch.zeros_like(rs))
        return r2.cpu().numpy()

    def results(self, sort_factors=True):
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
