"""TEST INFRASTRUCTURE: a CPU stand-in for muon_amd._backend.HipBackend.

It lets the *host logic* of tfidf / lsi / mofa (sharding, collectives, small dense algebra,
write-back) run under ``-m "not gpu"`` - in particular the world_size-2 ``gloo`` tests - in
a container without a GPU.  It is never importable from the product package: muon_amd has no
CPU path and HipBackend raises without a device.  Every method has the HipBackend signature
and is implemented with scipy / torch CPU ops in the same precision (f32 data, f64 Grams).
"""
import numpy as np
import scipy.sparse as sp
import torch

from muon_amd._backend import DeviceCSR
from muon_amd._ffi import TFIDF_LOG_IDF, TFIDF_LOG_TF, TFIDF_LOG_TFIDF


class CpuTestBackend:
    name = "cpu-test"
    device = torch.device("cpu")

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def to_device(self, arr, dtype=None):
        return torch.as_tensor(np.ascontiguousarray(arr, dtype=dtype)).clone()

    def to_host(self, t, out=None):
        if out is not None:
            np.copyto(out, t.detach().numpy())
            return out
        return t.detach().numpy().copy()

    def fetch_async(self, tensors):
        got = [t.detach().numpy().copy() for t in tensors]

        class _Done:
            def wait(self):
                return got

        return _Done()

    def upload_csr(self, indptr, indices, values, shape, values_dtype=None):
        return DeviceCSR(self.to_device(indptr, np.int64), self.to_device(indices, np.int32),
                         self.to_device(values, values_dtype), (int(shape[0]), int(shape[1])))

    @staticmethod
    def _sp(X):
        return sp.csr_matrix((X.values.numpy(), X.indices.numpy(), X.indptr.numpy()), shape=X.shape)

    def row_col_sums(self, X):
        m = self._sp(X).astype(np.float64)
        return (torch.from_numpy(np.asarray(m.sum(axis=1)).reshape(-1).copy()),
                torch.from_numpy(np.asarray(m.sum(axis=0)).reshape(-1).copy()))

    def idf(self, colsum, n_obs, flags, dtype):
        with np.errstate(divide="ignore"):
            v = (np.asarray(n_obs, dtype=np.float64) / colsum.numpy())
        v = torch.from_numpy(v).to(dtype)
        if flags & TFIDF_LOG_IDF:
            v = torch.log1p(v)
        return v

    def tfidf_scale(self, X, rowsum, idf, scale, flags, out=None):
        T = X.values.dtype
        rows = torch.repeat_interleave(torch.arange(X.shape[0]), X.indptr[1:] - X.indptr[:-1])
        inv = (1.0 / rowsum.to(T))[rows]
        t = inv * X.values
        if not (scale == 0 or scale == 1):
            t = t * torch.tensor(scale, dtype=T)
        if flags & TFIDF_LOG_TF:
            t = torch.log1p(t)
        t = t * idf[X.indices.long()]
        if flags & TFIDF_LOG_TFIDF:
            t = torch.log1p(t)
        if out is not None:
            out.copy_(t)
            t = out
        return t, torch.tensor([int((t == 0).sum())])

    def compact_nonzero(self, X):
        m = self._sp(X)
        keep = m.data != 0
        rows = np.repeat(np.arange(X.shape[0]), np.diff(m.indptr))
        cnt = np.bincount(rows[keep], minlength=X.shape[0])
        indptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        return DeviceCSR(torch.from_numpy(indptr), torch.from_numpy(m.indices[keep].copy()),
                         torch.from_numpy(m.data[keep].copy()), X.shape)

    def binarize_values(self, values):
        values[values != 0] = 1

    def transpose(self, X):
        t = self._sp(X).T.tocsr()
        t.sort_indices()
        return DeviceCSR(torch.from_numpy(t.indptr.astype(np.int64)), torch.from_numpy(t.indices.astype(np.int32)),
                         torch.from_numpy(t.data.copy()), (X.shape[1], X.shape[0]))

    def spmm(self, X, Q, out=None):
        dt = Q.numpy().dtype
        y = torch.from_numpy((self._sp(X).astype(dt) @ Q.numpy()).astype(dt))
        if out is not None:
            out.copy_(y)
            return out
        return y

    def gram(self, A):
        a = A.double()
        return a.T @ a, a.sum(dim=0)

    def gram_cross(self, A, Bm):
        return A.double().T @ Bm.double()

    def apply(self, A, M, bias=None, out=None):
        r = A @ M
        if bias is not None:
            r = r + bias
        if out is not None:
            out.copy_(r)
            return out
        return r

    def chol_rinv(self, G, w, flag):
        """what mu_chol_rinv_f64 computes: R^-1 (f32, B x B) of the leading w x w block of G = R^T R; a pivot that
        is not safely positive sets ``flag`` (and is replaced by the threshold, like the kernel does)"""
        Bw = G.shape[0]
        g = G.numpy()[:w, :w].astype(np.float64).copy()
        dmax = float(np.max(np.diag(g))) if w else 0.0
        tiny = dmax * 1e-13 if dmax > 0 else 1.0
        L = np.zeros((w, w))
        for k in range(w):  # (unblocked, pivots clamped)
            piv = g[k, k] - L[k, :k] @ L[k, :k]
            if not piv > tiny:
                flag[0] = 1
                piv = tiny
            L[k, k] = np.sqrt(piv)
            L[k + 1:, k] = (g[k + 1:, k] - L[k + 1:, :k] @ L[k, :k]) / L[k, k]
        M = np.zeros((Bw, Bw), dtype=np.float32)
        if w:
            M[:w, :w] = np.linalg.inv(L).T.astype(np.float32)
        return torch.from_numpy(M)

    def project_out_block(self, Q, C, Z):
        Z -= self.apply(Q, C.to(torch.float32).contiguous())
        return Z

    def randn(self, rows, B, seed):
        g = torch.Generator().manual_seed(int(seed))
        return torch.randn((rows, B), generator=g, dtype=torch.float32)

    # MOFA+ sweeps: same arithmetic as csrc/mofa.hip, vectorised over rows with torch
    def mofa_update_w(self, B, tau, Gz, Z2, alpha, lth, l1mth, spikeslab, EW, EW2, gamma, EWh2, sig2):
        G, D, K = B.shape
        for k in range(K):
            t = torch.zeros(D, dtype=EW.dtype)
            q = torch.zeros(D, dtype=EW.dtype)
            for g in range(G):
                cross = EW @ Gz[g][:, k] - EW[:, k] * Gz[g][k, k]
                t += tau[g] * (B[g][:, k] - cross)
                q += tau[g] * Z2[g][k]
            prec = q + alpha[k]
            s2 = 1.0 / prec
            mu = t * s2
            if spikeslab:
                lam = lth[k] - l1mth[k] + 0.5 * torch.log(alpha[k]) - 0.5 * torch.log(prec) + 0.5 * t * t * s2
                gam = torch.sigmoid(lam)
            else:
                gam = torch.ones(D, dtype=EW.dtype)
            EW[:, k] = gam * mu
            EW2[:, k] = gam * (mu * mu + s2)
            gamma[:, k] = gam
            EWh2[:, k] = gam * (mu * mu + s2) + (1 - gam) / alpha[k]
            sig2[:, k] = s2

    # tau / alpha / theta nodes and the ELBO: the equations of csrc/mofa_elbo.hip as tensor operations
    # (f64 arithmetic, like the kernels)
    def mofa_elbo_work(self, K):
        return torch.zeros((1,), dtype=torch.float64)

    @staticmethod
    def _gamma_kl(a0, b0, a, b, ex, elx):
        import math

        lp = a0 * math.log(b0) - math.lgamma(a0) + (a0 - 1.0) * elx - b0 * ex
        lq = a * torch.log(b) - torch.lgamma(a) + (a - 1.0) * elx - b * ex
        return lp - lq

    def mofa_tau_elbo(self, yy, Ngm, EW, EW2, B, Gz, Z2, a0, b0, tau, ltau, elbo, work):
        import math

        G = B.shape[0]
        f = torch.float64
        W, W2 = EW.to(f), EW2.to(f)
        for g in range(G):
            Gg = Gz[g].to(f)
            S = (yy[g].to(f) - 2.0 * (W * B[g].to(f)).sum(dim=1) + ((W @ Gg) * W).sum(dim=1)
                 + W2 @ Z2[g].to(f) - (W ** 2) @ torch.diagonal(Gg))
            n = Ngm[g].to(f)
            a = a0 + 0.5 * n
            b = b0 + 0.5 * S
            t, lt = a / b, torch.digamma(a) - torch.log(b)
            tau[g] = t.to(tau.dtype)
            ltau[g] = lt.to(tau.dtype)
            elbo += (0.5 * n * (lt - math.log(2 * math.pi)) - 0.5 * t * S).sum()
            elbo += self._gamma_kl(a0, b0, a, b, t, lt).sum()

    def mofa_w_elbo(self, EWh2, gamma, sig2, ard, spikeslab, a_alpha, a0, b0, th_a0, th_b0, alpha, lalpha,
                    lth, l1mth, elbo, work):
        import math

        f = torch.float64
        D, K = EWh2.shape
        H, gam, s2 = EWh2.to(f), gamma.to(f), sig2.to(f)
        aw, law = torch.ones(K, dtype=f), torch.zeros(K, dtype=f)
        if ard:
            a = torch.tensor(float(a_alpha), dtype=f)
            b = b0 + 0.5 * H.sum(dim=0)
            aw, law = a / b, torch.digamma(a) - torch.log(b)
            alpha.copy_(aw.to(alpha.dtype))
            lalpha.copy_(law.to(alpha.dtype))
            elbo += self._gamma_kl(a0, b0, a, b, aw, law).sum()
        elbo += (0.5 * law - 0.5 * aw * H).sum()
        elbo += (gam * 0.5 * torch.log(s2) + (1 - gam) * 0.5 * torch.log(1.0 / aw) + 0.5).sum()
        if spikeslab:
            sg = gam.sum(dim=0)
            a = th_a0 + sg
            b = th_b0 + D - sg
            lt = torch.digamma(a) - torch.digamma(a + b)
            l1 = torch.digamma(b) - torch.digamma(a + b)
            lth.copy_(lt.to(lth.dtype))
            l1mth.copy_(l1.to(lth.dtype))
            elbo += (gam * lt + (1 - gam) * l1).sum()
            elbo += torch.nan_to_num(-(torch.xlogy(gam, gam) + torch.xlogy(1 - gam, 1 - gam))).sum()
            lb = torch.lgamma(a) + torch.lgamma(b) - torch.lgamma(a + b)
            lb0 = math.lgamma(th_a0) + math.lgamma(th_b0) - math.lgamma(th_a0 + th_b0)
            elbo += ((lb - lb0) + (th_a0 - a) * lt + (th_b0 - b) * l1).sum()

    def mofa_z_sums(self, EZ2, sig2, n0, n1, out, work):
        out[0] = EZ2[n0:n1].to(torch.float64).sum(dim=0)
        out[1] = torch.log(sig2[n0:n1].to(torch.float64)).sum(dim=0)

    def mofa_z_elbo(self, zs, Ng, ard, a0, b0, alpha_z, lalpha_z, elbo):
        f = torch.float64
        G, _two, K = zs.shape
        n = Ng.to(f)[:, None]
        az, laz = torch.ones((G, K), dtype=f), torch.zeros((G, K), dtype=f)
        if ard:
            a = (a0 + 0.5 * n).expand(G, K)
            b = b0 + 0.5 * zs[:, 0]
            az, laz = a / b, torch.digamma(a) - torch.log(b)
            alpha_z.copy_(az.to(alpha_z.dtype))
            lalpha_z.copy_(laz.to(alpha_z.dtype))
            elbo += self._gamma_kl(a0, b0, a, b, az, laz).sum()
        elbo += (0.5 * laz * n - 0.5 * az * zs[:, 0] + 0.5 * zs[:, 1] + 0.5 * n).sum()

    # the tall-skinny products of a dense view (csrc/skinny.hip); Y may be stored in f32 under an f64 block
    skinny_mixed = True

    def skinny_nn(self, Y, T16):
        return Y.to(T16.dtype) @ T16

    def skinny_tn(self, Y, Z16):
        return Y.to(Z16.dtype).T @ Z16

    def mofa_rowstats_work(self, K):
        return torch.zeros((1,), dtype=torch.float64)

    def mofa_rowstats(self, E, E2, r0, r1, work, wgt=None, aux=None, scale_out=False, out_pad=None, col0=0,
                      out_t=None, gram=None, s2=None, s1=None):
        K = E.shape[1]
        e, e2 = E[r0:r1].double(), E2[r0:r1].double()
        w = wgt[r0:r1].double() if wgt is not None else torch.ones(r1 - r0, dtype=torch.float64)
        a = aux[r0:r1].double() if aux is not None else torch.ones(r1 - r0, dtype=torch.float64)
        o = (w[:, None] * e) if scale_out else e
        if out_pad is not None:
            out_pad[r0:r1, col0:col0 + K] = o.to(out_pad.dtype)
        if out_t is not None:
            out_t[:K, r0:r1] = o.T.to(out_t.dtype)
        if gram is not None:
            gram.copy_(((w[:, None] * e).T @ e).to(gram.dtype))
        if s2 is not None:
            s2.copy_((w[:, None] * e2).sum(dim=0).to(s2.dtype))
        if s1 is not None:
            s1.copy_(((a * w)[:, None] * e).sum(dim=0).to(s1.dtype))

    def mofa_update_z(self, A, pres, grp, Gw, dw2, alphaz, EZ, EZ2, sig2, corr=None):
        M, N, K = A.shape
        g = grp.long()
        if corr is not None:
            A = A - corr[:, g, :]  # M x N x K
        for k in range(K):
            num = torch.zeros(N, dtype=EZ.dtype)
            prec = alphaz[g, k].clone()
            for m in range(M):
                gw = Gw[m][g]  # N x K x K
                cross = (EZ * gw[:, :, k]).sum(dim=1) - EZ[:, k] * gw[:, k, k]
                num += pres[m] * (A[m][:, k] - cross)
                prec += pres[m] * dw2[m][g, k]
            EZ[:, k] = num / prec
            EZ2[:, k] = EZ[:, k] ** 2 + 1.0 / prec
            sig2[:, k] = 1.0 / prec
