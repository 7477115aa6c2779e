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

    def to_device(self, arr):
        return torch.as_tensor(np.ascontiguousarray(arr)).clone()

    def to_host(self, t):
        return t.detach().numpy().copy()

    def fetch_async(self, tensors):
        got = [t.detach().numpy().copy() for t in tensors]

        class _Done:
            def wait(self):
                return got

        return _Done()

    def upload_csr(self, indptr, indices, values, shape):
        return DeviceCSR(self.to_device(np.asarray(indptr, dtype=np.int64)),
                         self.to_device(np.asarray(indices, dtype=np.int32)),
                         self.to_device(values), (int(shape[0]), int(shape[1])))

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

    def mofa_update_z(self, A, pres, grp, Gw, dw2, alphaz, EZ, EZ2, sig2):
        M, N, K = A.shape
        g = grp.long()
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
