"""muon.atac.tl.lsi on MI355X.

Host side mirrors /root/reference/muon/_atac/tools.py:29-71 (signature, write-back into
``obsm["X_lsi"]``, ``uns["lsi"]["stdev"]``, ``varm["LSI"]``).  The truncated SVD itself is
NOT ARPACK (tools.py:53 -> scipy svds): it is a block subspace iteration built on the CSR
SpMM kernel, judged against the reference by the principal angle between the spans of the
top-k right singular vectors.

  Q0 = randn(d, B)                          B = 16/32/64 >= n_comps + oversample
  repeat:  Y = X Q ; Z = X^T Y ; (all-reduce Z over row shards) ; Q = orth(Z)
  finally: Y = X Q ; G = Y^T Y = W S^2 W^T ; V = Q W ; U = Y W S^-1

orth() is CholeskyQR2: B x B Gram on the f64 matrix cores, Cholesky of the tiny Gram on
the host, triangular apply on the f32 matrix cores.  Everything of size nnz, n x B or d x B
stays in HBM; only B x B matrices cross PCIe.
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np
import torch
from scipy.sparse import issparse

from .._comm import default_comm
from .._containers import is_anndata, is_mudata
from .preproc import canonical_csr, resident

logger = logging.getLogger("muon_amd")

DEFAULT_OVERSAMPLE = 14  # 50 comps -> one 64-wide block


def _pick_block(width: int) -> int:
    from .._backend import pick_block

    return pick_block(width)


def _chol_inverse(G: np.ndarray, w: int):
    """R^-1 (upper) of the leading w x w block of the Gram G = R^T R, embedded in B x B.
    Falls back to an eigen-decomposition when G is numerically semi-definite."""
    B = G.shape[0]
    Gw = G[:w, :w]
    M = np.zeros((B, B))
    try:
        L = np.linalg.cholesky(Gw)
        Rinv = np.linalg.inv(L).T
        if not np.all(np.isfinite(Rinv)):
            raise np.linalg.LinAlgError("non-finite")
    except np.linalg.LinAlgError:
        lam, V = np.linalg.eigh(Gw)
        lam = np.maximum(lam, lam.max() * 1e-14 if lam.max() > 0 else 1.0)
        Rinv = V / np.sqrt(lam)
    M[:w, :w] = Rinv
    return M


def _ritz_subspace_sine(Ga, Cx, Gb, Wa, Wb) -> float:
    """sin of the largest principal angle between span(Qa Wa) and span(Qb Wb), from the f64 Grams
    Ga = Qa^T Qa, Cx = Qa^T Qb, Gb = Qb^T Qb (exact for the stored f32 columns, orthonormal or not)."""
    A = Wa.T @ Ga @ Wa          # (Qa Wa)^T (Qa Wa)
    Bm = Wb.T @ Gb @ Wb
    Cm = Wa.T @ Cx @ Wb
    La = np.linalg.cholesky((A + A.T) / 2)
    Lb = np.linalg.cholesky((Bm + Bm.T) / 2)
    M = np.linalg.solve(La, Cm)            # La^-1 Cm
    M = np.linalg.solve(Lb, M.T).T         # La^-1 Cm Lb^-T: cosines are its singular values
    # sin^2 = eigenvalues of I - M^T M (symmetric PSD); take the largest
    S = np.eye(M.shape[1]) - M.T @ M
    return float(np.sqrt(max(np.linalg.eigvalsh((S + S.T) / 2)[-1], 0.0)))


def _orthonormalize(backend, Z: torch.Tensor, w: int, passes: int = 2):
    """CholeskyQR(passes) in place on the replicated d x B block; returns the eigenvalues
    source (first Gram, host f64) for the convergence test."""
    first = None
    for _ in range(passes):
        G, _cs = backend.gram(Z)
        Gh = G.cpu().numpy()
        if first is None:
            first = Gh
        M = _chol_inverse(Gh, w)
        backend.apply(Z, backend.to_device(M.astype(np.float32)), out=Z)
    return Z, first


def lsi_device(
    backend,
    X,
    n_comps: int = 50,
    scale_embeddings: bool = True,
    n_obs: Optional[int] = None,
    comm=None,
    n_iter: Optional[int] = None,
    tol: float = 1e-7,
    angle_tol: float = 1e-5,
    max_iter: int = 60,
    oversample: int = DEFAULT_OVERSAMPLE,
    seed: int = 1,
    Xt=None,
    return_info: bool = False,
    pack: Optional[bool] = None,
):
    """Truncated SVD of a device-resident CSR (row shard) by block subspace iteration.

    Returns ``(U, stdev, V[, info])``: U torch f32 [n_local, k] (this rank's rows, already
    scaled if asked), stdev numpy f64 [k], V torch f32 [d, k].  ``n_iter=None`` iterates
    until the top-k singular value estimates change by less than ``tol`` (relative)."""
    comm = default_comm(comm)
    n_local, d = X.shape
    if n_obs is None:
        n_obs = int(comm.sum_scalar(n_local))
    k = int(n_comps)
    if k < 1 or k >= min(n_obs, d):
        raise ValueError(f"`k` must be an integer satisfying `0 < k < min(A.shape)`; got k={k}")
    w = min(k + max(int(oversample), 0), min(n_obs, d))  # active block width
    B = _pick_block(max(w, k))
    w = min(B, min(n_obs, d))
    if X.values.dtype != torch.float32:
        X = X.with_values(X.values.to(torch.float32))
    # B = 64: both operands of the iteration are streamed from their packed chunked-row copies
    # (DESIGN.md §4); X^T's is built straight from the CSR of X, no CSR of X^T in between.
    if pack is None:
        pack = (Xt is None and hasattr(backend, "can_pack") and backend.can_pack(X, B)
                and 0 < n_local <= (1 << 20))
    if pack:
        Xt = backend.transpose_pack(X)
        X = backend.pack(X)
    elif Xt is None:
        Xt = backend.transpose(X)

    Q = backend.randn(d, B, seed)
    if w < B:
        Q[:, w:] = 0
    Y = backend.spmm(X, Q)
    Z = backend.spmm(Xt, Y)
    comm.all_reduce_sum(Z)

    # Stopping rule (n_iter=None): after Y = X Q_j the Rayleigh-Ritz step is cheap (Gram of the n x B
    # block), so every iteration measures s_j = sin of the largest principal angle between the top-k
    # Ritz subspaces V_{j-1} = Q_{j-1} W_{j-1} and V_j = Q_j W_j.  The angle is computed from f64
    # Grams of the f32 blocks (Q_{j-1}^T Q_{j-1}, Q_{j-1}^T Q_j, Q_j^T Q_j), which is exact for the
    # columns actually stored, whatever orthogonality they lost in f32.  With the contraction
    # rho_j = s_j / s_{j-1} (floored at 0.05, capped at 0.9; 0.5 while unknown) the distance of V_j
    # from the limit is  err_j ~ s_j rho_j / (1 - rho_j);  we stop when 3 err_j < angle_tol (default
    # 1e-5, ten times under the 1e-4 parity target) or when s_j stagnates at the f32 noise floor.
    # Stopping happens AFTER Y = X Q_j, which is exactly what the final Rayleigh-Ritz needs, so no
    # product is wasted: q iterations cost 2 q + 1 SpMMs.
    it = 0
    converged = n_iter is not None
    limit = n_iter if n_iter is not None else max_iter
    history = []   # singular value estimates per iteration
    angles = []    # s_j
    Qprev = Gprev = Wprev = None
    have_ritz = False
    lam = W = csh = None
    err = None

    def ritz(Yb):
        G, cs = backend.gram(Yb)
        comm.all_reduce_sum(G, cs)
        Gh = G.cpu().numpy()[:w, :w]
        lam_, W_ = np.linalg.eigh(Gh)
        order = np.argsort(lam_)[::-1][:k]
        return np.maximum(lam_[order], 0), W_[:, order], cs.cpu().numpy()[:w]

    while True:
        Z, _G1 = _orthonormalize(backend, Z, w, passes=2)
        Q = Z
        it += 1
        Y = backend.spmm(X, Q, out=Y)
        if it >= limit:
            break
        if n_iter is None:
            lam, W, csh = ritz(Y)
            have_ritz = True
            history.append(np.sqrt(lam))
            Gq = backend.gram(Q)[0].cpu().numpy()[:w, :w]
            stop = False
            if Qprev is not None:
                Cx = backend.gram_cross(Qprev, Q).cpu().numpy()[:w, :w]
                s_j = _ritz_subspace_sine(Gprev, Cx, Gq, Wprev, W)
                angles.append(s_j)
                rho = 0.5 if len(angles) < 2 or angles[-2] <= 0 else min(0.9, max(0.05, s_j / angles[-2]))
                err = s_j * rho / (1.0 - rho)
                if 3.0 * err < angle_tol:
                    stop = True
                elif len(angles) >= 2 and s_j < 1e-4 and s_j > 0.5 * angles[-2]:
                    stop = True  # stagnated at the f32 noise floor
                if len(history) >= 2:
                    dsv = np.max(np.abs(history[-1] - history[-2]) / np.maximum(history[-1], 1e-300))
                    if dsv < tol * 1e-3:
                        stop = True
            if stop:
                converged = True
                break
            Qprev = Q.clone()
            Gprev, Wprev = Gq, W
            have_ritz = False
        Z = backend.spmm(Xt, Y)
        comm.all_reduce_sum(Z)

    # Rayleigh-Ritz on span(Q): Y = X Q is already there
    if not have_ritz:
        lam, W, csh = ritz(Y)
    s = np.sqrt(lam)
    # deterministic signs: largest-magnitude coefficient of each Ritz vector positive
    sg = np.sign(W[np.argmax(np.abs(W), axis=0), np.arange(k)])
    sg[sg == 0] = 1
    W = W * sg

    Mv = np.zeros((B, B))
    Mv[:w, :k] = W
    V = backend.apply(Q, backend.to_device(Mv.astype(np.float32)))[:, :k]

    with np.errstate(divide="ignore", invalid="ignore"):
        WS = W / s  # U = Y W S^-1 has unit-norm columns
    bias = None
    if scale_embeddings:
        # tools.py:60-63: (U - mean) / std with population std; mean(u^2) = 1/n exactly
        mean = (csh @ WS) / n_obs
        var = 1.0 / n_obs - mean**2
        std = np.sqrt(np.maximum(var, 0))
        with np.errstate(divide="ignore", invalid="ignore"):
            WS = WS / std
            b = np.zeros(B)
            b[:k] = -mean / std
        bias = backend.to_device(b.astype(np.float32))
    Mu = np.zeros((B, B))
    Mu[:w, :k] = WS
    U = backend.apply(Y, backend.to_device(Mu.astype(np.float32)), bias=bias)[:, :k]

    stdev = s / np.sqrt(n_obs - 1)  # tools.py:65
    if return_info:
        info = {"iterations": it, "converged": bool(converged), "block": B, "width": w,
                "svalues": s, "history": history, "angles": angles, "predicted_angle": err}
        return U, stdev, V, info
    return U, stdev, V


def lsi(data, scale_embeddings=True, n_comps=50, *, comm=None, n_iter: Optional[int] = None,
        tol: float = 1e-7, oversample: int = DEFAULT_OVERSAMPLE, seed: int = 1, backend=None):
    """
    Run Latent Semantic Indexing

    PARAMETERS
    ----------
    data:
            AnnData object or MuData object with 'atac' modality
    scale_embeddings: bool (default: True)
            Scale embeddings to zero mean and unit variance
    n_comps: int (default: 50)
            Number of components to calculate with SVD

    Keyword-only extras (not in the reference): ``comm`` for row-sharded input, ``n_iter`` /
    ``tol`` / ``oversample`` / ``seed`` of the block subspace iteration.
    """
    if is_anndata(data):
        adata = data
    elif is_mudata(data) and "atac" in data.mod:
        adata = data.mod["atac"]
    else:
        raise TypeError("Expected AnnData or MuData object with 'atac' modality")

    # In an unlikely scnenario when there are less 50 features, set n_comps to that value
    n_comps = min(n_comps, adata.X.shape[1])

    logger.info("Performing SVD")
    if backend is None:
        from .._backend import get_backend

        backend = get_backend()
    comm = default_comm(comm)

    X = adata.X
    Xd = resident(X, backend)  # still on the device from tfidf(): no PCIe upload
    if Xd is None:
        host = canonical_csr(X)
        Xd = backend.upload_csr(host.indptr, host.indices, host.data.astype(np.float32), host.shape)
    out_dtype = X.dtype if X.dtype in (np.float32, np.float64) else np.float64

    U, stdev, V = lsi_device(backend, Xd, n_comps=n_comps, scale_embeddings=scale_embeddings,
                             comm=comm, n_iter=n_iter, tol=tol, oversample=oversample, seed=seed)

    adata.obsm["X_lsi"] = backend.to_host(U.contiguous()).astype(out_dtype)
    adata.uns["lsi"] = {"stdev": stdev.astype(out_dtype)}
    adata.varm["LSI"] = backend.to_host(V.contiguous()).astype(out_dtype)

    return None
