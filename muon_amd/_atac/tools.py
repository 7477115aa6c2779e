"""muon.atac.tl.lsi on MI355X.

Host side mirrors /root/reference/muon/_atac/tools.py:29-71 (signature, write-back into
``obsm["X_lsi"]``, ``uns["lsi"]["stdev"]``, ``varm["LSI"]``).  The truncated SVD itself is
NOT ARPACK (tools.py:53 -> scipy svds): it is a block Lanczos process on X^T X with full
re-orthogonalisation, thick restarts and Rayleigh-Ritz over the whole block Krylov space,
built on the row-stream SpMM kernel and judged against the reference by the largest principal
angle between the spans of the top-k right singular vectors.

  Q_0 = CholeskyQR2(randn(d, B))            B = 16/32/64 >= n_comps + oversample (wider: several blocks)
  step j:  Y_j = X Q_j                      (SpMM; cells row-sharded over the ranks)
           T = [Y_i^T Y_j], M = [Q_i^T Q_j] (f64 Grams on the matrix cores; T all-reduced)
           Ritz pairs of (T, M)             (<= 192 x 192: device Jacobi when available, else host LAPACK)
           stop on the Lanczos residual bound of the top-k subspace (Davis-Kahan), else
           Z = X^T Y_j (SpMM, all-reduce), Z <- (I - K K^T) Z, Q_{j+1} = CholeskyQR2(Z)
  finally: V = sum_i Q_i c_i ;  U = sum_i Y_i c_i S^-1 (scaling of tools.py:60-63 folded in)

Everything of size nnz, n x B or d x B stays in HBM; only B x B blocks cross PCIe.
"""
from __future__ import annotations

import logging
import os
import time
from typing import Optional

import numpy as np
import torch
import scipy.linalg

from .._comm import default_comm
from .._containers import is_anndata, is_mudata
from .preproc import canonical_csr, resident, upload_canonical  # noqa: F401

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


class _RedoOnHost(Exception):
    """The device-side Cholesky met a pivot that was not safely positive."""


def _orthonormalize(backend, Z: torch.Tensor, w: int, passes: int = 2, flag=None):
    """CholeskyQR(passes) in place on the replicated d x B block; returns the first Gram, whose
    trace tells how much of the block survived the projection before it.  With ``flag`` (an int32
    device scalar) the B x B Cholesky and the inverse of its factor run on the device
    (mu_chol_rinv_f64) and the Gram comes back as a device tensor: no host round trip."""
    first = None
    if flag is not None:
        for _ in range(passes):
            G, _cs = backend.gram(Z)
            if first is None:
                first = G
            backend.apply(Z, backend.chol_rinv(G, w, flag), out=Z)
        return Z, first
    for _ in range(passes):
        G, _cs = backend.gram(Z)
        Gh = G.cpu().numpy()
        if first is None:
            first = Gh
        M = _chol_inverse(Gh, w)
        backend.apply(Z, backend.to_device(M.astype(np.float32)), out=Z)
    return Z, first


def _project_out(backend, Z: torch.Tensor, blocks, passes: int = 2) -> torch.Tensor:
    """Z <- (I - K K^T) Z for the (near-)orthonormal blocks K = [Q_0 .. Q_j], classical
    Gram-Schmidt per block with f64 coefficients, repeated ("twice is enough")."""
    for _ in range(passes):
        for Qi in blocks:
            C = backend.gram_cross(Qi, Z)  # Q_i^T Z, B x B, f64
            backend.project_out_block(Qi, C, Z)  # Z -= Q_i C
    return Z


# ---- SURVEY 8e's form of an expansion: reduce-scatter -> CholeskyQR and projection on row slices -> all-gather (r06) ----
# MUON_AMD_Z_COLLECTIVE=rsqr.  After the reduce-scatter of Z = X^T Y every rank holds the summed rows [r0, r1) of the d x B
# block; the projection against the Krylov basis and the CholeskyQR2 run on that slice (the basis blocks are replicated:
# their slices are views), the B x B cross Grams and Grams of the slices are all-reduced - one packed message per pass -
# and the finished block is all-gathered.  The replicated d x B passes of the default form become d / W-row passes at
# the price of four small all-reduces per expansion.  Opt-in: nothing here can time it with more than one RCCL rank;
# tests/test_distributed_gloo.py checks it against the single-process result (the sums over the ranks are the same
# numbers added in another grouping: 1e-6 rad, not bits).
def _slice_or_none(t, rows):
    r0, r1 = rows
    return t[r0:r1] if r1 > r0 else None


def _project_out_rows(backend, comm, Z, blocks, rows):
    """One pass of Z <- (I - K K^T) Z on this rank's rows: every block's cross Gram from the same Z (the blocks are
    mutually orthonormal: the order of the subtractions does not matter beyond rounding), one all-reduce for all."""
    Zs = _slice_or_none(Z, rows)
    B = Z.shape[1]
    Cs = [backend.gram_cross(Qi[rows[0]:rows[1]], Zs) if Zs is not None else backend.zeros((B, B), torch.float64)
          for Qi in blocks]
    if Cs:
        comm.all_reduce_sum(*Cs)
    if Zs is not None:
        for Qi, C in zip(blocks, Cs):
            backend.project_out_block(Qi[rows[0]:rows[1]], C, Zs)


def _orthonormalize_rows(backend, comm, Z, w, rows, passes=2, flag=None):
    """CholeskyQR(passes) with the Gram summed over the ranks' row slices; returns the first Gram like _orthonormalize
    (a device tensor with ``flag``, a host array without)."""
    Zs = _slice_or_none(Z, rows)
    B = Z.shape[1]
    first = None
    for _ in range(passes):
        G = backend.gram(Zs)[0] if Zs is not None else backend.zeros((B, B), torch.float64)
        comm.all_reduce_sum(G)
        if flag is not None:
            if first is None:
                first = G
            M = backend.chol_rinv(G, w, flag)
        else:
            Gh = G.cpu().numpy()
            if first is None:
                first = Gh
            M = backend.to_device(_chol_inverse(Gh, w).astype(np.float32))
        if Zs is not None:
            backend.apply(Zs, M, out=Zs)
    return first


def _ritz(Tm: np.ndarray, Mm: np.ndarray, want: int):
    """Top ``want`` pairs of  T c = theta M c  (T = K^T X^T X K, M = K^T K from f64 Grams of the
    stored f32 blocks), descending.  M is the identity up to f32 rounding unless the Krylov space is
    exhausted: Cholesky reduction, and only if that is ill-conditioned an eigen-decomposition of M
    whose directions without independent content are truncated instead of inverted."""
    n = Mm.shape[0]
    Mm = (Mm + Mm.T) / 2
    # M = I + E with E at the level of f32 rounding (orthonormalised, projected blocks: the normal case): the symmetric
    # reduction M^-1/2 T M^-1/2 to first order, T - (E T + T E) / 2, and M^-1/2 = I - E / 2 for the vectors - two small
    # matrix products instead of Cholesky + triangular inverse + two products (r04: 0.9 -> 0.2 ms on a 128-row pencil;
    # the neglected terms are O(E^2) <= 1e-10 relative, far below the f32 floor of the blocks themselves)
    E = Mm - np.eye(n)
    if n and float(np.abs(E).max()) < 1e-5:
        H = E @ Tm
        Tr = Tm - (H + H.T) / 2
        Tr = (Tr + Tr.T) / 2
        got = min(want, n)
        th, Wr, info_ = scipy.linalg.lapack.dsyevd(Tr, lower=1)
        if info_ != 0:
            th, Wr = np.linalg.eigh(Tr)
        Wt = np.ascontiguousarray(Wr[:, n - got:][:, ::-1])
        C = np.zeros((n, want))
        C[:, :got] = Wt - (E @ Wt) / 2
        lam = np.zeros(want)
        lam[:got] = np.maximum(th[::-1][:got], 0)
        return lam, C
    S = None
    try:
        L = np.linalg.cholesky(Mm)
        dg = np.diag(L)
        if dg.min() > 1e-3 * dg.max():
            S = np.ascontiguousarray(scipy.linalg.lapack.dtrtri(L, lower=1)[0].T)  # L^-T
    except np.linalg.LinAlgError:
        pass
    if S is None:
        mu, E = np.linalg.eigh(Mm)
        keep = mu > 1e-8 * max(mu[-1], 1e-300)
        S = E[:, keep] / np.sqrt(mu[keep])
    Tr = S.T @ Tm @ S
    Tr = (Tr + Tr.T) / 2
    r = Tr.shape[0]
    got = min(want, r)
    # (the full divide-and-conquer solve beats LAPACK's subset drivers here; scipy's direct dsyevd
    #  wrapper is 1.7x faster than np.linalg.eigh on these 64 .. 192-row matrices - r03, measured)
    th, Wr, info_ = scipy.linalg.lapack.dsyevd(Tr, lower=1)
    if info_ != 0:
        th, Wr = np.linalg.eigh(Tr)
    C = np.zeros((n, want))
    C[:, :got] = S @ np.ascontiguousarray(Wr[:, r - got:][:, ::-1])  # (contiguous: stays on the BLAS path)
    lam = np.zeros(want)
    lam[:got] = np.maximum(th[::-1][:got], 0)
    return lam, C


_blas_threads = None
_debug_cb = None           # tests / experiments: called with the Ritz state of every step
_F32_EPS = 2.0 ** -24
ANGLE_TARGET = 1e-4        # BASELINE.json north_star: LSI subspace angle < 1e-4 vs the reference


def _single_threaded_host_blas():
    """The host side factorises matrices of a few hundred rows between kernel launches; a
    multi-threaded OpenBLAS turns those into millisecond-to-second stalls (thread wake-ups and
    contention with the process' other pools), so they run on the calling thread."""
    global _blas_threads
    if _blas_threads is None:
        from threadpoolctl import ThreadpoolController

        _blas_threads = ThreadpoolController()
    return _blas_threads.limit(limits=1, user_api="blas")


def _refine_f64(backend, comm, X, Xt, V0, k, n_obs, scale_embeddings, angle_goal=1e-6, max_blocks=8):
    """f64 arithmetic for f64 input (r06).  The reference hands an f64 X to f64 ARPACK
    (/root/reference/muon/_atac/tools.py:53); the Krylov process above keeps its blocks in f32 and stops at the f32 floor
    (~1e-5 rad on a 1 % gap).  This continues it in f64 from where it stopped: a block Krylov space
    [V0, A V0, A^2 V0, ...] of A = X^T X started from the top-w Ritz block V0 of the f32 run, every product accumulated in
    f64 (the row-stream SpMM's f64 blocks, 32 columns a launch; the stream's VALUES stay f32 - rounding them costs 4e-8
    rad on the 1.3 % gap of the suite's hardest case), blocks kept in f64 and orthogonalised twice against all before
    (CholeskyQR2, the w x w factorisations on the host as above), Rayleigh-Ritz on T = K^T (A K) with the EXACT residual
    R = A K c - theta K c of the top-k pairs, stopped when ||R||_F / (theta_k - theta_k+1) - the Davis-Kahan bound of the
    subspace angle - is below ``angle_goal``, or after ``max_blocks`` blocks.  The f32 noise of V0 is white over all d
    directions, so most of it dies with the first product; what lies next to sigma_k is inside the block.  Measured on
    the suite's hardest gapped case (3000 x 2500, 80 topics; true angle to f64 ARPACK of the same operand / bound):
    k = 50, 1.3 % gap: 1 block 9e-7 / 5e-4, 2: 2e-7 / 2e-5, 4: 3e-8 / 2e-6, 5 (where the 1e-6 goal stops): < 1e-8 / 4e-7;
    k = 26, 0.16 % gap: 5 blocks, < 7e-9 / 8e-7 - the bound is a guarantee and ~100 x the truth.
    Returns U [rows, k], s [k], V [d, k] (torch f64; U scaled like the f32 tail does) and a dict for ``info``.
    Rows are sharded like X: Y = X Q is local, Z = X^T Y all-reduced, everything d x w replicated."""
    from .._backend import DeviceCSR

    f64 = torch.float64
    world = getattr(comm, "world_size", 1)

    def operand(A):  # (a CSR operand multiplies in one dtype: f64 values for the f64 blocks; the row streams take both)
        if isinstance(A, DeviceCSR) and A.values.dtype != f64:
            return A.with_values(A.values.to(f64))
        return A

    X, Xt = operand(X), operand(Xt)

    def mult(A, Q):
        outs = []
        for c0 in range(0, Q.shape[1], 32):
            wd = min(32, Q.shape[1] - c0)
            blk = backend.zeros((Q.shape[0], 16 if wd <= 16 else 32), f64)
            blk[:, :wd] = Q[:, c0:c0 + wd]
            outs.append(backend.spmm(A, blk)[:, :wd])
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)

    def tn(A, B):
        """A^T B for two tall f64 blocks of a few dozen columns.  The library's dgemm puts this shape - 64 x 64 out of a
        200 000-long reduction - on ONE workgroup (10.7 ms a call at d = 200 000: 2 / 3 of the continuation's time in
        the first version); cut into 256 row chunks it is a batched product on 256 workgroups and a small sum."""
        rows, chunks = A.shape[0], 256
        per = rows // chunks
        if per < 64:
            return A.T @ B
        main = per * chunks
        G = torch.bmm(A[:main].view(chunks, per, A.shape[1]).transpose(1, 2), B[:main].view(chunks, per, B.shape[1])).sum(dim=0)
        if main < rows:
            G += A[main:].T @ B[main:]
        return G

    def orth(Z, scale=None):
        """Orthonormal basis of the block's span in f64 (Gram, eigen-decomposition on the host, a second pass for the
        rounding of the first).  Directions without independent content - a converged Ritz vector adds nothing to the
        next block, and a Krylov space that has reached the dimension of the matrix has none left: deflation - are
        dropped: squared lengths below 1e-13 of the block's largest, or below 1e-20 of ``scale``, the squared length its
        columns had BEFORE they were projected against the basis (what is left of such a direction is rounding, and as
        much of it lies along the basis as across it: normalising it would put a vector into the basis that is not
        orthogonal to it).  None if nothing is left."""
        for it in range(2):
            G = backend.to_host(tn(Z, Z))
            G = 0.5 * (G + G.T)
            if not np.all(np.isfinite(G)):
                return None
            mu, E = np.linalg.eigh(G)
            # (the Gram's own eigenvalues are good to ~1e-16 of its largest: squared lengths below 1e-13 of it are not
            #  told apart from rounding, whatever the block's history)
            floor_ = max(float(mu[-1]), 0.0) * 1e-13
            if it == 0 and scale is not None:
                floor_ = max(floor_, float(scale) * 1e-20)
            keep = mu > floor_
            if mu[-1] <= 0 or not keep.any():
                return None
            S = E[:, keep] / np.sqrt(mu[keep])
            Z = Z @ backend.to_device(np.ascontiguousarray(S), np.float64)
        return Z

    Q = orth(V0.to(f64))
    if Q is None:
        raise _RedoOnHost("f64 refinement: the f32 Ritz block is rank deficient")
    Qs, Ys, Zs = [], [], []
    T = np.zeros((0, 0))
    hist = []
    theta = C = None
    bound = float("inf")
    for m in range(max_blocks):
        Y = mult(X, Q)
        Z = mult(Xt, Y)
        if world > 1:
            Z = Z.contiguous()
            (getattr(comm, "all_reduce_sum_big", None) or comm.all_reduce_sum)(Z)
        Qs.append(Q)
        Ys.append(Y)
        Zs.append(Z)
        # T = K^T A K: the new block row and column (Z is summed over the ranks already: nothing to reduce; blocks may
        # have lost columns to deflation: offsets by their widths)
        col = backend.to_host(torch.cat([tn(Qi, Z) for Qi in Qs], dim=0))  # [columns so far, this block's]
        tot, wm = col.shape
        Tn = np.zeros((tot, tot))
        Tn[:tot - wm, :tot - wm] = T
        Tn[:, tot - wm:] = col
        Tn[tot - wm:, :] = col.T
        T = 0.5 * (Tn + Tn.T)
        lam, Cm = np.linalg.eigh(T)
        order = np.argsort(lam)[::-1]
        theta, C = lam[order], Cm[:, order]
        Ck = backend.to_device(np.ascontiguousarray(C[:, :k]), np.float64)
        th = backend.to_device(np.ascontiguousarray(theta[:k]), np.float64)
        R = torch.cat(Zs, dim=1) @ Ck - (torch.cat(Qs, dim=1) @ Ck) * th
        res = float(torch.linalg.norm(R))
        gap = float(theta[k - 1] - theta[k]) if theta.size > k else float(theta[k - 1])
        bound = res / gap if gap > 0 else float("inf")
        hist.append({"blocks": m + 1, "residual": res, "gap": gap, "angle_bound": bound})
        stop = bound <= angle_goal or (m >= 1 and res >= 0.5 * hist[-2]["residual"] and res <= 1e-13 * float(theta[0]))
        if world > 1:
            stop = comm.agree(stop)
        if stop or m + 1 == max_blocks:
            break
        Zn = Z.clone()
        for _ in range(2):
            for Qi in Qs:
                Zn -= Qi @ tn(Qi, Zn)
        Q = orth(Zn, scale=float(torch.max((Z * Z).sum(dim=0))))
        dead = Q is None
        if world > 1:
            dead = comm.agree(dead)
        if dead:  # nothing left outside the space: the Ritz pairs are exact
            bound = 0.0
            break
    Ck_h = C[:, :k]
    Kc = torch.cat(Qs, dim=1)
    V = Kc @ backend.to_device(np.ascontiguousarray(Ck_h), np.float64)
    # deterministic signs, independent of the basis: the largest-magnitude entry of every right vector positive
    idx = torch.argmax(V.abs(), dim=0)
    sg = torch.sign(V[idx, torch.arange(k, device=V.device)])
    sg[sg == 0] = 1
    V = V * sg
    s = np.sqrt(np.maximum(theta[:k], 0))
    U = (torch.cat(Ys, dim=1) @ backend.to_device(np.ascontiguousarray(Ck_h), np.float64)) * sg
    with np.errstate(divide="ignore", invalid="ignore"):
        U = U * backend.to_device(1.0 / s, np.float64)
    if scale_embeddings:
        # tools.py:60-63: (U - mean) / std with population std; mean(u^2) = 1/n exactly
        tot = U.sum(dim=0)
        if world > 1:
            comm.all_reduce_sum(tot)
        mean = tot / n_obs
        std = torch.sqrt(torch.clamp(1.0 / n_obs - mean * mean, min=0))
        U = (U - mean) / std
    return U, s, V, {"blocks": len(Qs), "products": 2 * len(Qs), "angle_bound": float(bound), "history": hist}


def lsi_device(backend, X, n_comps: int = 50, scale_embeddings: bool = True, *args, **kwargs):
    """See ``_lsi_device`` (same arguments); runs it with the host BLAS pinned to the calling thread."""
    with _single_threaded_host_blas():
        try:
            return _lsi_device(backend, X, n_comps, scale_embeddings, *args, **kwargs)
        except _RedoOnHost:
            # a block with (numerically) dependent columns: the host path truncates those directions
            kwargs["device_qr"] = False
            return _lsi_device(backend, X, n_comps, scale_embeddings, *args, **kwargs)


def _lsi_device(
    backend,
    X,
    n_comps: int = 50,
    scale_embeddings: bool = True,
    n_obs: Optional[int] = None,
    comm=None,
    n_iter: Optional[int] = None,
    tol: float = 1e-7,
    angle_tol: float = 3e-5,
    max_iter: int = 60,
    oversample: int = DEFAULT_OVERSAMPLE,
    seed: int = 1,
    Xt=None,
    return_info: bool = False,
    pack: Optional[bool] = None,
    max_blocks: Optional[int] = None,
    device_qr: Optional[bool] = None,
    start: Optional[torch.Tensor] = None,
    return_basis: bool = False,
    refine_f64: bool = False,
):
    """Truncated SVD of a device-resident CSR (row shard) by block Lanczos on X^T X with full
    reorthogonalisation and Rayleigh-Ritz over the whole block Krylov space.

    Returns ``(U, stdev, V[, info])``: U torch f32 [n_local, k] (this rank's rows, already
    scaled if asked), stdev numpy f64 [k], V torch f32 [d, k] - torch f64 with ``refine_f64`` (what ``lsi`` asks for when
    the caller's X is f64, like the reference's f64 ARPACK: the f32 process is continued in f64 arithmetic, ``_refine_f64``).  ``n_iter=None`` expands the
    Krylov space until the top-k Ritz subspace has settled (``angle_tol``); ``n_iter=q`` does
    exactly q expansions.  m blocks cost 2 m - 1 SpMMs."""
    comm = default_comm(comm)
    n_local, d = X.shape
    if n_obs is None:
        n_obs = int(comm.sum_scalar(n_local))
    k = int(n_comps)
    if k < 1 or k >= min(n_obs, d):
        raise ValueError(f"`k` must be an integer satisfying `0 < k < min(A.shape)`; got k={k}")
    target = min(k + max(int(oversample), 0), min(n_obs, d))  # Ritz vectors worth keeping
    B = _pick_block(min(target, 64))           # block width of the products (16 / 32 / 64)
    w = min(B, min(n_obs, d))                   # its active columns
    keep_blocks = -(-target // w)               # blocks a thick restart keeps (1 for n_comps <= 50)
    keep = keep_blocks * w
    # Blocks kept before a thick restart.  Every block more makes the host's Ritz problems bigger
    # ((blocks w)^2: 0.7 / 2.4 / 5.8 ms at 64 / 128 / 192 rows) and saves products on slowly converging
    # spectra.  Where a product is cheaper than that (< ~5e8 stored entries) the first expansions
    # restart after every block (10k x 30k: same 5 expansions, 17.4 -> 14.7 ms per call); a run that
    # has not converged by then is a hard one and keeps the larger space (measured on a 1.3 % gap:
    # 45 products with 3 blocks, > 120 with 2).
    early_cap = None
    # (stored entries per rank, averaged over the ranks: every rank must take the same decisions)
    nnz_rank = comm.sum_scalar(int(getattr(X, "nnz", 0))) / max(getattr(comm, "world_size", 1), 1)
    if max_blocks is None:
        max_blocks = keep_blocks + 2
        if nnz_rank <= 500_000_000:
            early_cap = keep_blocks + 1
    max_blocks = max(int(max_blocks), keep_blocks + 1)
    if X.values.dtype != torch.float32:
        X = X.with_values(X.values.to(torch.float32))
    # both operands of the iteration are read from their row streams; X^T's is built
    # straight from the CSR of X, no CSR of X^T in between.  (`pack=False`: plain CSR kernels.)
    if pack is None:
        pack = Xt is None and hasattr(backend, "can_stream") and backend.can_stream(X, B)
    from .._trace import mark

    Xcsr = X  # (the CSR as it came: the warm start below cuts a row range out of it)
    t4_err = take = None
    mark("lsi/operands")
    if pack:
        X, Xt = backend.stream_both(X)
        take = getattr(backend, "take_tpack4_err", None)
        t4_err = take() if take is not None else None  # (read with the first Gram fetch: no synchronisation of its own)
    elif Xt is None:
        Xt = backend.transpose(X)

    # Algorithm (ours; the reference's is ARPACK, tools.py:53).  Blocks Q_0 .. Q_j (d x B, replicated)
    # are an orthonormal basis K_j of the block Krylov space of A = X^T X started at a random block:
    #     Y_j = X Q_j                       (SpMM; rows sharded)
    #     T   = K^T A K = [Y_i^T Y_j]       (f64 Grams on the matrix cores, all-reduced)
    #     Rayleigh-Ritz on (T, M = K^T K)   -> theta (sigma^2), top-k Ritz vectors K c
    #     Z   = X^T Y_j  (SpMM + all-reduce), Z <- (I - K K^T) Z, Q_{j+1} = CholeskyQR2(Z), projected once more
    # Against plain subspace iteration (same two SpMMs per step) the Ritz step sees the whole Krylov
    # space instead of its last block, i.e. the best polynomial filter of degree j instead of the
    # monomial: the error drops like 1 / T_j(gamma) (Chebyshev) instead of gamma^-j.  Everything the
    # Ritz step needs (Y_i) is a by-product of the iteration; U = X V comes from the Y_i as well.
    #
    # Stopping rule (n_iter=None), evaluated right after Y_j = X Q_j so that no product is wasted: the
    # Lanczos residual estimate of every top-k Ritz pair, ||r_i|| <= ||B_{j+1}|| ||c_i[last block]||
    # (valid across thick restarts: Krylov-Schur form), is turned into an angle by Davis-Kahan,
    # sin(angle_i) <~ ||r_i|| / (theta_i - theta_{k+1}); `bound` is the root sum of squares over the k
    # vectors and `floor` = 2^-24 theta_1 / gap what f32 storage of the blocks leaves.  Stop when
    # bound < angle_tol (3e-5), when four expansions bought < 30 % (the f32 floor of an ill-conditioned
    # subspace), when the gap is zero and the top-k Ritz VALUES have settled to `tol` (relative change
    # over one expansion: all that can be said about k beyond the rank / sigma_k = sigma_k+1), when the
    # Krylov space is exhausted, or at max_iter.  `converged` = hypot(bound, floor) < 1e-4: a statement
    # about the angle, not about having stopped.
    # CholeskyQR with the B x B factorisation on the device wherever the Krylov space cannot run out
    # of independent directions (it can on matrices with a few thousand rows or columns: the host
    # path detects that and truncates)
    if device_qr is None:
        device_qr = hasattr(backend, "chol_rinv") and min(n_obs, d) >= 8192
    qr_flag = backend.zeros((1,), torch.int32) if device_qr else None
    pending_g1 = None  # first Gram of the last orthonormalisation, still on the device
    # (`start`: a d x B block to begin with instead of Gaussian noise; it is orthonormalised like the noise)
    Q0 = start.clone() if start is not None else backend.randn(d, B, seed)
    if w < B:
        Q0[:, w:] = 0
    Q0, _ = _orthonormalize(backend, Q0, w, passes=2, flag=qr_flag)
    # r05 - SUBSAMPLED POWER WARM START (big inputs, row streams): before the first full product the start block takes q
    # steps of the power iteration of the FIRST n / frac CELLS of this rank: Q0 <- orth(X_S^T (X_S Q0)), summed over the
    # ranks.  X_S^T X_S is (n_s / n) X^T X up to sampling noise, so one such step does to the block what the first Krylov
    # expansion would - at 2 / frac of the price of its two products plus the operands of the slice - and the block
    # Lanczos process that follows (same stopping rule, same Ritz procedure, on the WHOLE matrix) needs one expansion
    # less: 5 instead of 7 products at 1e6 x 200k.  Nothing about the answer changes: the iteration converges to the
    # top-k subspace of X as before and stops on its own residual bound; a start that does not help costs its 2 q / frac
    # products.  Measured at 1e6 x 200k (scripts/probes/lsi_warm_probe.py): cold 7 products, 319 ms per call; 1 / 16 of
    # the cells and one step 6 products (the bound after the first expansion, 1.5, still asks for a speculative product),
    # two steps 5 products, 264 ms; 1 / 32 and two steps 257 ms; 1 / 64: 254 ms; the top-50 subspaces of warm and cold
    # runs 1.0e-5 rad apart (two f32 runs through different Krylov spaces: the level at which f32 ARPACK repeats
    # itself), singular values 5e-9.  MUON_AMD_LSI_WARM = "frac:q" (default "32:2" above 5e8 stored entries per rank;
    # "0": cold start).
    # (several ranks: 1 / 16 of each rank's cells - since r06 the slice's products cost next to nothing (ranged products,
    #  below), and a rank's 3 500 cells were what made the emulated rank-of-8 call need a sixth product)
    warm_spec = os.environ.get("MUON_AMD_LSI_WARM", ("32:2" if getattr(comm, "world_size", 1) == 1 else "16:2")
                               if nnz_rank > 500_000_000 else "0")
    warm_used = None
    mark("lsi/warm_start")
    # (the block below holds collectives, so entering it must be ONE decision of all ranks.  `pack` is per
    #  rank - a rank whose shard has no rows has no row stream - and such a rank still takes part: with an empty slice it
    #  contributes zeros to the sums.  `warm_spec`, `start`, `n_iter` are the same on every rank by construction.)
    can_slice = bool(pack and hasattr(Xcsr, "indptr") and hasattr(backend, "stream_both"))
    if start is None and warm_spec != "0" and n_iter is None:
        frac, qsteps = (int(v) for v in (warm_spec.split(":") + ["2"])[:2])
        # this rank's slice: 1 / frac of its cells; the slices of ALL ranks together at least 16 384 cells (an experiment
        # of a few 1e5 cells still gets a slice whose top subspace means something - but eight ranks with 125 000 cells
        # each need 2 048 apiece for that, not 16 384), at most a quarter, whole 512-row blocks
        floor_rows = -(-16384 * n_local // max(int(n_obs), 1))
        n_s = min(max(n_local // max(frac, 1), floor_rows), n_local // 4)
        n_s = (n_s // 512) * 512 if can_slice else 0
        # r06: the slice as <= 16 ranges of whole row blocks of the transposition - X_S^T Y_S then runs on the row stream
        # of X^T as it is (the slice's cells are a contiguous piece of every row, found through the count pass' prefix
        # table) and X_S Q on a compact copy of the slice's rows with the column slabs split over the chip: no
        # transposition of the slice, no sorted layout, no device -> host read after the first call (the plan is cached
        # with the index arrays).  At a 125 000-cell shard the slice's operands and four products took 8.3 ms of r05's 50.
        wplan = None
        if n_s > 0 and getattr(Xt, "t4", None) is not None and hasattr(backend, "slice_plan") \
                and os.environ.get("MUON_AMD_LSI_WARM_SLICE", "ranges") == "ranges":
            wplan = backend.slice_plan(Xcsr, n_s)
            if wplan is not None:
                n_s = wplan["n_s"]
        if comm.agree(qsteps >= 1 and comm.sum_scalar(n_s) >= 8192):  # (all ranks take part in the collectives or none does)
            Ss = St = None
            if wplan is not None:
                Ss = backend.slice_stream(Xcsr, wplan)
            elif n_s > 0:
                # the slice = 16 row ranges spread evenly over this rank's cells (files list cells sample by sample:
                # the FIRST n_s cells may be one batch; a start from an odd slice costs an expansion, never the answer)
                chunks = 16 if n_s >= 2048 else 1  # (n_s is a multiple of 512: ranges of >= 128 rows)
                per = n_s // chunks
                starts = [c * (n_local // chunks) for c in range(chunks)]
                ip = Xcsr.indptr
                edge = ip[torch.tensor([v for st in starts for v in (st, st + per)], device=ip.device)].tolist()
                los, his = edge[0::2], edge[1::2]
                if sum(h - l for l, h in zip(los, his)) > 0:
                    parts, off = [], 0
                    for st, l, h in zip(starts, los, his):
                        parts.append(ip[st: st + per + (1 if st == starts[-1] else 0)] - l + off)
                        off += h - l
                    sub_ip = torch.cat(parts) if chunks > 1 else parts[0]
                    cat = (lambda a: torch.cat([a[l:h] for l, h in zip(los, his)])) if chunks > 1 else (lambda a: a[los[0]:his[0]])
                    Xsub = type(Xcsr)(sub_ip.contiguous(), cat(Xcsr.indices), cat(Xcsr.values), (per * chunks, d))
                    Ss, St = backend.stream_both(Xsub)
                    e2 = take() if take is not None else None  # (the slice's transposition is checked with the shard's)
                    if e2 is not None:
                        t4_err = e2 if t4_err is None else (t4_err | e2)
                    n_s = per * chunks
            for _ in range(qsteps):
                if wplan is not None:
                    Zs = backend.spmm_slice_t(Xt, wplan, backend.spmm_slice(Ss, Q0))
                else:
                    Zs = backend.spmm(St, backend.spmm(Ss, Q0)) if Ss is not None else torch.zeros_like(Q0)
                comm.all_reduce_sum(Zs)
                if w < B:
                    Zs[:, w:] = 0
                Q0, _ = _orthonormalize(backend, Zs, w, passes=2, flag=qr_flag)
            del Ss, St
            warm_used = {"cells": n_s, "power_steps": qsteps, "slice": "ranges" if wplan is not None else "operands"}
    mark("lsi/krylov")
    Qs, Ys, css = [Q0], [], []
    Tb, Mb = {}, {}  # (i, j), i <= j  ->  w x w f64 host blocks

    def assemble(blocks, m):
        A = np.zeros((m * w, m * w))
        for (i, j), G in blocks.items():
            A[i * w:(i + 1) * w, j * w:(j + 1) * w] = G
            if i != j:
                A[j * w:(j + 1) * w, i * w:(i + 1) * w] = G.T
        return A

    def launch_grams(Ynew, Yolds, Qnew, Qolds):
        # a new row / column of T (needs the sum over row shards) and of M (replicated) against the given blocks;
        # the results travel to the host asynchronously so that the caller can queue more device work first
        Gj, cs = backend.gram(Ynew)
        cross = [backend.gram_cross(Yi, Ynew) for Yi in Yolds]
        comm.all_reduce_sum(Gj, cs, *cross)
        mq = [backend.gram(Qnew)[0]] + [backend.gram_cross(Qi, Qnew) for Qi in Qolds]
        extra = [pending_g1, qr_flag] if (device_qr and pending_g1 is not None) else []
        nonlocal t4_err
        chk = [t4_err] if t4_err is not None else []  # the transposition's error word rides on the first fetch
        t4_err = None
        return backend.fetch_async([Gj, cs] + cross + mq + extra + chk), bool(extra), len(Yolds), bool(chk)

    def launch_block_grams(j):
        return launch_grams(Ys[j], Ys[:j], Qs[j], Qs[:j])

    def collect_block_grams(j, handle_extra, through=None):
        """Block row / column j of T and M from a launch_grams handle.  ``through`` (an (m_old w) x (j w) coefficient
        matrix): the launch was made against the m_old blocks of BEFORE a thick restart - the cross blocks against
        the j kept blocks K Cw are Cw^T applied to the stacked old ones."""
        nonlocal beta_hat, pending_g1
        handle, has_extra, n_old, has_chk = handle_extra
        got = handle.wait()
        if has_chk:
            backend.raise_tpack4(int(got[-1].reshape(-1)[0]))
            got = got[:-1]
        if has_extra:
            if int(got[-1].reshape(-1)[0]) != 0:
                raise _RedoOnHost()
            # ||B_{j+1}||_2 <= sqrt(||B^T B||_F): the safe side of the estimate, without an eigensolve
            beta_hat = float(np.sqrt(np.linalg.norm(got[-2][:w, :w])))
            pending_g1 = None
            got = got[:-2]
        Tb[(j, j)] = got[0][:w, :w]
        css.append(got[1][:w])
        Mb[(j, j)] = got[2 + n_old][:w, :w]
        t_old = [got[2 + i][:w, :w] for i in range(n_old)]
        m_old = [got[3 + n_old + i][:w, :w] for i in range(n_old)]
        if through is None:
            assert n_old == j
            for i in range(j):
                Tb[(i, j)], Mb[(i, j)] = t_old[i], m_old[i]
        else:
            ts, ms = np.vstack(t_old), np.vstack(m_old)  # (m_old w) x w
            for i in range(j):
                blk = slice(i * w, (i + 1) * w)
                Tb[(i, j)], Mb[(i, j)] = through[:, blk].T @ ts, through[:, blk].T @ ms

    def add_block_grams(j):
        collect_block_grams(j, launch_block_grams(j))

    def combine(blocks, coef, bias=None, chunk=None):
        # sum_i blocks[i] @ coef[i*w:(i+1)*w]  ->  list of [rows, B] tensors, `chunk` columns of coef each
        chunk = B if chunk is None else chunk
        starts = list(range(0, coef.shape[1], chunk))
        # every coefficient block (and bias row) of this call crosses PCIe in ONE copy
        # (r02: one small synchronous upload per block: ~20 per call on 10k x 30k)
        host = np.zeros((len(starts), len(blocks) + 1, B, B), dtype=np.float32)
        for ci, c0 in enumerate(starts):
            cols = coef[:, c0:c0 + chunk]
            for i in range(len(blocks)):
                host[ci, i, :w, :cols.shape[1]] = cols[i * w:(i + 1) * w]
            if bias is not None:
                host[ci, len(blocks), 0, :cols.shape[1]] = bias[c0:c0 + chunk]
        dev = backend.to_device(host)
        chunks = []
        for ci in range(len(starts)):
            bvec = dev[ci, len(blocks), 0] if bias is not None else None
            out = None
            for i, Bi in enumerate(blocks):
                part = backend.apply(Bi, dev[ci, i], bias=bvec if i == 0 else None)
                out = part if out is None else out.add_(part)
            chunks.append(out)
        return chunks

    def first_columns(chunks, ncol):
        full = chunks[0] if len(chunks) == 1 else torch.cat(chunks, dim=1)
        return full[:, :ncol]

    it = 0          # Krylov expansions done
    grow_left = 0 if os.environ.get("MUON_AMD_LSI_GROW", "1") == "0" else 1  # (see `grow` in the loop)
    beta_hat = 0.0
    big_products = nnz_rank > 500_000_000
    restarts = 0
    converged = n_iter is not None
    limit = n_iter if n_iter is not None else max_iter
    history, bounds = [], []
    bound, floor = np.inf, np.inf
    expect_final = False  # the last Ritz step predicted that the next one passes the test
    wasted = 0
    host = {"wait_ms": 0.0, "ritz_ms": 0.0}

    def product(A, Qd):
        return backend.spmm(A, Qd)

    # (SURVEY 8e's form, opt-in: see _project_out_rows above)
    rsqr = (os.environ.get("MUON_AMD_Z_COLLECTIVE", "allreduce") == "rsqr" and getattr(comm, "world_size", 1) > 1
            and hasattr(comm, "reduce_scatter_rows"))
    zrows = [None]  # rsqr: the rows of Z this rank owns between the reduce-scatter and the all-gather

    def expand_product(j):
        Zn = product(Xt, Ys[j])
        if rsqr:
            zrows[0] = comm.reduce_scatter_rows(Zn)
            return Zn
        big = getattr(comm, "all_reduce_sum_big", None)  # (plain all-reduce, or reduce-scatter + all-gather: _comm.py)
        (big or comm.all_reduce_sum)(Zn)
        return Zn

    # Small inputs (a product costs less than the host's Ritz step): the half of the NEXT expansion that does not need
    # this step's Ritz pairs - projection of Z against the blocks as they are, CholeskyQR, the product X Q_{j+1} and its
    # Grams against the present blocks - is queued before the host starts the Ritz step and runs under it (r04).  A
    # thick restart in between only re-expresses the kept blocks as K Cw: the new block is orthogonal to their span
    # either way, and its cross Grams are Cw^T applied to the ones computed against the old blocks (collect_block_grams
    # `through`).  A wrong guess ("not the last step") costs the two products it queued.
    pipeline = device_qr and not big_products and not rsqr and os.environ.get("MUON_AMD_LSI_PIPELINE", "1") != "0"

    def speculate(Z):
        nonlocal pending_g1
        if w < B:
            Z[:, w:] = 0
        Zp = _project_out(backend, Z, Qs, passes=1)
        Zp, G1 = _orthonormalize(backend, Zp, w, passes=2, flag=qr_flag)
        pending_g1 = G1
        Zp = _project_out(backend, Zp, Qs, passes=1)
        Yn = product(X, Zp)
        handle = launch_grams(Yn, list(Ys), Zp, list(Qs))
        pending_g1 = None  # (travels with `handle`)
        return Zp, Yn, handle

    cur = None  # (Grams of block j already launched, coefficient matrix of a restart in between)
    while True:
        j = len(Qs) - 1
        if cur is None:
            Ys.append(product(X, Qs[j]))
            cur = (launch_block_grams(j), None)
        # The host's Ritz step (a few ms of LAPACK on (m w)^2 matrices) would leave the GPU idle.
        # Unless this step is expected to be the last one, X^T Y_j - needed by every step but the
        # last - is queued BEFORE the host waits for the Grams, so the Ritz step runs under it.
        Z = None
        if it < limit and not (n_iter is None and expect_final):
            Z = expand_product(j)
        nxt = speculate(Z) if (pipeline and Z is not None) else None
        t_w = time.perf_counter()
        collect_block_grams(j, cur[0], through=cur[1])
        cur = None
        t_r = time.perf_counter()
        host["wait_ms"] += 1e3 * (t_r - t_w)
        # The Ritz step of the very first block cannot end the iteration (no residual estimate exists before the
        # first expansion: beta_hat = 0, bound = inf) and nothing else reads its result: skipped (r04; 0.4 ms of
        # host LAPACK per call, exposed on small inputs)
        # (host QR - small matrices - may find the Krylov space exhausted right after this block and needs the pairs)
        first = it == 0 and limit > 0 and len(Qs) == 1 and beta_hat == 0.0 and device_qr
        enough = False
        if first:
            history.append(np.zeros(k))
            bounds.append(np.inf)
        else:
            m = j + 1
            Tm, Mm = assemble(Tb, m), assemble(Mb, m)
            lam_all, C_all = _ritz(Tm, Mm, keep + 1)  # top-k pairs, the restart's `keep`, the first unwanted value
            # r06 - k INSIDE A CLUSTER of singular values (planted rank 80, n_comps = 50: a 1.3 % gap; real spectra decay
            # smoothly): a thick restart that keeps ONE block (64 vectors) cuts through the cluster every time and the
            # iteration crawls - 47 products on the 3000 x 2500 test matrix where keeping two blocks needs 25.  The Ritz
            # values say so before the first restart: when the gap behind sigma_k is under 3 % and the restart is due
            # while the bound is still far from the target, the restarts keep one block more from here on (and the space
            # one more before it restarts); the host's Ritz problems grow to (4 w)^2.  One decision of all ranks.
            grow = False
            if (n_iter is None and grow_left > 0 and m * w > k and len(Qs) >= max_blocks - (1 if early_cap is None else 0)
                    and lam_all[k - 1] > 0 and (lam_all[k - 1] - lam_all[k]) < 0.03 * lam_all[k - 1]):
                grow = True
            if comm.agree(grow) if n_iter is None and getattr(comm, "world_size", 1) > 1 else grow:
                grow_left -= 1
                keep_blocks += 1
                keep = keep_blocks * w
                max_blocks += 2
                early_cap = None
                lam_all, C_all = _ritz(Tm, Mm, keep + 1)
            lam, C, rest = lam_all[:k], C_all[:, :k], lam_all[k:]
            history.append(np.sqrt(lam))
            enough = m * w > k  # (n_comps > block width: the first Ritz steps cannot deliver k vectors yet)
            # Accuracy of the current top-k Ritz subspace, from the Krylov decomposition itself: every block
            # but the newest is mapped back into the space by A = X^T X (that is how the next block was
            # made), so the residual of a Ritz pair (theta_i, K c_i) is Q_{j+1} B_{j+1} c_i[last block] and
            # ||r_i|| <= ||B_{j+1}|| ||c_i[last]|| - the Lanczos residual estimate, also across thick restarts
            # (Krylov-Schur form).  B_{j+1} is not known before the next expansion; its norm is taken from
            # the last expansion (beta_hat).  Davis-Kahan turns residuals into angles vector by vector:
            # sin(angle_i) <~ ||r_i|| / (theta_i - theta_{k+1}); the subspace figure is their root sum of
            # squares.  Measured against f64 ARPACK it over-estimates the largest principal angle 2-4x on
            # gapped spectra and up to ~50x when sigma_k ~ sigma_{k+1} (never under, down to the f32 floor).
            gap = max(lam[k - 1] - rest[0], 0.0) if enough else 0.0
            bound = np.inf
            if enough and beta_hat > 0 and gap > 0:
                c_last = C[(m - 1) * w:, :]
                bound = float(np.sqrt(np.sum((beta_hat * np.linalg.norm(c_last, axis=0)
                                              / np.maximum(lam - rest[0], 1e-300)) ** 2)))
            # what f32 storage of the blocks leaves, whatever the iteration does: eps ||A|| / gap
            floor = float(_F32_EPS * lam_all[0] / gap) if gap > 0 else np.inf
            bounds.append(bound)
            if _debug_cb is not None:
                _debug_cb({"Qs": Qs, "C": C, "w": w, "m": m, "bound": bound, "floor": floor,
                           "gap_rel": float(gap / max(lam[k - 1], 1e-300))})
            if it >= limit and enough:
                host["ritz_ms"] += 1e3 * (time.perf_counter() - t_r)
                if n_iter is None:
                    converged = False  # max_iter expansions without meeting angle_tol
                break
            stop = False
            if n_iter is None and enough:
                # contraction per expansion: the last measured one, not below the asymptotic Chebyshev rate
                # computed from the Ritz values (unwanted spectrum in [0, theta_out])
                th_k, th_out = lam[k - 1], rest[-1]
                cheb = 0.0
                if th_k > th_out > 0:
                    g = 1.0 + 2.0 * (th_k - th_out) / th_out
                    cheb = 1.0 / (g + np.sqrt(g * g - 1.0))
                rho = 0.5
                if len(bounds) >= 2 and np.isfinite(bounds[-2]) and bounds[-2] > 0:
                    rho = min(0.9, max(bound / bounds[-2], 1.5 * cheb, 1e-3))
                # One more expansion multiplies the error by about rho again.  A wrong "final" costs an
                # exposed Ritz step (ms), a wrong "not final" an unused SpMM (tens of ms at 1e6 rows): lean
                # to "final" - the bound itself over-estimates 3x and more.
                expect_final = np.isfinite(bound) and bound * rho < 100.0 * angle_tol
                # Block Lanczos converges superlinearly once the space holds the wanted vectors (c3: bounds
                # 1.27, 0.21, 2.7e-7 in consecutive steps), so the rate says little.  When a product costs
                # more than the exposed Ritz step it would hide (> ~5e8 stored entries: 3 ms and up against
                # ~6 ms of host LAPACK and launch gaps for a wrong "final"), stop speculating as soon as the
                # vectors have started to converge.
                if big_products and np.isfinite(bound) and bound < 1.0:
                    expect_final = True
                if bound < angle_tol:
                    stop = True
                elif len(bounds) >= 6 and bound > 0.7 * bounds[-5] and bound < 1e-2:
                    stop = True  # four expansions bought < 30 %: the f32 floor of an ill-conditioned subspace
                elif not np.isfinite(bound) and gap <= 0 and len(history) >= 2 and it >= 2:
                    # no gap behind sigma_k (k beyond the rank, sigma_k = sigma_k+1): no angle can be promised;
                    # stop once the Ritz VALUES have settled to `tol` instead of running to max_iter
                    prev, cur = history[-2] ** 2, lam
                    if np.max(np.abs(cur - prev)) <= tol * max(lam_all[0], 1e-300):
                        stop = True
                # ranks must leave the loop together AND speculate together (expect_final gates a product
                # with its all-reduce): one broadcast carries both of rank 0's decisions
                stop, expect_final = comm.agree(stop, expect_final)
            host["ritz_ms"] += 1e3 * (time.perf_counter() - t_r)
            if stop:
                # "converged" is a statement about the ANGLE, not about having stopped: the Lanczos bound
                # and the f32 floor together must be under the parity target of the north star (1e-4)
                converged = bool(np.hypot(bound, floor) < ANGLE_TARGET)
                wasted += (Z is not None) + (nxt is not None)  # queued on a wrong prediction; the results are simply not used
                if nxt is not None and hasattr(nxt[2][0], "release"):
                    nxt[2][0].release()
                break
        # expand: next Krylov block
        if Z is None:
            Z = expand_product(j)
        if not enough:
            expect_final = False
        if nxt is None:
            if w < B:
                Z[:, w:] = 0
            if rsqr:
                before = None
                if not device_qr:
                    Zs0 = _slice_or_none(Z, zrows[0])
                    G0 = backend.gram(Zs0)[0] if Zs0 is not None else backend.zeros((B, B), torch.float64)
                    comm.all_reduce_sum(G0)
                    before = float(np.trace(G0.cpu().numpy()[:w, :w]))
                _project_out_rows(backend, comm, Z, Qs, zrows[0])
            else:
                before = None if device_qr else float(np.trace(backend.gram(Z)[0].cpu().numpy()[:w, :w]))
                # the next block is the part of A Q_j outside the WHOLE space built so far - also when that
                # space is about to be compressed: the residuals of all its Ritz vectors lie in this block
                Z = _project_out(backend, Z, Qs, passes=1)  # (the second pass follows the normalisation below)
        through = None
        if len(Qs) >= (early_cap if (early_cap is not None and it < 5) else max_blocks):
            # thick restart: the top-w Ritz vectors (and their images X v, linear combinations of
            # the Y_i: no SpMM) replace the blocks; the Krylov process continues from Z
            Cw = C_all[:, :keep]
            Vw = combine(Qs, Cw, chunk=w)  # `keep_blocks` new blocks of w active columns
            Yw = combine(Ys, Cw, chunk=w)
            # Grams of the new blocks, without a device pass and the host wait behind it (r04; was: recomputed
            # from the stored blocks, one more synchronisation per restart - a third of a 10k x 30k call):
            # the blocks are K Cw and (X K) Cw with T c_i = theta_i M c_i, c_i^T M c_j = delta_ij, so
            # T = diag(theta), M = I and the column sums are Cw^T applied to the old ones - exact for the
            # f64 combinations, and the f32 storage of the new blocks moves their true Grams by 2^-24
            # relative, the same order as the storage of every other block (`floor` accounts for it)
            cs_old = np.concatenate(css)
            Qs, Ys, css, Tb, Mb = list(Vw), list(Yw), [], {}, {}
            for jj in range(keep_blocks):
                blk = slice(jj * w, (jj + 1) * w)
                Tb[(jj, jj)] = np.diag(lam_all[blk])
                Mb[(jj, jj)] = np.diag((np.abs(Cw[:, blk]).sum(axis=0) > 0).astype(np.float64))  # (rank < keep: zero columns)
                css.append(Cw[:, blk].T @ cs_old)
                for ii in range(jj):
                    Tb[(ii, jj)] = np.zeros((w, w))
                    Mb[(ii, jj)] = np.zeros((w, w))
            C = np.zeros((keep, k))
            C[:k, :k] = np.eye(k)  # the kept Ritz vectors are the leading columns of the new blocks
            restarts += 1
            through = Cw
        if nxt is not None:
            # the next block, its image and their Grams (against the blocks of before the restart) are under way
            Qs.append(nxt[0])
            Ys.append(nxt[1])
            cur = (nxt[2], through)
            it += 1
            continue
        if rsqr:
            G1 = _orthonormalize_rows(backend, comm, Z, w, zrows[0], passes=2, flag=qr_flag)
        else:
            Z, G1 = _orthonormalize(backend, Z, w, passes=2, flag=qr_flag)
        if device_qr:
            pending_g1 = G1  # read with the Grams of the next step (no host round trip here)
        else:
            # ||B_{j+1}||_2 <= sqrt(||B^T B||_F): the safe side of the estimate, without an eigensolve
            beta_hat = float(np.sqrt(np.linalg.norm(G1[:w, :w])))
            if comm.agree(np.trace(G1[:w, :w]) <= 1e-12 * max(before, 1e-300)):
                converged = True  # nothing left outside the Krylov space: the Ritz pairs are exact
                bound = floor = 0.0
                break
        if rsqr:
            _project_out_rows(backend, comm, Z, Qs, zrows[0])  # the normalisation amplified what the f32 projection left
            comm.all_gather_rows_into(Z, zrows[0])
        else:
            Z = _project_out(backend, Z, Qs, passes=1)  # the normalisation amplified what the f32 projection left
        Qs.append(Z)
        it += 1

    mark("lsi/ritz_vectors")
    s = np.sqrt(lam)
    # deterministic signs: largest-magnitude coefficient of each Ritz vector positive
    sg = np.sign(C[np.argmax(np.abs(C), axis=0), np.arange(k)])
    sg[sg == 0] = 1
    C = C * sg

    V = first_columns(combine(Qs, C), k)

    with np.errstate(divide="ignore", invalid="ignore"):
        CS = C / s  # U = Y C S^-1 has unit-norm columns
    bias = None
    if scale_embeddings:
        # tools.py:60-63: (U - mean) / std with population std; mean(u^2) = 1/n exactly
        csh = np.concatenate(css)
        mean = (csh @ CS) / n_obs
        var = 1.0 / n_obs - mean**2
        std = np.sqrt(np.maximum(var, 0))
        with np.errstate(divide="ignore", invalid="ignore"):
            CS = CS / std
            bias = -mean / std
    U = first_columns(combine(Ys, CS, bias=bias), k)

    refined = None
    if refine_f64:
        mark("lsi/refine_f64")
        # the Ritz vectors the f32 run ends with: the top k and what its restarts keep beside them (at least one more:
        # the continuation's gap estimate; n_comps beyond the block width makes this several blocks' worth of columns)
        # (every column of the start is a column of every f64 product: the multiple of 32 that holds k and at least
        #  eight more - with one or two spare columns the continuation crawls: k = 63 of 64 took 8 blocks to 3e-6)
        ncol = min(C_all.shape[1], max(32, 32 * (-(-(k + 8) // 32))))
        V0 = first_columns(combine(Qs, np.ascontiguousarray(C_all[:, :ncol])), ncol)
        U, s, V, refined = _refine_f64(backend, comm, X, Xt, V0, k, n_obs, scale_embeddings)
        if refined["angle_bound"] <= 1e-4:
            converged = True

    stdev = s / np.sqrt(n_obs - 1)  # tools.py:65
    mark("")
    if return_info:
        basis = None
        if return_basis:  # the top-w Ritz vectors (one full block): what a warm start of another run begins with
            Cb = np.zeros((C_all.shape[0], w))
            Cb[:, :min(w, C_all.shape[1])] = C_all[:, :w]
            basis = combine(Qs, Cb)[0]
        info = {"iterations": it, "converged": bool(converged), "block": B, "width": w, "basis": basis,
                "blocks": len(Qs), "restarts": restarts, "spmm": 2 * it + 1 + wasted, "warm_start": warm_used,
                "spmm_unused": wasted, "host": host,
                "svalues": s, "history": history, "bounds": bounds,
                "angle_bound": float(np.hypot(bound, floor)), "lanczos_bound": float(bound),
                "f32_floor": float(floor), "gap_rel": float(gap / max(lam[k - 1], 1e-300)) if lam[k - 1] > 0 else 0.0,
                "refine_f64": refined}
        if refined is not None:
            info["spmm"] += refined["products"]
            info["angle_bound"] = float(refined["angle_bound"])
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
    ``tol`` / ``oversample`` / ``seed`` of the block Lanczos iteration (``lsi_device``).
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
        _host, Xd = upload_canonical(backend, X, values_dtype=np.float32)
    out_dtype = X.dtype if X.dtype in (np.float32, np.float64) else np.float64

    # an f64 X is answered in f64 arithmetic, like the reference's f64 ARPACK (tools.py:53): the f32 process, continued
    U, stdev, V, info = lsi_device(backend, Xd, n_comps=n_comps, scale_embeddings=scale_embeddings,
                                   comm=comm, n_iter=n_iter, tol=tol, oversample=oversample, seed=seed,
                                   return_info=True, refine_f64=bool(X.dtype == np.float64))
    if not info["converged"]:
        # never silently: sigma_k ~ sigma_{k+1} makes the top-k subspace itself ill-conditioned (the
        # reference's f32 ARPACK is only repeatable to ~6e-4 there); the singular values are still good
        logger.warning("lsi: top-%d subspace not resolved to 1e-4 rad (relative gap %.1e between sigma_k^2 and "
                       "sigma_k+1^2, guaranteed angle %.1e after %d products)", n_comps, info["gap_rel"],
                       info["angle_bound"], info["spmm"])

    adata.obsm["X_lsi"] = backend.to_host(U.contiguous()).astype(out_dtype)
    adata.uns["lsi"] = {"stdev": stdev.astype(out_dtype)}
    adata.varm["LSI"] = backend.to_host(V.contiguous()).astype(out_dtype)

    return None
