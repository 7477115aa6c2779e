"""On-disk layouts -> row-sharded device CSR.

The reference reads a 10x matrix through scanpy (/root/reference/muon/_core/io.py:23-72 ``read_10x_h5``,
/root/reference/muon/_atac/io.py:11-22 ``read_10x_h5`` / ``read_10x_mtx`` with ``atac_only``, :125
``read_snap``'s triplets) into a host ``csr_matrix``, slices the peak columns on the host and hands the
result to ``tfidf`` - three host passes over every stored entry before the first kernel runs.

This module takes the ARRAYS those files hold (h5py is not in this image, so opening the container is
left to the caller: ``f["matrix"]`` of a 10x .h5 is exactly the mapping expected here) and builds the
device CSR of this rank's cells directly:

  * 10x ``matrix/{data, indices, indptr, shape}`` is CSC over (features x barcodes), i.e. already the CSR
    of cells x features: every rank uploads only ITS cells' slice of ``indices`` / ``data`` (pinned,
    pipelined, converted to int32 / f32 on the way - no ``astype`` copies, no ``tocsr()``);
  * ``atac_only``: the peak columns are selected ON THE DEVICE (a column map + the compaction kernel
    that tfidf already owns), cells keep their stored entries in order;
  * coordinate triplets (Matrix Market / snap): key sort + run-length row pointers on the device;
  * feature-major CSC (features compressed): transposed on the device (csrc/tpack4.hip).

``read_10x_arrays`` wraps the single-process case in an AnnData whose ``X`` shares the caller's arrays
when no column is dropped and carries the device copy, so ``tfidf`` / ``lsi`` start without another
PCIe upload (the residency mechanism of _atac/preproc.py).
"""
from __future__ import annotations

from typing import Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from .._comm import default_comm


def shard_rows(n_rows: int, comm=None) -> Tuple[int, int]:
    """[r0, r1): the cells of this rank (the same split bench.py uses)."""
    comm = default_comm(comm)
    w, r = getattr(comm, "world_size", 1), getattr(comm, "rank", 0)
    return r * n_rows // w, (r + 1) * n_rows // w


def _get(matrix, key):
    v = matrix[key]
    return v[()] if hasattr(v, "shape") and not isinstance(v, np.ndarray) and v.shape == () else v


def device_csr_from_10x(matrix: Mapping, comm=None, backend=None, atac_only: bool = True,
                        feature_types: Optional[Sequence] = None, values_dtype=np.float32):
    """Device CSR (cells x features) of this rank's cells from a 10x ``matrix`` group.

    ``matrix``: mapping with ``data``, ``indices``, ``indptr``, ``shape`` (= [n_features, n_barcodes])
    as array-likes supporting slicing (numpy arrays, h5py datasets); ``feature_types`` (or
    ``matrix["features"]["feature_type"]``) selects the "Peaks" columns when ``atac_only``.
    Returns ``(DeviceCSR, kept_feature_indices or None, (r0, r1))``."""
    if backend is None:
        from .._backend import get_backend

        backend = get_backend()
    shape = np.asarray(_get(matrix, "shape")).astype(np.int64)
    n_feat, n_cells = int(shape[0]), int(shape[1])
    r0, r1 = shard_rows(n_cells, comm)
    indptr = matrix["indptr"]
    ip = np.asarray(indptr[r0:r1 + 1]).astype(np.int64)
    p0, p1 = int(ip[0]), int(ip[-1])
    # only this rank's stored entries cross PCIe; index / value conversion happens on the way
    try:
        X = backend.upload_csr(ip - p0, matrix["indices"][p0:p1], matrix["data"][p0:p1], (r1 - r0, n_feat),
                               values_dtype=values_dtype, slab_ptr=False)  # (made below, for the arrays that stay)
    except TypeError:  # operator sets without the option (CPU tests)
        X = backend.upload_csr(ip - p0, matrix["indices"][p0:p1], matrix["data"][p0:p1], (r1 - r0, n_feat),
                               values_dtype=values_dtype)
    keep = None
    if atac_only:
        ft = feature_types
        if ft is None and "features" in matrix and "feature_type" in matrix["features"]:
            ft = matrix["features"]["feature_type"]
        if ft is not None:
            ft = np.asarray(ft)
            if ft.dtype.kind in ("S", "O"):
                ft = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in ft])
            mask = ft == "Peaks"
            if not mask.all():
                keep = np.nonzero(mask)[0]
                X = select_columns(backend, X, mask)
    X = canonicalize(backend, X)
    if hasattr(backend, "with_slab_ptr"):
        # ingest is where the slab pointers of the final index arrays are searched, once (tfidf and lsi read them)
        flag = getattr(X, "canonical_as_given", None)
        X = backend.with_slab_ptr(X)
        if flag is not None:
            X.canonical_as_given = flag
    return X, keep, (r0, r1)


def select_columns(backend, X, mask: np.ndarray):
    """Keep the columns where ``mask`` is True (renumbered), on the device: dropped entries get a zero
    value and fall out in the compaction kernel (which drops exactly-zero entries, as scipy's SpGEMM
    does in the reference's tfidf), the survivors' columns go through a lookup table."""
    from .._backend import DeviceCSR

    mask = np.asarray(mask, dtype=bool)
    newcol = np.cumsum(mask, dtype=np.int64) - 1
    newcol[~mask] = -1
    lut = backend.to_device(newcol.astype(np.int32))
    mapped = lut[X.indices.long()]
    vals = torch.where(mapped >= 0, X.values, torch.zeros_like(X.values))
    Y = backend.compact_nonzero(DeviceCSR(X.indptr, mapped.contiguous(), vals, (X.shape[0], int(mask.sum()))))
    return Y


def canonicalize(backend, X):
    """Column indices strictly ascending inside every row (10x files are written that way; checked on the
    device).  Otherwise the rows are sorted and duplicate (row, column) entries summed - what scipy's
    ``sum_duplicates`` does to such input on the host.  ``X.canonical_as_given`` tells the caller whether its own
    arrays already were canonical (the returned object is then X itself)."""
    if X.nnz < 2:
        X.canonical_as_given = True
        return X
    d = X.indices[1:] - X.indices[:-1]
    starts = torch.zeros(X.nnz, dtype=torch.bool, device=X.indices.device)
    first = X.indptr[1:-1]
    starts[first[first < X.nnz]] = True  # entry that opens a row: no order constraint across rows
    if bool(((d > 0) | starts[1:]).all()):
        X.canonical_as_given = True
        return X
    Y = _sort_rows(backend, X)
    Y.canonical_as_given = False
    return Y


def _sort_rows(backend, X):
    from .._backend import DeviceCSR

    n, d = X.shape
    rows = torch.repeat_interleave(torch.arange(n, device=X.indices.device), X.indptr[1:] - X.indptr[:-1])
    key = rows * int(d) + X.indices.long()
    order = torch.argsort(key, stable=True)
    key = key[order]
    vals = X.values[order]
    if bool((key[1:] == key[:-1]).any()):
        # duplicate (row, column) pairs: one entry each.  Entries without a duplicate pass through UNCHANGED (bit for
        # bit); the duplicated groups are summed group by group in f64, in stored order (torch.segment_reduce: one
        # sequential sum per segment - the same result run to run; index_add_ is atomics in arbitrary order, ADVICE
        # r04).  r05 took differences of ONE f64 running sum over all entries: exact for counts, but for general values
        # its error grew with the running total of the whole matrix and a single NaN / Inf poisoned every later entry
        #  - a group's sum now sees that group's values only.
        ukey, seg = torch.unique_consecutive(key, return_counts=True)
        ends = torch.cumsum(seg, 0) - 1
        uvals = vals[ends].clone()  # (singletons: their own value; duplicated groups: overwritten below)
        dup = seg > 1
        in_dup = torch.repeat_interleave(dup, seg)
        sums = torch.segment_reduce(vals[in_dup].to(torch.float64), "sum", lengths=seg[dup])
        uvals[dup] = sums.to(vals.dtype)
        cnt = torch.bincount(torch.div(ukey, int(d), rounding_mode="floor"), minlength=n)
        indptr = torch.zeros(n + 1, dtype=torch.int64, device=key.device)
        torch.cumsum(cnt, 0, out=indptr[1:])
        return DeviceCSR(indptr, (ukey % int(d)).to(torch.int32).contiguous(), uvals.contiguous(), X.shape)
    return DeviceCSR(X.indptr, X.indices[order].contiguous(), vals.contiguous(), X.shape)


def device_csr_from_coo(rows, cols, vals, shape, backend=None, values_dtype=np.float32, sum_duplicates=True):
    """Coordinate triplets (Matrix Market bodies, snap count tables: _atac/io.py:125) -> device CSR of
    ``shape`` = (n_rows, n_cols): one key sort on the device, duplicates summed."""
    from .._backend import DeviceCSR

    if backend is None:
        from .._backend import get_backend

        backend = get_backend()
    n, d = int(shape[0]), int(shape[1])
    r = backend.to_device(np.asarray(rows), np.int64)
    c = backend.to_device(np.asarray(cols), np.int64)
    v = backend.to_device(np.asarray(vals), values_dtype)
    key = r * d + c
    key, order = torch.sort(key, stable=True)
    v = v[order]
    if sum_duplicates and key.numel() > 1:
        uniq, inv = torch.unique_consecutive(key, return_inverse=True)
        if uniq.numel() != key.numel():
            acc = torch.zeros(uniq.numel(), dtype=v.dtype, device=v.device)
            acc.index_add_(0, inv, v)
            key, v = uniq, acc
    rr = torch.div(key, d, rounding_mode="floor")
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=v.device)
    indptr[1:] = torch.cumsum(torch.bincount(rr, minlength=n), dim=0)
    return DeviceCSR(indptr, (key - rr * d).to(torch.int32).contiguous(), v.contiguous(), (n, d))


def device_csr_from_csc(indptr, indices, data, shape, backend=None, values_dtype=np.float32):
    """Column-compressed cells x features (``indptr`` over the features) -> device CSR, transposed on
    the device (no host ``tocsr()``)."""
    if backend is None:
        from .._backend import get_backend

        backend = get_backend()
    n, d = int(shape[0]), int(shape[1])
    Xt = backend.upload_csr(indptr, indices, data, (d, n), values_dtype=values_dtype)  # CSR of X^T
    tr = getattr(backend, "transpose_csr", None) if Xt.values.dtype == torch.float32 else None
    return (tr or backend.transpose)(Xt)


def read_10x_arrays(matrix: Mapping, atac_only: bool = True, feature_types=None, barcodes=None,
                    feature_names=None, backend=None):
    """Single-process convenience (the shape of ``muon.atac.read_10x_h5``, _atac/io.py:11-15, for a file
    already opened by the caller): AnnData (cells x features, f32 CSR like sc.read_10x_h5 yields) whose
    ``X`` carries its device copy, so that ``pp.tfidf`` / ``tl.lsi`` start without a PCIe upload."""
    import pandas as pd
    from scipy.sparse import csr_matrix

    from .._atac.preproc import attach_device
    from .._containers import AnnData

    if backend is None:
        from .._backend import get_backend

        backend = get_backend()
    X, keep, _ = device_csr_from_10x(matrix, None, backend, atac_only, feature_types)
    shape = np.asarray(_get(matrix, "shape")).astype(np.int64)
    if keep is None and np.asarray(matrix["data"]).dtype == np.float32 and getattr(X, "canonical_as_given", False):
        # the caller's arrays ARE the canonical CSR: the host view shares them (the flags below used to be
        # set on this branch also when only the device copy had been sorted)
        host = csr_matrix((np.asarray(matrix["data"]), np.asarray(matrix["indices"]),
                           np.asarray(matrix["indptr"])), shape=(int(shape[1]), int(shape[0])))
    else:  # columns dropped / values converted / rows sorted on the device: the host view is the device result
        host = csr_matrix((backend.to_host(X.values), backend.to_host(X.indices), backend.to_host(X.indptr)),
                          shape=X.shape)
    host.has_sorted_indices = True
    host.has_canonical_format = True
    attach_device(host, X, backend)
    obs = pd.DataFrame(index=pd.Index([b.decode() if isinstance(b, bytes) else str(b) for b in barcodes])) \
        if barcodes is not None else None
    var = None
    if feature_names is not None:
        names = [f.decode() if isinstance(f, bytes) else str(f) for f in feature_names]
        if keep is not None:
            names = [names[i] for i in keep]
        var = pd.DataFrame(index=pd.Index(names))
    return AnnData(host, obs=obs, var=var)
