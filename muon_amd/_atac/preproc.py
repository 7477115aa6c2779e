"""muon.atac.pp.tfidf / binarize on MI355X.

Host side mirrors /root/reference/muon/_atac/preproc.py:16-152 (same signature, same
errors and warnings, same in-place write-back); the arithmetic (:92-117) runs as HIP
kernels through the C-ABI (csrc/tfidf.hip).  Extra keyword-only arguments (``comm``,
``n_obs``, ``match_scipy_order``, ``keep_on_device``) default to the reference's behaviour.
"""
from __future__ import annotations

from typing import Optional, Union
from warnings import warn

import numpy as np
from scipy.sparse import csr_matrix, issparse

from .._comm import default_comm
from .._containers import is_anndata, is_mudata, view_to_actual
from .._ffi import TFIDF_LOG_IDF, TFIDF_LOG_TF, TFIDF_LOG_TFIDF

DEVICE_ATTR = "_muon_amd_device"


_HASH_CHUNK = 1 << 26  # 64 MiB per task


def _digest_fn():
    """xxh3-64 when the ``xxhash`` package is there (~10 GB/s per thread), else blake2b from the
    standard library (~1 GB/s per thread; both release the GIL on large buffers).  Logged once."""
    global _DIGEST
    if _DIGEST is None:
        try:
            import xxhash

            _DIGEST = xxhash.xxh3_64_intdigest
        except ImportError:
            import hashlib
            import logging

            logging.getLogger("muon_amd").info(
                "xxhash is not installed: fingerprints of resident matrices use hashlib.blake2b (slower)")
            _DIGEST = lambda buf: int.from_bytes(hashlib.blake2b(buf, digest_size=8).digest(), "little")  # noqa: E731
    return _DIGEST


_DIGEST = None


_POOL = None


def _pool():
    """One thread pool for the host-side passes over whole matrices (hashing, copies), made once: starting threads
    costs ~12 ms each next to busy workers - r04 made a fresh 16-thread pool for every array it hashed, 0.7 s of the
    API path at 250 000 cells (scripts/probes/api_profile.py)."""
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor

        _POOL = ThreadPoolExecutor(max_workers=max(4, min(32, os.cpu_count() or 4)), thread_name_prefix="muon_amd_host")
    return _POOL


def _host_threads() -> int:
    """Cores this process may actually use: the affinity mask, capped by a cgroup CPU quota (a container with 256
    visible cores and a quota of 16 runs 64 threads slower than 16)."""
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:  # noqa: BLE001  (no cgroup v2 file: keep the affinity count)
        pass
    return max(1, min(n, 32))


def _native_hash():
    """libmuon_amd's multi-threaded host digest (mu_host_hash64) when the library is there; None otherwise."""
    global _NATIVE
    if _NATIVE is None:
        try:
            import ctypes as C

            from .. import _ffi

            lib = _ffi.lib()
            T = _host_threads()

            def h(b):
                out = C.c_uint64(0)
                _ffi.check(lib.mu_host_hash64(b.ctypes.data, b.size, T, 0, C.byref(out)))
                return int(out.value)

            _NATIVE = h
        except Exception:  # noqa: BLE001
            _NATIVE = False
    return _NATIVE or None


_NATIVE = None


def _hash_arrays(arrays) -> tuple:
    """64-bit digests over ALL bytes of every array (xxh3: ~10 GB/s per thread; every 64 MiB chunk of every array is
    one task of the shared pool, a chunk's digests are combined in order): the price of trusting a resident copy."""
    views = [np.ascontiguousarray(a).view(np.uint8).reshape(-1) for a in arrays]
    native = _native_hash()
    if native is not None:
        # one call per array, every core of the process inside it (the Python xxhash binding holds the GIL: one core,
        # 35-39 GB/s whatever the number of threads - scripts/probes/hash_rate.py)
        return tuple(native(b) for b in views)
    digest = _digest_fn()
    tasks = [(i, o) for i, b in enumerate(views) for o in range(0, max(b.size, 1), _HASH_CHUNK)]
    if sum(b.size for b in views) <= _HASH_CHUNK:
        parts = [digest(views[i][o:o + _HASH_CHUNK]) for i, o in tasks]
    else:
        parts = list(_pool().map(lambda t: digest(views[t[0]][t[1]:t[1] + _HASH_CHUNK]), tasks))
    out = []
    for i in range(len(views)):
        mine = [p for (j, _o), p in zip(tasks, parts) if j == i]
        out.append(mine[0] if len(mine) == 1 else digest(np.asarray(mine, dtype=np.uint64).view(np.uint8)))
    return tuple(out)


def _hash_array(a: np.ndarray) -> int:
    return _hash_arrays([a])[0]


def _copy_array(a: np.ndarray) -> np.ndarray:
    """``a.copy()`` on a few threads for the index arrays of a whole experiment (6e9 int32 = 25 GB: a single-threaded
    memcpy is seconds; np.copyto releases the GIL)."""
    a = np.ascontiguousarray(a)
    if a.nbytes < (64 << 20):
        return a.copy()
    out = np.empty_like(a)
    T = 16
    cuts = [a.size * k // T for k in range(T + 1)]
    list(_pool().map(lambda k: np.copyto(out[cuts[k]:cuts[k + 1]], a[cuts[k]:cuts[k + 1]]), range(T)))
    return out


def _refs_seen(obj) -> int:
    import sys

    return sys.getrefcount(obj)


class _Box:
    pass


def _matrix_refs(m, held_by_caller: int) -> int:
    """What ``_refs_seen`` reports for ``m`` inside a callee with _dies_with_rebinding's signature (the twin the
    calibration calls: same arguments, same call path down to sys.getrefcount)."""
    return _refs_seen(m)


def _probe_call(box):
    # the production call site's shape (tfidf below): the matrix comes out of a container attribute into the local
    # `counts`, `host` is a second local name for it, the callee gets the local and the literal 2
    counts = box.X
    host = counts
    return _matrix_refs(counts, 2), host is counts


_CALIBRATION = None


def _calibrate_refs():
    """The interpreter's own references, MEASURED on this interpreter instead of assumed (CPython 3.11+
    moves call arguments into the callee's frame without a second reference, 3.14 borrows locals - a hard-coded
    "+ 2" is one version's truth): (a) an array held by ONE container attribute, passed as an attribute lookup;
    (b) a matrix held by one container attribute and two locals of the caller, seen from a callee with
    _dies_with_rebinding's signature through the same call path.  Then the SELF-TEST of the whole decision on throw-away
    matrices: a sole owner must be taken over, a matrix with one more owner (a layer entry, a variable, a second
    matrix on the same arrays) must not.  ``ok`` False = the counts cannot be trusted here: never take over."""
    global _CALIBRATION
    if _CALIBRATION is not None:
        return _CALIBRATION
    b = _Box()
    b.x = np.empty(1)
    attr = _refs_seen(b.x)
    b.X = csr_matrix(np.eye(3, dtype=np.float32))
    matrix2, _ = _probe_call(b)
    _CALIBRATION = {"attr": attr, "matrix2": matrix2, "ok": True}  # (provisional: the self-test runs through it)

    def fresh():
        bx = _Box()
        m = csr_matrix(np.eye(4, dtype=np.float32))
        # (scipy leaves views of the constructor's arrays behind; own them like a matrix read from a file does)
        m.data, m.indices, m.indptr = m.data.copy(), m.indices.copy(), m.indptr.copy()
        bx.X = m
        m = None
        return bx

    def decide(bx):
        counts = bx.X
        host = counts
        return _dies_with_rebinding(counts, 2), host is counts

    ok = decide(fresh())[0] is True
    bx = fresh()
    bx.layers = {"counts": bx.X}            # one more owner: a layer
    ok = ok and decide(bx)[0] is False
    bx = fresh()
    keep = bx.X                              # ... a variable
    ok = ok and decide(bx)[0] is False
    keep = None
    bx = fresh()
    arr = bx.X.data                          # ... one of its arrays
    ok = ok and decide(bx)[0] is False
    arr = None
    bx = fresh()
    other = csr_matrix((bx.X.data, bx.X.indices, bx.X.indptr), shape=bx.X.shape)  # ... a second matrix on the arrays
    ok = ok and decide(bx)[0] is False
    other = None
    _CALIBRATION["ok"] = bool(ok)
    return _CALIBRATION


def _reuse_host_enabled(reuse_host) -> bool:
    """The takeover is OPT-IN (r06; it was on by default in r05): ``tfidf(..., reuse_host=True)`` or
    ``MUON_AMD_REUSE_HOST=1``.  A holder of a raw pointer into the old matrix (ctypes, a C extension, an
    ``__array_interface__`` consumer) is invisible to reference counts, and the reference never mutates the matrix it
    replaces (preproc.py:121-127): the default must not either."""
    import os

    if reuse_host is None:
        return os.environ.get("MUON_AMD_REUSE_HOST", "0") == "1"
    return bool(reuse_host)


def _dies_with_rebinding(m, held_by_caller: int) -> bool:
    """True when host CSR ``m`` and its three arrays are referenced by NOTHING but the container attribute the caller is
    about to rebind (plus ``held_by_caller`` local names of the caller): the result may then take the matrix's index
    arrays over and be downloaded into its value array - nobody holding a Python reference can observe the difference,
    and a 250 000 x 200 000 experiment saves a 6 GB copy, 12 GB of first-touch page faults and the release of the 12 GB
    it replaces (34 ms per GB on the bench host: 430 of the API path's 1090 ms, profiles/r05_api_profile.txt).
    Anything else that holds the matrix or one of its arrays - ``adata.layers["counts"] = adata.X``, a view, a variable
    of the user's - counts as a reference and turns this off.  The reference rebinds ``adata.X`` to a new matrix
    (preproc.py:121-127); an owner of the old one must keep seeing the counts.  Only reached when the caller opted in
    (_reuse_host_enabled); the expected counts are measured on this interpreter and self-tested (_calibrate_refs)."""
    if type(m) is not csr_matrix:
        return False
    cal = _calibrate_refs()
    if not cal["ok"]:
        return False
    attr = cal["attr"]
    if _refs_seen(m) != cal["matrix2"] + (held_by_caller - 2):
        return False
    for name in ("data", "indices", "indptr"):
        a = m.__dict__.get(name)
        if type(a) is not np.ndarray or not a.flags.writeable or not a.flags.c_contiguous or a.ndim != 1:
            return False
        if a.base is not None:
            # scipy's constructors leave `indices[:nnz]`-style views behind: a view of an array that owns its memory
            # and that nothing but the view refers to is as good as that array
            b = a.base
            ok = type(b) is np.ndarray and b.base is None and b.flags.owndata and b.flags.writeable
            b = None
            if not ok or _refs_seen(a.base) != attr:
                return False
        elif not a.flags.owndata:
            return False
        a = None
        if _refs_seen(getattr(m, name)) != attr:
            return False
    return True


def _fingerprint(m: csr_matrix):
    """Identity of a host CSR: shape, dtype and a hash of EVERY byte of data, indices and indptr.
    Any in-place edit between two calls - a single entry, a permuted index array - invalidates the
    resident copy (r01 sampled 2^16 values and could serve stale HBM data)."""
    return (m.shape, int(m.nnz), m.data.dtype.str, m.indices.dtype.str, m.indptr.dtype.str) + _hash_arrays(
        [m.data, m.indices, m.indptr])


def attach_device(m: csr_matrix, dev_csr, backend) -> None:
    """Remember the device-resident copy of host matrix ``m`` (binarize -> tfidf ->
    lsi pays PCIe for the upload once)."""
    try:
        setattr(m, DEVICE_ATTR, (dev_csr, backend, _fingerprint(m)))
    except Exception:  # noqa: BLE001  (objects without a __dict__)
        pass


def resident(m, backend):
    """The DeviceCSR attached to ``m`` by an earlier call if it still describes ``m``, else None."""
    if not issparse(m):
        return None
    ent = getattr(m, DEVICE_ATTR, None)
    if ent is None or ent[1] is not backend or m.format != "csr":
        return None
    if ent[2] != _fingerprint(m):
        try:
            delattr(m, DEVICE_ATTR)
        except Exception:  # noqa: BLE001
            pass
        return None
    return ent[0]


def _is_canonical(m) -> bool:
    return (issparse(m) and m.format == "csr" and m.dtype in (np.float32, np.float64)
            and m.has_canonical_format and m.has_sorted_indices)


def _is_canonical_csc(m) -> bool:
    return (issparse(m) and m.format == "csc" and m.dtype in (np.float32, np.float64)
            and m.has_canonical_format and m.has_sorted_indices)


def _flags(log_tf, log_idf, log_tfidf):
    return (TFIDF_LOG_TF if log_tf else 0) | (TFIDF_LOG_IDF if log_idf else 0) | (
        TFIDF_LOG_TFIDF if log_tfidf else 0
    )


def _effective_scale(scale_factor) -> float:
    # preproc.py:101: the multiply is skipped for None / 0 / 1
    if scale_factor is None or scale_factor == 0 or scale_factor == 1:
        return 1.0
    return float(scale_factor)


def _flags_unknown(m) -> bool:
    """scipy caches ``has_canonical_format`` / ``has_sorted_indices`` in private attributes; a matrix that has never
    been asked (a fresh ``.copy()``, a matrix assembled from arrays) answers with a single-threaded scan of every
    entry - 0.34 s at 1.6e9 entries, inside every tfidf() call of r04."""
    return getattr(m, "_has_canonical_format", None) is None


def canonical_csr_deferred(counts):
    """``(host CSR, checked)``: like ``canonical_csr`` but a float CSR whose canonical flags scipy has not computed yet
    is handed on UNCHECKED (``checked`` False) - the caller uploads it and lets the device say in milliseconds whether
    its rows are sorted and free of duplicates (``_core.io.canonicalize``), falling back to ``canonical_csr`` if not."""
    if (issparse(counts) and counts.format == "csr" and counts.dtype in (np.float32, np.float64)
            and _flags_unknown(counts)):
        return counts, False
    return canonical_csr(counts), True


def upload_canonical(backend, counts, values_dtype=None):
    """Host matrix -> (host canonical CSR, DeviceCSR with its slab pointers)."""
    host, checked = canonical_csr_deferred(counts)
    if not checked and hasattr(backend, "with_slab_ptr"):
        from .._core.io import canonicalize

        X = backend.upload_csr(host.indptr, host.indices, host.data, host.shape, values_dtype=values_dtype,
                               slab_ptr=False)
        if getattr(canonicalize(backend, X), "canonical_as_given", False):
            host.has_canonical_format = True  # (verified entry by entry on the device: scipy need not scan it again)
            return host, backend.with_slab_ptr(X)
        del X  # unsorted rows or duplicates: the host route (it copies, sorts and sums like scipy)
    host = canonical_csr(counts)
    return host, backend.upload_csr(host.indptr, host.indices, host.data, host.shape, values_dtype=values_dtype)


def canonical_csr(counts) -> csr_matrix:
    """Host-side *layout* normalisation before upload (no arithmetic on values):
    CSR, duplicates summed, sorted column indices, int32 indices, int64 indptr, and the
    dtype promotion the reference's first statement performs (ints -> float64)."""
    if issparse(counts):
        m = counts.tocsr()
        if m is counts and not (m.dtype in (np.float32, np.float64) and m.has_canonical_format
                                and m.has_sorted_indices):
            m = m.copy()  # sum_duplicates / sort_indices below work in place: not on the caller's arrays
        # (already canonical float CSR: returned as is, no copy - at 6e9 entries the copy alone is
        #  75 GB of host memory and a single-threaded memcpy; callers only read it)
    else:
        m = csr_matrix(np.asarray(counts))  # dense branch (:97-99,113-114) ends in CSR too
    if m.dtype not in (np.float32, np.float64):
        m = m.astype(np.float64)  # 1.0 / n_peaks is float64 and promotes integer counts
    if not m.has_canonical_format:
        m.sum_duplicates()  # also sorts
    if not m.has_sorted_indices:
        m.sort_indices()
    return m


def tfidf_device(backend, X, n_obs, flags: int, scale: float, comm=None, out=None, emit_stream: bool = True):
    """Device-resident TF-IDF of a (row shard of a) CSR.  Returns a DeviceCSR that shares
    indptr / indices with ``X`` unless zeros had to be dropped.

    Pipeline: one reduction sweep (row sums, LDS-staged column sums), an all-reduce of the
    d column sums when rows are sharded, the idf vector, one fused scale pass."""
    from .._trace import phase

    comm = default_comm(comm)
    with phase("tfidf/sums"):
        rowsum, colsum = backend.row_col_sums(X)
        comm.all_reduce_sum(colsum)
        idf = backend.idf(colsum, float(n_obs), flags, X.values.dtype)
    # r05: the scale sweep also writes the ROW STREAM of the result - the operand layout of lsi's products - while it has
    # every entry in registers (the layout needs the row lengths only); lsi then skips its streaming copy of X and
    # transposes from the stream.  Operator sets without the kernel (CPU tests) simply do not offer it.
    emit = None
    can = getattr(backend, "can_emit_stream", None)
    with phase("tfidf/scale"):
        if emit_stream and can is not None and can(X):
            emit = backend.stream_layout(X)
            vals, zero_count = backend.tfidf_scale(X, rowsum, idf, scale, flags, out=out, emit=emit)
        else:
            vals, zero_count = backend.tfidf_scale(X, rowsum, idf, scale, flags, out=out)
    res = X.with_values(vals)
    take = getattr(backend, "take_slab_ptr", None)  # (a method: wrappers of the backend forward it)
    sp = take() if take is not None else None
    if int(zero_count.item()) != 0:
        # scipy's SpGEMM drops entries whose product is exactly 0
        res = backend.compact_nonzero(res)
    else:
        # the result shares X's index arrays: the slab pointers the sweeps searched go with it, lsi's transposition
        # cuts the same 8192-column slabs (csrc/tpack4.hip) and does not search them again
        if sp is not None and getattr(res, "slab_ptr", None) is None:  # (else: X came with its table, `with_values` kept it)
            res.slab_ptr = (sp, (res.indptr.data_ptr(), res.indices.data_ptr(), res.shape[0], res.shape[1]))
        if emit is not None:  # (keyed by the arrays it mirrors: a result whose zeros were compacted has no stream)
            res.xstream = (emit[0], emit[1], (res.indptr.data_ptr(), res.indices.data_ptr(), res.values.data_ptr(),
                                              res.shape[0], res.shape[1], res.nnz))
    return res


def _reverse_rows(m: csr_matrix) -> csr_matrix:
    """Emit every row in descending column order (what two scipy SpGEMMs + one sparse
    log1p leave behind for the default flags); data follows the indices."""
    indptr = m.indptr.astype(np.int64)
    nnz = m.nnz
    row_of = np.repeat(np.arange(m.shape[0], dtype=np.int64), np.diff(indptr))
    pos = np.arange(nnz, dtype=np.int64)
    perm = indptr[row_of] + (indptr[row_of + 1] - 1 - pos)
    out = csr_matrix((m.data[perm], m.indices[perm], m.indptr.copy()), shape=m.shape)
    out.has_sorted_indices = False
    return out


def tfidf(
    data,
    log_tf: bool = True,
    log_idf: bool = True,
    log_tfidf: bool = False,
    scale_factor: Union[int, float] = 1e4,
    inplace: bool = True,
    copy: bool = False,
    from_layer: Optional[str] = None,
    to_layer: Optional[str] = None,
    *,
    comm=None,
    n_obs: Optional[int] = None,
    match_scipy_order: bool = False,
    keep_on_device: bool = True,
    reuse_host: Optional[bool] = None,
    backend=None,
):
    """
    Transform peak counts with TF-IDF (Term Frequency - Inverse Document Frequency).

    TF: peak counts are normalised by total number of counts per cell
    DF: total number of counts for each peak
    IDF: number of cells divided by DF

    By default, log(TF) * log(IDF) is returned.  Signature, validation and write-back follow
    the reference (preproc.py:16-129); see the module docstring for the extra keywords.

    comm
            ``TorchDistComm`` when ``data`` holds this rank's row (cell) shard.
    n_obs
            Global number of cells when sharded (defaults to the sum over ranks).
    match_scipy_order
            Emit rows in the (descending) column order scipy's SpGEMM produces instead of
            canonical sorted CSR.
    keep_on_device
            Keep the device copy attached to the result so that ``lsi`` skips the upload.
    reuse_host
            Opt-in (default off; ``None`` reads ``MUON_AMD_REUSE_HOST=1``): when ``adata.X`` is rebound and NOTHING else
            holds a Python reference to the matrix it replaces or to its arrays, the result adopts that matrix's index
            arrays and value buffer instead of allocating 12 bytes per stored entry.  The reference never touches the
            old matrix (preproc.py:121-127) and a raw-pointer holder is invisible to reference counts: hence opt-in.
    """
    if is_anndata(data):
        adata = data
    elif is_mudata(data) and "atac" in data.mod:
        adata = data.mod["atac"]
    else:
        raise TypeError("Expected AnnData or MuData object with 'atac' modality")

    if log_tfidf and (log_tf or log_idf):
        raise AttributeError(
            "When returning log(TF*IDF), \
            applying neither log(TF) nor log(IDF) is possible."
        )

    if copy and not inplace:
        raise ValueError("`copy=True` cannot be used with `inplace=False`.")

    if to_layer is not None and not inplace:
        raise ValueError(f"`to_layer='{str(to_layer)}'` cannot be used with `inplace=False`.")

    if copy:
        adata = adata.copy()

    view_to_actual(adata)

    counts = adata.X if from_layer is None else adata.layers[from_layer]

    # Check before the computation
    if to_layer is not None and to_layer in adata.layers:
        warn(f"Existing layer '{str(to_layer)}' will be overwritten")

    if backend is None:
        from .._backend import get_backend

        backend = get_backend()  # raises without a GPU: there is no CPU fallback
    comm = default_comm(comm)

    X = resident(counts, backend)  # left on the device by binarize(): no second upload
    if X is not None:
        host = counts
    elif _is_canonical_csc(counts):
        # column-compressed input (.h5ad / .h5mu files may store X that way): its arrays are the CSR
        # of the transpose, which the device turns into the CSR of X instead of a
        # single-threaded scipy tocsr() on the host
        n_r, n_c = counts.shape
        Xc = backend.upload_csr(counts.indptr, counts.indices, counts.data, (n_c, n_r))
        # f32: the tile-staged transposition of csrc/tpack4.hip (3x the rate of the general kernel)
        fast = counts.dtype == np.float32 and hasattr(backend, "transpose_csr") and counts.nnz > 0
        X = backend.transpose_csr(Xc) if fast else backend.transpose(Xc)
        host = None
    else:
        host, X = upload_canonical(backend, counts)
    if n_obs is None:
        n_obs = comm.sum_scalar(adata.shape[0])
    flags = _flags(log_tf, log_idf, log_tfidf)
    # (the result's row stream - lsi's operand - is only worth its 8 bytes per entry when the device copy stays attached)
    R = tfidf_device(backend, X, n_obs, flags, _effective_scale(scale_factor), comm=comm,
                     emit_stream=keep_on_device and not (match_scipy_order and log_tf and not log_tfidf))

    shares = host is not None and R.indices is X.indices
    # the matrix this call replaces, when nothing else can see it (or a canonicalised temporary of this call): the
    # result takes its index arrays and is downloaded into its value array - see _dies_with_rebinding
    if shares and host is not counts:  # a canonicalised temporary (canonical_csr copies before it sorts / sums / widens)
        own = not any(np.may_share_memory(getattr(host, k), getattr(counts, k)) for k in ("data", "indices", "indptr")
                      if isinstance(getattr(counts, k, None), np.ndarray))
    else:
        own = (shares and inplace and to_layer is None and from_layer is None and not copy
               and _reuse_host_enabled(reuse_host) and _dies_with_rebinding(counts, 2))
    if own:
        want = np.dtype(np.float32) if R.values.element_size() == 4 else np.dtype(np.float64)
        buf = host.data if host.data.dtype == want else (host.data.view(want) if host.data.dtype.itemsize == want.itemsize
                                                         else None)
        vals = backend.to_host(R.values, out=buf) if buf is not None and buf.shape == (int(R.values.numel()),) else \
            backend.to_host(R.values)
        res = csr_matrix((vals, host.indices, host.indptr), shape=host.shape, copy=False)
        if host is counts:
            # the fingerprint of the device copy attached to the old matrix describes values that are gone
            try:
                delattr(counts, DEVICE_ATTR)
            except Exception:  # noqa: BLE001
                pass
    elif shares:
        vals = backend.to_host(R.values)
        res = csr_matrix((vals, _copy_array(host.indices), host.indptr.copy()), shape=host.shape)
    else:  # explicit zeros were dropped on the device, or the CSR only ever existed there
        vals = backend.to_host(R.values)
        ip = backend.to_host(R.indptr)
        if host is not None:
            ip = ip.astype(host.indptr.dtype)
        elif ip[-1] < 2**31:
            ip = ip.astype(np.int32)  # what scipy would have chosen
        res = csr_matrix((vals, backend.to_host(R.indices), ip), shape=tuple(counts.shape))
    res.has_sorted_indices = True
    if match_scipy_order and log_tf and not log_tfidf:
        res = _reverse_rows(res)
    if keep_on_device and not (match_scipy_order and log_tf and not log_tfidf):
        attach_device(res, R, backend)

    # res = np.nan_to_num(tf_idf, nan=0.0) is a no-op on sparse matrices (preproc.py:119)
    if not inplace:
        return res

    if to_layer is not None:
        adata.layers[to_layer] = res
    else:
        adata.X = res

    if copy:
        return adata


def binarize(data, *, backend=None):
    """
    Transform peak counts to the binary matrix (all the non-zero values become 1).
    Mirrors preproc.py:132-152: sparse X has its stored values set to 1 in place.
    """
    if is_anndata(data):
        adata = data
    elif is_mudata(data) and "atac" in data.mod:
        adata = data.mod["atac"]
    else:
        raise TypeError("Expected AnnData or MuData object with 'atac' modality")

    if backend is None:
        from .._backend import get_backend

        backend = get_backend()
    if issparse(adata.X):
        # Sparse matrix: stored values that are non-zero become 1, in place (:148-150)
        data = adata.X.data
        if data.dtype not in (np.float32, np.float64):
            data[data != 0] = 1  # integer counts: no floating-point kernel involved
            return
        m = adata.X
        if _is_canonical(m):
            # upload the whole CSR once and leave it resident for the tfidf() that follows
            Xd = backend.upload_csr(m.indptr, m.indices, data, m.shape)
            backend.binarize_values(Xd.values)
            data[...] = backend.to_host(Xd.values)
            attach_device(m, Xd, backend)
        else:
            vals = backend.to_device(data)
            backend.binarize_values(vals)
            data[...] = backend.to_host(vals)
    else:
        # dense input is a toy-size branch in the reference too (:151-152)
        adata.X[adata.X != 0] = 1
