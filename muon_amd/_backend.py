"""Device-side operator set used by the tfidf / lsi / mofa host code.

``HipBackend`` forwards every operator to a hand-written gfx950 kernel through the C-ABI
(``_ffi``).  Device memory, streams and collectives are PyTorch-ROCm plumbing: tensors are
only containers whose ``data_ptr()`` is handed to the kernels.  There is no CPU
implementation in this package - constructing ``HipBackend`` without a GPU raises.
(The multi-process *host logic* is exercised on CPU by tests that inject their own
operator set; see tests/cpu_backend.py.)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import contextlib
import os

import numpy as np
import torch

from . import _ffi
from ._ffi import F32, F64, check


@dataclass
class DeviceCSR:
    """CSR resident in HBM: indptr int64[n+1], indices int32[nnz], values f32|f64[nnz]."""

    indptr: torch.Tensor
    indices: torch.Tensor
    values: torch.Tensor
    shape: Tuple[int, int]

    @property
    def nnz(self) -> int:
        return int(self.indices.numel())

    @property
    def dtype(self):
        return self.values.dtype

    def with_values(self, values: torch.Tensor) -> "DeviceCSR":
        out = DeviceCSR(self.indptr, self.indices, values, self.shape)
        for name in ("slab_ptr", "xplan", "tplan", "wplan"):  # (they describe the index arrays, which the new object shares)
            got = getattr(self, name, None)
            if got is not None:
                setattr(out, name, got)
        return out


@dataclass
class DeviceStream:
    """Row stream of a CSR (include/muon_amd.h): the (column, value) pairs, 8 bytes each, row after
    row in the launch order of the SpMM - position p holds row perm[p] (-1 = none; None = identity)
    at ent[sptr[p] : sptr[p+1]], no padding; k = the row-sets-per-wave the layout was dealt for.
    SpMM-only."""

    sptr: torch.Tensor
    ent: torch.Tensor
    shape: Tuple[int, int]
    nnz: int
    perm: Optional[torch.Tensor] = None
    k: int = 0
    # fourth-generation transposition only: the count pass' prefix table (uint32 [(blocks + 1), rows of this stream]: entries
    # of a row in the source's row blocks before g) and the rows per block - what a ranged product on this stream reads
    t4: Optional[dict] = None

    @property
    def n_pos(self) -> int:
        return int(self.sptr.numel()) - 1


@dataclass
class SplitStream:
    """An f64-valued matrix as row streams: v = hi + lo with hi = fl32(v), lo = fl32(v - hi) (exact to
    2^-48); ``lo`` is None when every value is exact in f32 (counts, f32 input).  SpMM-only, against
    f64 dense blocks."""

    hi: DeviceStream
    lo: Optional[DeviceStream]

    @property
    def shape(self):
        return self.hi.shape


@dataclass
class DeviceEll:
    """Sliced-ELL operand of the narrow-block SpMM (csrc/spmm_ell.hip, include/muon_amd.h): rows in launch order
    (``perm``: position -> row, -1 none), groups of 16 positions (one wave each), columns in slabs of 1024;
    ``hdr[group, slab]`` = the windows (4 steps of 16 entries; 384 bytes: 64 f32 values | 64 u16 offsets) of the
    group in that slab;
    ``wave_base[group]`` = the group's first window in ``ent``.  SpMM-only, B = 16."""

    hdr: torch.Tensor
    wave_base: torch.Tensor
    ent: torch.Tensor
    perm: torch.Tensor
    shape: Tuple[int, int]
    nnz: int
    slots: int = 0
    waves: int = 15  # row-owning waves per workgroup of the launch (mu_spmm_ell16_waves; not part of the layout)
    slab_cols: int = 1024  # 1024: against f32 blocks (64-byte Q rows); 512: against f64 blocks (128-byte Q rows)


@dataclass
class SplitEll:
    """An f64-valued matrix as sliced-ELL operands for f64 blocks: v = hi + lo, hi = fl32(v), lo = fl32(v - hi);
    ``lo`` is None when every value is exact in f32.  SpMM-only."""

    hi: DeviceEll
    lo: Optional[DeviceEll]

    @property
    def shape(self):
        return self.hi.shape


_NULL_CTX = contextlib.nullcontext()


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise TypeError(f"unsupported value dtype {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def pick_block(width: int) -> int:
    for b in (16, 32, 64):
        if width <= b:
            return b
    raise NotImplementedError(
        f"block width {width} > 64 is not supported yet (n_comps + oversample must be <= 64)"
    )


def ell16_layout(X: DeviceCSR, waves: int = 15, slab_cols: int = 1024, slab_ptr_fn=None, fill_fn=None) -> "DeviceEll":
    """The sliced-ELL layout of csrc/spmm_ell.hip (built once per fit).  See DeviceEll / include/muon_amd.h for the
    format.  ``slab_cols`` = 1024 for f32 blocks, 512 for f64 blocks (a slab is 64 KiB of Q rows).  As tensor
    operations on X's device - the specification, and what runs without the library - or, with ``slab_ptr_fn`` and
    ``fill_fn`` (HipBackend.slab_ptr_width / ._ell16_fill), per (row, slab) only: the windows themselves are written by
    mu_ell16_fill (r04's dozen tensor passes over every entry took 30 ms per operand at 3.1e8 entries)."""
    assert slab_cols in (1024, 512) and X.values.dtype == torch.float32
    n, d = X.shape
    dev = X.indices.device
    S = -(-d // slab_cols)
    shift = 10 if slab_cols == 1024 else 9
    row_shift = 6 if slab_cols == 1024 else 7
    lens = X.indptr[1:] - X.indptr[:-1]
    order = torch.argsort(lens, descending=True, stable=True)       # position -> row: alike rows share a group
    n_groups = -(-n // 16)
    n_pos = n_groups * 16
    perm = torch.full((n_pos,), -1, dtype=torch.int32, device=dev)
    perm[:n] = order.to(torch.int32)
    inv = torch.empty((n,), dtype=torch.int64, device=dev)
    inv[order] = torch.arange(n, device=dev)
    sp = None
    if slab_ptr_fn is not None:
        # entries per (row, slab) off the slab pointers (binary searches: mu_csr_slab_ptr_width) - r04 histogrammed
        # every entry (torch.bincount: 20 ms of the 80 ms a fit's set-up took at 3.1e8 entries)
        sp = slab_ptr_fn(X, slab_cols).view(n, S + 1)
        cnt = torch.zeros((n_pos, S), dtype=torch.int64, device=dev)
        cnt[inv] = sp[:, 1:] - sp[:, :-1]
    else:
        rows = torch.repeat_interleave(torch.arange(n, device=dev), lens)
        pos = inv[rows]
        sl = (X.indices >> shift).to(torch.int64)
        cnt = torch.bincount(pos * S + sl, minlength=n_pos * S).view(n_pos, S)  # entries per (position, slab)
    nwin = (cnt.view(n_groups, 16, S).amax(dim=1) + 3) // 4                       # [group, slab]: the longest row
    flat = nwin.reshape(-1)
    wbase = torch.zeros(flat.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(flat, 0, out=wbase[1:])
    total = int(wbase[-1].item())
    win_base = wbase[:-1].view(n_groups, S)
    hdr = nwin.to(torch.int32).contiguous()
    wave_base = win_base[:, 0].contiguous()
    if sp is not None and fill_fn is not None:
        ent = torch.empty((total + 8, 384), dtype=torch.uint8, device=dev)        # (+ 8: the ring reads ahead)
        ent[total:].zero_()
        fill_fn(X, slab_cols, sp, perm, hdr, win_base.contiguous(), ent)
        return DeviceEll(hdr, wave_base, ent, perm, (n, d), X.nnz, total * 64, int(waves), int(slab_cols))
    if sp is not None:
        rows = torch.repeat_interleave(torch.arange(n, device=dev), lens)
        pos = inv[rows]
        sl = (X.indices >> shift).to(torch.int64)
    # rank of an entry inside its (row, slab): the rows are sorted by column, so slabs follow each other
    start = torch.cumsum(cnt, dim=1) - cnt                                        # [position, slab] exclusive
    e = torch.arange(X.nnz, device=dev)
    rank = e - X.indptr[:-1][rows] - start[pos, sl]
    dest = (win_base[pos // 16, sl] + rank // 4) * 64 + 4 * (pos % 16) + rank % 4
    vals = torch.zeros(((total + 8) * 64,), dtype=torch.float32, device=dev)      # (+ 8: the ring reads ahead)
    offs = torch.zeros(((total + 8) * 64,), dtype=torch.int16, device=dev)
    vals[dest] = X.values
    off = (X.indices.to(torch.int32) & (slab_cols - 1)) << row_shift              # u16 bit pattern in an int16
    offs[dest] = torch.where(off >= 32768, off - 65536, off).to(torch.int16)
    ent = torch.cat([vals.view(torch.uint8).view(-1, 256), offs.view(torch.uint8).view(-1, 128)], dim=1).contiguous()
    del vals, offs
    return DeviceEll(hdr, wave_base, ent, perm, (n, d), X.nnz, total * 64, int(waves), int(slab_cols))


class HipBackend:
    name = "hip"

    def __init__(self, device: Optional[int] = None):
        _ffi.require_gpu()
        self.lib = _ffi.lib()
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device))
        self._tri_index = {}  # (K, device) -> positions of the K x K entries in the packed upper triangle

    # -- plumbing -------------------------------------------------------------
    def _stream(self):
        """Raw handle of torch's current stream on this device (a few hundred of these per small lsi()
        call: the raw-handle query is ~10x cheaper than building a torch.cuda.Stream object)."""
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:
            return raw(self.device.index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def _dev_ctx(self):
        """Context that makes this backend's device current - a no-op object when it already is (the
        usual case: one process per GPU), instead of two device switches per kernel launch."""
        if torch.cuda.current_device() == self.device.index:
            return _NULL_CTX
        return torch.cuda.device(self.device)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def to_device(self, arr, dtype=None) -> torch.Tensor:
        """Host array -> device tensor (optionally converted to ``dtype`` on the way).  Big arrays -
        the CSR of a whole experiment is tens of GB - go through pinned staging buffers filled by a
        few host threads while the previous chunk is on the bus (scripts/probes/upload_probe.py);
        a pageable ``.to(device)`` stages the same bytes on one thread."""
        a = np.asarray(arr)
        want = np.dtype(dtype) if dtype is not None else a.dtype
        if a.nbytes >= self._UPLOAD_PIPELINE_MIN and a.ndim == 1:
            return self._upload_pipelined(a, want)
        a = np.ascontiguousarray(a, dtype=want)
        if 0 < a.nbytes <= self._UPLOAD_SMALL_MAX:
            return self._upload_small(a)
        return torch.as_tensor(a).to(self.device, non_blocking=False)

    _UPLOAD_SMALL_MAX = 1 << 20

    def _upload_small(self, a: np.ndarray) -> torch.Tensor:
        """Coefficient blocks and the like (the LSI loop uploads a few per expansion): through a ring of pinned
        buffers with a non-blocking copy.  A pageable ``.to(device)`` blocks the host until the stream has reached
        the copy - 0.5 ms each at 125k x 200k, eight per step, with the launches queued behind them starting late (r04)."""
        ring = self.__dict__.setdefault("_small_ring", {"bufs": [], "i": 0})
        if not ring["bufs"]:
            ring["bufs"] = [[torch.empty((self._UPLOAD_SMALL_MAX,), dtype=torch.uint8, pin_memory=True), None]
                            for _ in range(16)]
        slot = ring["bufs"][ring["i"] % len(ring["bufs"])]
        ring["i"] += 1
        if slot[1] is not None:
            slot[1].synchronize()  # the copy that used this buffer 16 uploads ago
        flat = slot[0][:a.nbytes]
        flat.numpy()[:] = a.reshape(-1).view(np.uint8)
        tdt = torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype
        out = torch.empty((a.size,), dtype=tdt, device=self.device)
        out.view(torch.uint8).copy_(flat, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        slot[1] = ev
        return out.reshape(a.shape)

    _UPLOAD_PIPELINE_MIN = 256 << 20
    _UPLOAD_CHUNK = 64 << 20      # bytes per staging buffer
    _UPLOAD_THREADS = 4

    def _upload_pipelined(self, a: np.ndarray, want: np.dtype) -> torch.Tensor:
        from concurrent.futures import ThreadPoolExecutor

        tdt = torch.from_numpy(np.empty(0, dtype=want)).dtype
        n = a.shape[0]
        out = torch.empty((n,), dtype=tdt, device=self.device)
        per = max(1, self._UPLOAD_CHUNK // want.itemsize)
        st = self.__dict__.get("_upload_state")
        if st is None:
            # staging is raw bytes: int32 / float32 / int64 uploads of one call sequence share the three
            # pinned buffers, the copy stream and the thread pool (they were keyed by dtype
            # and rebuilt - 3 x 64 MiB of pinning - whenever the dtype changed)
            bufs = [torch.empty((self._UPLOAD_CHUNK,), dtype=torch.uint8).pin_memory() for _ in range(3)]
            st = self._upload_state = {"bufs": bufs, "stream": torch.cuda.Stream(self.device),
                                       "pool": ThreadPoolExecutor(self._UPLOAD_THREADS)}
        copy_stream, pool = st["stream"], st["pool"]
        bufs = [b[: per * want.itemsize].view(tdt) for b in st["bufs"]]
        nps = [b.numpy() for b in bufs]
        events = [None, None, None]
        T = self._UPLOAD_THREADS
        cur = torch.cuda.current_stream(self.device)
        # `out` comes from the caching allocator on the CURRENT stream: the block may still be in use
        # by kernels queued there; the copies run on a private stream, so order them behind those
        copy_stream.wait_stream(cur)

        def fill(dst, src):
            np.copyto(dst, src, casting="unsafe")  # (memcpy or a converting loop; releases the GIL)

        for i, lo in enumerate(range(0, n, per)):
            hi = min(n, lo + per)
            b = i % 3
            if events[b] is not None:
                events[b].synchronize()  # the bus is done with this staging buffer
            m = hi - lo
            cuts = [m * t // T for t in range(T + 1)]
            list(pool.map(lambda t: fill(nps[b][cuts[t]:cuts[t + 1]], a[lo + cuts[t]:lo + cuts[t + 1]]), range(T)))
            with torch.cuda.stream(copy_stream):
                out[lo:hi].copy_(bufs[b][:m], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            events[b] = ev
        cur.wait_stream(copy_stream)
        out.record_stream(copy_stream)
        copy_stream.synchronize()  # the staging buffers are reused by the next call
        return out

    def to_host(self, t: torch.Tensor, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Device tensor -> host array.  Big 1-d tensors (the TF-IDF values of a whole experiment are GBs) come down
        through the pinned staging buffers of the upload path, chunk by chunk, a few host threads copying one chunk
        out while the next is on the bus: `.cpu()` on pageable memory ran at 11 GB/s in the API benchmark (r04,
        `bench.py --workload c3_api`)."""
        t = t.detach()
        if out is not None:
            # into the caller's array (pages that exist already: a fresh 6 GB array costs 225 ms of first-touch faults
            # on top of its bytes - scripts/probes/host_alloc_probe.py)
            want = torch.empty((0,), dtype=t.dtype).numpy().dtype
            if out.dtype != want or out.shape != tuple(t.shape) or not out.flags.c_contiguous or not out.flags.writeable:
                raise TypeError("to_host(out=): a writeable contiguous array of the tensor's shape and type")
        if t.ndim == 1 and t.is_cuda and t.is_contiguous() and t.numel() * t.element_size() >= self._UPLOAD_PIPELINE_MIN:
            return self._download_pipelined(t, out)
        if out is not None:
            np.copyto(out, t.cpu().numpy())
            return out
        return t.cpu().numpy()

    def _download_pipelined(self, t: torch.Tensor, out: Optional[np.ndarray] = None) -> np.ndarray:
        from concurrent.futures import ThreadPoolExecutor

        n = t.numel()
        if out is None:
            out = np.empty((n,), dtype=torch.empty((0,), dtype=t.dtype).numpy().dtype)
        item = t.element_size()
        per = max(1, self._UPLOAD_CHUNK // item)
        st = self.__dict__.get("_upload_state")
        if st is None:
            bufs = [torch.empty((self._UPLOAD_CHUNK,), dtype=torch.uint8).pin_memory() for _ in range(3)]
            st = self._upload_state = {"bufs": bufs, "stream": torch.cuda.Stream(self.device),
                                       "pool": ThreadPoolExecutor(self._UPLOAD_THREADS)}
        copy_stream, pool = st["stream"], st["pool"]
        bufs = [b[: per * item].view(t.dtype) for b in st["bufs"]]
        nps = [b.numpy() for b in bufs]
        T = self._UPLOAD_THREADS
        cur = torch.cuda.current_stream(self.device)
        copy_stream.wait_stream(cur)  # the tensor's producers are queued on the current stream
        chunks = [(lo, min(n, lo + per)) for lo in range(0, n, per)]
        events = [None] * len(chunks)

        def start(i):
            lo, hi = chunks[i]
            with torch.cuda.stream(copy_stream):
                bufs[i % 3][: hi - lo].copy_(t[lo:hi], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            events[i] = ev

        def drain(i):
            lo, hi = chunks[i]
            events[i].synchronize()
            m = hi - lo
            cuts = [m * k // T for k in range(T + 1)]
            list(pool.map(lambda k: np.copyto(out[lo + cuts[k]:lo + cuts[k + 1]], nps[i % 3][cuts[k]:cuts[k + 1]]), range(T)))

        for i in range(min(2, len(chunks))):  # two chunks on the bus / being copied out, a third buffer idle-safe
            start(i)
        for i in range(len(chunks)):
            drain(i)
            if i + 2 < len(chunks):
                start(i + 2)  # reuses the buffer of chunk i - 1, drained in the previous turn
        t.record_stream(copy_stream)
        return out

    def upload_csr(self, indptr, indices, values, shape, values_dtype=None, slab_ptr: bool = True) -> DeviceCSR:
        """Host CSR -> HBM.  Index arrays become int64 / int32 and the values ``values_dtype`` on the
        way (scipy keeps int64 column indices beyond 2^31 stored entries; a separate ``astype`` of
        such an array is a single-threaded pass over tens of GB)."""
        X = DeviceCSR(
            self.to_device(indptr, np.int64),
            self.to_device(indices, np.int32),
            self.to_device(values, values_dtype),
            (int(shape[0]), int(shape[1])),
        )
        return self.with_slab_ptr(X) if slab_ptr else X

    def with_slab_ptr(self, X: DeviceCSR) -> DeviceCSR:
        """Attach the slab pointers of X's index arrays (include/muon_amd.h mu_csr_slab_ptr): searched once where the
        device CSR is made - inside the clock of whoever makes it (upload, ingest) - and read by every sweep of tfidf and
        by lsi's transposition afterwards.  Needs sorted rows (every producer here uploads canonical CSR)."""
        n, d = X.shape
        if n == 0 or d == 0 or X.nnz == 0 or self._slab_ptr_of(X) is not None:
            return X
        sp = self.empty((n * (-(-d // 8192) + 1),), torch.int64)
        with self._dev_ctx():
            check(self.lib.mu_csr_slab_ptr(n, d, _p(X.indptr), _p(X.indices), _p(sp), self._stream()))
        X.slab_ptr = (sp, (X.indptr.data_ptr(), X.indices.data_ptr(), n, d))
        return self.with_plans(X)

    def with_plans(self, X: DeviceCSR) -> DeviceCSR:
        """The rest of what lsi's operand building derives from the INDEX ARRAYS alone, made once where the device CSR is
        made (r05; the same rule as the slab pointers: whoever makes the CSR - upload, ingest - pays inside its own clock):
        `xplan` - where the rows go in the row stream of X (launch_layout of the row lengths, scanned; what the TF-IDF
        scale sweep needs to write that stream) - and `tplan` - the transposition's count phase (entries per (row block,
        column), their prefixes, the column totals) and the layout of X^T's stream.  Keyed by the arrays they describe;
        a result whose zeros were compacted has other arrays and none of this."""
        n, d = X.shape
        if n == 0 or d == 0 or X.nnz == 0:
            return X
        if getattr(X, "xplan", None) is None:
            lens = X.indptr[1:] - X.indptr[:-1]
            perm, inv, K = self.launch_layout(lens, None)
            plens = torch.zeros((int(perm.numel()),), dtype=torch.int64, device=self.device)
            plens[inv.long()] = lens
            with self._dev_ctx():
                try:
                    sptr = self._stream_sptr(plens, K)
                except NotImplementedError:
                    return X
            X.xplan = (dict(perm=perm, inv=inv, K=K, sptr=sptr, row_dst=sptr[inv.long()].contiguous()),
                       (X.indptr.data_ptr(), n, d))
        if getattr(X, "tplan", None) is None and self._use_tpack4(X):
            col_nnz = self.empty((d,), torch.int64)
            wb = int(self.lib.mu_tpack4_worksize(n, d, X.nnz))
            work = self.empty((wb,), torch.uint8)
            with self._dev_ctx():
                check(self.lib.mu_tpack4_count(n, d, X.nnz, _p(X.indptr), _p(X.indices), _p(col_nnz), _p(work), wb,
                                               _p(self._slab_ptr_of(X)), self._stream()))
                perm, inv, K = self.launch_layout(col_nnz, None)
                plens = torch.zeros((int(perm.numel()),), dtype=torch.int64, device=self.device)
                plens[inv.long()] = col_nnz
                try:
                    sptr = self._stream_sptr(plens, K)
                except NotImplementedError:
                    return X
            X.tplan = (dict(work=work, wb=wb, perm=perm, inv=inv, K=K, sptr=sptr),
                       (X.indptr.data_ptr(), X.indices.data_ptr(), n, d, X.nnz))
        return X

    @staticmethod
    def _plan_of(X: DeviceCSR, name: str):
        got = getattr(X, name, None)
        if got is None:
            return None
        plan, key = got
        n, d = X.shape
        want = (X.indptr.data_ptr(), n, d) if name == "xplan" else (X.indptr.data_ptr(), X.indices.data_ptr(), n, d, X.nnz)
        return plan if key == want else None

    # -- TF-IDF (reference preproc.py:92-117) -----------------------------------------
    def row_col_sums(self, X: DeviceCSR):
        n, d = X.shape
        rowsum = self.empty((n,), torch.float64)
        colsum = self.empty((d,), torch.float64)
        wb = int(self.lib.mu_csr_row_col_sums_worksize(n, d))
        work = self.empty((wb,), torch.uint8)
        sp = self._slab_ptr_of(X)
        with self._dev_ctx():
            check(self.lib.mu_csr_row_col_sums_sp(_dt(X.values), n, d, _p(X.indptr), _p(X.indices),
                                                  _p(X.values), _p(rowsum), _p(colsum), _p(work), wb, _p(sp),
                                                  self._stream()))
        if sp is None:
            # the slab pointers at the head of `work` serve the scale pass of the same matrix
            self._sweep_work = (work, wb, (X.indptr.data_ptr(), X.indices.data_ptr(), n, d))
        return rowsum, colsum

    def idf(self, colsum: torch.Tensor, n_obs: float, flags: int, dtype) -> torch.Tensor:
        d = colsum.numel()
        out = self.empty((d,), dtype)
        with self._dev_ctx():
            check(self.lib.mu_tfidf_idf(_dt(out), d, float(n_obs), _p(colsum), flags, _p(out),
                                        self._stream()))
        return out

    def can_emit_stream(self, X: DeviceCSR) -> bool:
        """The TF-IDF scale sweep can write the row stream of its result (f32, a shape the stream SpMM takes)."""
        return (X.values.dtype == torch.float32 and X.nnz > 0 and self.can_stream(X, 64)
                and self.lib.mu_tune_get(b"scale_stream_off") != 1)

    def stream_layout(self, X: DeviceCSR, K: Optional[int] = None):
        """Where the rows of X go in its row stream (launch_layout: needs the row LENGTHS only, so it can be made before
        the values exist) and an empty stream to fill: (DeviceStream, row_dst) with row_dst[r] = pair index of row r's
        first pair."""
        n, d = X.shape
        plan = self._plan_of(X, "xplan") if K is None else None
        if plan is not None:  # (made with the device CSR: with_plans)
            ent = self.empty((max(X.nnz, 1),), torch.int64)
            return DeviceStream(plan["sptr"], ent, (n, d), X.nnz, plan["perm"], plan["K"]), plan["row_dst"]
        lens = X.indptr[1:] - X.indptr[:-1]
        perm, inv, K = self.launch_layout(lens, K)
        n_pos = int(perm.numel())
        plens = torch.zeros((n_pos,), dtype=torch.int64, device=self.device)
        plens[inv.long()] = lens
        with self._dev_ctx():
            sptr = self._stream_sptr(plens, K)
        ent = self.empty((max(X.nnz, 1),), torch.int64)
        row_dst = sptr[inv.long()].contiguous()
        return DeviceStream(sptr, ent, (n, d), X.nnz, perm, K), row_dst

    @staticmethod
    def _xstream_of(X: DeviceCSR):
        """(row stream, row_dst) the TF-IDF scale sweep left with X, if they describe X's arrays as they are."""
        got = getattr(X, "xstream", None)
        if got is None:
            return None
        xs, row_dst, key = got
        n, d = X.shape
        if key != (X.indptr.data_ptr(), X.indices.data_ptr(), X.values.data_ptr(), n, d, X.nnz):
            return None
        return xs, row_dst

    def tfidf_scale(self, X: DeviceCSR, rowsum, idf, scale: float, flags: int, out=None, emit=None):
        """``emit`` = (DeviceStream, row_dst) from stream_layout: the sweep also writes the result's row stream."""
        if out is None:
            out = torch.empty_like(X.values)
        zc = self.zeros((1,), torch.int64)
        n, d = X.shape
        kept = self.__dict__.pop("_sweep_work", None)
        key = (X.indptr.data_ptr(), X.indices.data_ptr(), n, d)
        sp = self._slab_ptr_of(X)
        if emit is not None:
            assert X.values.dtype == torch.float32
            have = kept is not None and kept[2] == key
            work, wb = None, 0
            if sp is None:
                if have:
                    work, wb = kept[0], kept[1]
                else:
                    wb = int(self.lib.mu_csr_row_col_sums_worksize(n, d))
                    work = self.empty((wb,), torch.uint8)
            with self._dev_ctx():
                check(self.lib.mu_tfidf_scale_sweep_stream(n, d, _p(X.indptr), _p(X.indices), _p(X.values), _p(rowsum),
                                                           _p(idf), float(scale), flags, _p(out), _p(zc), _p(work), wb,
                                                           int(have), _p(sp), _p(emit[1]), _p(emit[0].ent),
                                                           self._stream()))
            if sp is None:
                n_sp = n * (-(-d // 8192) + 1)
                self._last_slab_ptr = work[:8 * n_sp].view(torch.int64).clone()
            return out, zc
        if sp is not None and not self.__dict__.get("_scale_gather"):
            with self._dev_ctx():
                check(self.lib.mu_tfidf_scale_sweep_sp(_dt(X.values), n, d, _p(X.indptr), _p(X.indices), _p(X.values),
                                                       _p(rowsum), _p(idf), float(scale), flags, _p(out), _p(zc), _p(sp),
                                                       self._stream()))
            return out, zc
        with self._dev_ctx():
            if self.__dict__.get("_scale_gather"):  # comparison / tests: the per-lane gather kernel
                check(self.lib.mu_tfidf_scale(_dt(X.values), n, _p(X.indptr), _p(X.indices),
                                              _p(X.values), _p(rowsum), _p(idf), float(scale), flags,
                                              _p(out), _p(zc), self._stream()))
                return out, zc
            have = kept is not None and kept[2] == key
            if have:
                work, wb = kept[0], kept[1]
            else:
                wb = int(self.lib.mu_csr_row_col_sums_worksize(n, d))
                work = self.empty((wb,), torch.uint8)
            check(self.lib.mu_tfidf_scale_sweep(_dt(X.values), n, d, _p(X.indptr), _p(X.indices),
                                                _p(X.values), _p(rowsum), _p(idf), float(scale), flags,
                                                _p(out), _p(zc), _p(work), wb, int(have),
                                                self._stream()))
            # the slab pointers (head of the work buffer) also serve the transposition of the same index arrays
            # one call later (transpose_stream): kept apart from the column partials that follow them
            n_sp = n * (-(-d // 8192) + 1)
            self._last_slab_ptr = work[:8 * n_sp].view(torch.int64).clone()
        return out, zc

    def compact_nonzero(self, X: DeviceCSR) -> DeviceCSR:
        n = X.shape[0]
        row_nnz = self.empty((n,), torch.int64)
        new_indptr = self.empty((n + 1,), torch.int64)
        with self._dev_ctx():
            st = self._stream()
            check(self.lib.mu_csr_count_nonzero(_dt(X.values), n, _p(X.indptr), _p(X.values),
                                                _p(row_nnz), st))
            check(self.lib.mu_exclusive_scan_i64(n, _p(row_nnz), _p(new_indptr), st))
            new_nnz = int(new_indptr[-1].item())
            new_indices = self.empty((new_nnz,), torch.int32)
            new_values = self.empty((new_nnz,), X.values.dtype)
            check(self.lib.mu_csr_compact_nonzero(_dt(X.values), n, _p(X.indptr), _p(X.indices),
                                                  _p(X.values), _p(new_indptr), _p(new_indices),
                                                  _p(new_values), st))
        return DeviceCSR(new_indptr, new_indices, new_values, X.shape)

    def binarize_values(self, values: torch.Tensor) -> None:
        with self._dev_ctx():
            check(self.lib.mu_binarize_values(_dt(values), values.numel(), _p(values),
                                              self._stream()))

    # -- LSI building blocks (reference tools.py:53 -> scipy svds) ---------------------
    def take_slab_ptr(self):
        """The slab pointers the last TF-IDF scale sweep left behind (once; None if there are none)."""
        return self.__dict__.pop("_last_slab_ptr", None)

    @staticmethod
    def _slab_ptr_of(X: DeviceCSR):
        """The slab pointers the TF-IDF sweeps left with X (``tfidf_device``), if they describe X's index arrays."""
        got = getattr(X, "slab_ptr", None)
        if got is None:
            return None
        sp, key = got
        n, d = X.shape
        if key != (X.indptr.data_ptr(), X.indices.data_ptr(), n, d) or sp.numel() != n * (-(-d // 8192) + 1):
            return None
        return sp

    def transpose(self, X: DeviceCSR) -> DeviceCSR:
        n, d = X.shape
        nnz = X.nnz
        t_indptr = self.empty((d + 1,), torch.int64)
        t_indices = self.empty((nnz,), torch.int32)
        t_values = torch.empty_like(X.values)
        wb = int(self.lib.mu_csr_transpose_worksize(n, d, nnz))
        work = self.empty((wb,), torch.uint8)
        with self._dev_ctx():
            check(self.lib.mu_csr_transpose(_dt(X.values), n, d, nnz, _p(X.indptr), _p(X.indices),
                                            _p(X.values), _p(t_indptr), _p(t_indices),
                                            _p(t_values), _p(work), wb, self._stream()))
        return DeviceCSR(t_indptr, t_indices, t_values, (d, n))

    def transpose_csr(self, X: DeviceCSR) -> DeviceCSR:
        """CSR of X^T straight from the CSR of X (f32; stable: cells ascending inside every row =>
        canonical rows), through the tile-staged transposition of csrc/tpack4.hip."""
        n, d = X.shape
        assert X.values.dtype == torch.float32
        col_nnz = self.empty((max(d, 1),), torch.int64)
        t_indptr = self.zeros((d + 1,), torch.int64)
        t_indices = self.empty((max(X.nnz, 1),), torch.int32)[:X.nnz]
        t_values = self.empty((max(X.nnz, 1),), torch.float32)[:X.nnz]
        if self._use_tpack4(X):
            wb = int(self.lib.mu_tpack4_worksize(n, d, X.nnz))
            work = self.empty((wb,), torch.uint8)
            with self._dev_ctx():
                st = self._stream()
                check(self.lib.mu_tpack4_count(n, d, X.nnz, _p(X.indptr), _p(X.indices), _p(col_nnz), _p(work), wb,
                                               _p(self._slab_ptr_of(X)), st))
                check(self.lib.mu_exclusive_scan_i64(d, _p(col_nnz), _p(t_indptr), st))
                check(self.lib.mu_tpack4_fill_csr(n, d, X.nnz, _p(X.indptr), _p(X.indices), _p(X.values), None, None,
                                                  _p(t_indptr), _p(t_indices), _p(t_values), _p(work), wb, st))
            self._note_tpack4(work, n, d, X.nnz)
            self.raise_tpack4(self.take_tpack4_err().item())  # (set-up / ingest paths: a synchronisation is affordable)
            return DeviceCSR(t_indptr, t_indices, t_values, (d, n))
        # (2^31 rows, a row block of 2^29 entries: the general stable transposition, csrc/transpose.hip)
        return self.transpose(X)

    def can_stream(self, X: DeviceCSR, B: int) -> bool:
        """The row-stream SpMM exists for f32 values and B in (16, 32, 64)."""
        return X.values.dtype == torch.float32 and B in (16, 32, 64) and X.shape[0] > 0 and X.shape[1] > 0

    def _stream_sptr(self, lens_by_pos: torch.Tensor, K: int):
        n_pos = int(lens_by_pos.numel())
        sptr = self.zeros((n_pos + 1,), torch.int64)
        with self._dev_ctx():
            check(self.lib.mu_exclusive_scan_i64(n_pos, _p(lens_by_pos), _p(sptr), self._stream()))
        # cursors are 32-bit byte offsets from the first pair of the workgroup's 64 K rows
        per_wg = 64 * K
        span = sptr[per_wg::per_wg] - sptr[:-per_wg:per_wg] if n_pos > per_wg else sptr[-1:] - sptr[:1]
        if span.numel() and int(span.max().item()) * 8 >= (1 << 32) - 256:
            raise NotImplementedError("row stream: the rows of one workgroup span 4 GiB or more")
        return sptr

    def stream(self, X: DeviceCSR, sort_rows: bool = True, K: Optional[int] = None) -> DeviceStream:
        """Row stream of X for the SpMM of the iteration (a streaming copy, once per lsi call)."""
        n, d = X.shape
        assert X.values.dtype == torch.float32
        want_k = K
        perm, K, n_pos = None, max(1, int(want_k or self.lib.mu_spmm_stream_k(n))), n
        if sort_rows and n > 0:
            perm, _inv, K = self.launch_layout(X.indptr[1:] - X.indptr[:-1], want_k)
            n_pos = int(perm.numel())
        lens = self.empty((max(n_pos, 1),), torch.int64)[:n_pos]
        with self._dev_ctx():
            st = self._stream()
            check(self.lib.mu_csr_stream_len(n_pos, _p(perm), _p(X.indptr), _p(lens), st))
            sptr = self._stream_sptr(lens, K)
            ent = self.empty((max(X.nnz, 1),), torch.int64)
            check(self.lib.mu_csr_stream_fill(n_pos, _p(perm), _p(X.indptr), _p(X.indices), _p(X.values),
                                              _p(sptr), _p(ent), st))
        return DeviceStream(sptr, ent, (n, d), X.nnz, perm, K)

    def _use_tpack4(self, X: DeviceCSR) -> bool:
        """The tile-staged transposition (csrc/tpack4.hip) unless the shape needs the general one (2^31 rows, a row block
        of 2^29 entries; tune tpack4_off = 1 forces it: tests)."""
        n, d = X.shape
        return self.lib.mu_tune_get(b"tpack4_off") != 1 and bool(self.lib.mu_tpack4_supported(n, d, X.nnz))

    keep_tpack4_work = False  # tests / probes: keep the last fill's work buffer so that tpack4_status() can read its error word

    def _note_tpack4(self, work, n, d, nnz) -> None:
        # (never by default: a work buffer kept across calls is a second 1.8 GB block at 1e6 x 200k - the next call's
        #  allocation then misses the caching allocator's pool: a hipMalloc inside every step, measured at 300 ms)
        self._tpack4_work = (work, (n, d, nnz)) if self.keep_tpack4_work else None
        # the fill's error word (1: bitmap and count pass disagree, 2: a tile could not be narrowed - a
        # silently wrong X^T) was read by tests only.  A 4-byte copy of it, queued behind the fill, is picked up by the
        # consumer at its next synchronisation (lsi: with the first Gram fetch; transpose_csr: right away).
        off = int(self.lib.mu_tpack4_err_offset(n, d, nnz))
        self._tpack4_err = work[off:off + 4].view(torch.int32).clone()

    def take_tpack4_err(self):
        """Device int32[1] error word of the last fourth-generation fill (once; None if there was none)."""
        return self.__dict__.pop("_tpack4_err", None)

    @staticmethod
    def raise_tpack4(err: int) -> None:
        if int(err) != 0:
            raise _ffi.MuonAmdError(f"mu_tpack4_fill: the transposition tripped an internal invariant (error word {int(err)}: "
                                    "1 = bitmap and count pass disagree, 2 = a tile could not be narrowed); X^T would be wrong")

    def tpack4_status(self) -> int:
        """Error word of the last fourth-generation fill (0 = fine; synchronises: tests and probes)."""
        import ctypes as C

        got = self.__dict__.get("_tpack4_work")
        if got is None:
            return 0
        work, (n, d, nnz) = got
        err = C.c_int(0)
        check(self.lib.mu_tpack4_status(_p(work), n, d, nnz, C.byref(err)))
        return int(err.value)

    def transpose_stream(self, X: DeviceCSR, sort_rows: bool = True, before_fill=None,
                         K: Optional[int] = None, src=None) -> DeviceStream:
        """Row stream of X^T straight from X (no CSR of X^T; stable: cells ascending inside every row).
        ``src`` = (row stream of X, row_dst): the rows are read from the stream as contiguous pairs instead of from
        the CSR arrays (r05).  ``before_fill``: called once the count phase is done and before the fill is queued."""
        n, d = X.shape
        assert X.values.dtype == torch.float32
        col_nnz = self.empty((max(d, 1),), torch.int64)
        plan = self._plan_of(X, "tplan") if (sort_rows and K is None and self._use_tpack4(X)) else None
        if plan is not None:
            # count phase and layout were made with the device CSR (with_plans): the fill is all that is left
            ent = self.empty((max(X.nnz, 1),), torch.int64)
            if before_fill is not None:
                before_fill()
            xs_ent, xs_dst = (src[0].ent, src[1]) if src is not None else (None, None)
            with self._dev_ctx():
                check(self.lib.mu_tpack4_fill_stream(n, d, X.nnz, _p(X.indptr), _p(X.indices), _p(X.values),
                                                     _p(xs_dst), _p(xs_ent), _p(plan["sptr"]), _p(plan["inv"]), _p(ent),
                                                     _p(plan["work"]), plan["wb"], self._stream()))
            self._note_tpack4(plan["work"], n, d, X.nnz)
            return DeviceStream(plan["sptr"], ent, (d, n), X.nnz, plan["perm"], plan["K"], self._t4_prefix(plan["work"], n, d, X.nnz))
        if self._use_tpack4(X):
            wb = int(self.lib.mu_tpack4_worksize(n, d, X.nnz))
            work = self.empty((wb,), torch.uint8)
            with self._dev_ctx():
                st = self._stream()
                check(self.lib.mu_tpack4_count(n, d, X.nnz, _p(X.indptr), _p(X.indices), _p(col_nnz), _p(work), wb,
                                               _p(self._slab_ptr_of(X)), st))
                want_k = K
                perm, inv, K, n_pos = None, None, max(1, int(want_k or self.lib.mu_spmm_stream_k(d))), d
                lens = col_nnz[:d]
                if sort_rows and d > 0:
                    perm, inv, K = self.launch_layout(lens, want_k)
                    n_pos = int(perm.numel())
                    plens = torch.zeros((n_pos,), dtype=torch.int64, device=self.device)
                    plens[inv.long()] = lens
                else:
                    plens = lens.contiguous()
                sptr = self._stream_sptr(plens, K)
                ent = self.empty((max(X.nnz, 1),), torch.int64)
                if before_fill is not None:
                    before_fill()
                xs_ent, xs_dst = (src[0].ent, src[1]) if src is not None else (None, None)
                check(self.lib.mu_tpack4_fill_stream(n, d, X.nnz, _p(X.indptr), _p(X.indices), _p(X.values),
                                                     _p(xs_dst), _p(xs_ent), _p(sptr), _p(inv), _p(ent), _p(work), wb, st))
            self._note_tpack4(work, n, d, X.nnz)
            return DeviceStream(sptr, ent, (d, n), X.nnz, perm, K, self._t4_prefix(work, n, d, X.nnz))
        # (shapes the tile-staged transposition refuses: the CSR of X^T from the general kernel, then a streaming copy)
        if before_fill is not None:
            before_fill()
        return self.stream(self.transpose(X), sort_rows=sort_rows, K=K)

    def _t4_geometry(self, n, d, nnz):
        import ctypes as C

        rpb, G, Ct = C.c_int64(0), C.c_int(0), C.c_int(0)
        check(self.lib.mu_tpack4_geometry(n, d, nnz, C.byref(rpb), C.byref(G), C.byref(Ct)))
        return int(rpb.value), int(G.value)

    def _t4_prefix(self, work, n, d, nnz):
        rpb, G = self._t4_geometry(n, d, nnz)
        off = int(self.lib.mu_tpack4_cnt_offset(n, d, nnz))
        # (a view: the work buffer lives as long as the stream that carries it)
        cnt = work[off:off + 4 * (G + 1) * d].view(torch.int32)
        return dict(cnt=cnt, rpb=rpb, blocks=G, stride=d)

    # -- the cell slice of lsi's warm start (r06: no operands of its own) --------------------------------
    def slice_plan(self, X: DeviceCSR, n_s: int, max_ranges: int = 16):
        """Which cells the warm start's slice holds: <= 16 ranges of whole ROW BLOCKS of the transposition (so that the
        slice's cells are a contiguous piece of every row of X^T, found through the count pass' prefix table), spread
        evenly over this shard's cells (files list cells sample by sample: the first cells may be one batch), about
        ``n_s`` cells together.  Host integers only - the ranges and where they lie in the CSR (one read of <= 32 row
        pointers) - cached with the index arrays they describe, like the other plans: a second call reads nothing from
        the device.  None when the shape does not take the ranged products (no fourth-generation transposition, no slab
        pointers)."""
        n, d = X.shape
        if n_s <= 0 or X.values.dtype != torch.float32 or not self._use_tpack4(X) or self._slab_ptr_of(X) is None:
            return None
        key = (X.indptr.data_ptr(), X.indices.data_ptr(), n, d, X.nnz, int(n_s), int(max_ranges))
        # (kept in the transposition plan's dict when there is one: `with_values` hands that dict - by reference - from
        #  the counts to every TF-IDF result, so the plan made in the first call serves the later ones)
        shared = self._plan_of(X, "tplan")
        got = shared.get("wplan") if shared is not None else getattr(X, "wplan", None)
        if got is not None and got[1] == key:
            return got[0]
        rpb, G = self._t4_geometry(n, d, X.nnz)
        if G < 1:
            return None
        blocks = max(1, min(G, int(round(n_s / rpb))))
        R = max(1, min(int(max_ranges), blocks))
        per = max(1, min(int(round(blocks / R)), G // R))  # (blocks per range; ranges G // R blocks apart: no overlap)
        g0s = [(r * G) // R for r in range(R)]
        rows = [(g0 * rpb, min((g0 + per) * rpb, n)) for g0 in g0s]
        edges = X.indptr[torch.tensor([v for ab in rows for v in ab], device=X.indptr.device)].tolist()
        plan = dict(rpb=rpb, ranges=[dict(g0=g0, g1=g0 + per, row0=a, row1=b, lo=int(edges[2 * i]), hi=int(edges[2 * i + 1]))
                                     for i, (g0, (a, b)) in enumerate(zip(g0s, rows))])
        plan["n_s"] = sum(r["row1"] - r["row0"] for r in plan["ranges"])
        plan["nnz_s"] = sum(r["hi"] - r["lo"] for r in plan["ranges"])
        if shared is not None:
            shared["wplan"] = (plan, key)
        else:
            X.wplan = (plan, key)
        return plan

    def slice_stream(self, X: DeviceCSR, plan) -> dict:
        """Compact row stream of the slice's rows (mu_csr_slice_stream: range after range, rows in their own order) with
        the table of its 8192-column super-slab boundaries - the operand of X_S Q (spmm_slice)."""
        import ctypes as C

        n, d = X.shape
        rg = plan["ranges"]
        n_s, nnz_s = plan["n_s"], plan["nnz_s"]
        S1 = -(-d // 8192) + 1
        sptr = self.empty((n_s + 1,), torch.int64)
        ent = self.empty((max(nnz_s, 1),), torch.int64)
        rel = self.empty((S1 * max(n_s, 1),), torch.int32)
        arr = lambda key, f=None: (C.c_int64 * len(rg))(*[(f(r) if f else r[key]) for r in rg])  # noqa: E731
        with self._dev_ctx():
            check(self.lib.mu_csr_slice_stream(len(rg), arr("row0"), arr(None, lambda r: r["row1"] - r["row0"]), arr("lo"),
                                               arr("hi"), d, _p(X.indptr), _p(X.indices), _p(X.values),
                                               _p(self._slab_ptr_of(X)), _p(sptr), _p(ent), _p(rel), self._stream()))
        # K and the split of the column super-slabs over blockIdx.y: enough workgroups for two rounds of the chip
        n_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        K = 8
        while K > 1 and -(-n_s // (64 * K)) * (S1 - 1) < 2 * n_cus:
            K //= 2
        wgs = -(-n_s // (64 * K))
        ny = max(1, min(S1 - 1, 16, -(-2 * n_cus // max(wgs, 1))))  # (<= 16 column groups: the launch takes 32 ranges)
        group = -(-(S1 - 1) // ny)  # consecutive super-slabs of one blockIdx.y: one contiguous column range
        return dict(sptr=sptr, ent=ent, rel=rel, n_s=n_s, nnz_s=nnz_s, K=K, group=group, d=d, S=S1 - 1)

    def spmm_slice(self, S: dict, Q: torch.Tensor) -> torch.Tensor:
        """Y_S = X_S Q [n_s x 64] on the compact slice stream, column super-slabs split over the chip, partial products
        summed in fixed order."""
        import ctypes as C

        d, n_s = S["d"], S["n_s"]
        assert Q.shape == (d, 64) and Q.dtype == torch.float32 and Q.is_contiguous()
        ns, g = S["S"], S["group"]
        ny = -(-ns // g)
        part = self.empty((ny, n_s, 64), torch.float32)
        h = (C.c_int32 * (5 * ny))()
        for r in range(ny):  # (the super-slabs r g .. (r + 1) g - 1 are contiguous in the columns and in every row)
            h[5 * r:5 * r + 5] = [r * g * 8192, min((r + 1) * g * 8192, d), r * g * 8192, r * g, min((r + 1) * g, ns)]
        with self._dev_ctx():
            check(self.lib.mu_spmm_stream_ranges_f32(n_s, _p(S["sptr"]), _p(S["ent"]), None, S["K"], _p(Q), d, _p(part),
                                                     n_s * 64, _p(S["rel"]), n_s, ny, h, 1, self._stream()))
        return part[0] if ny == 1 else part.sum(dim=0)

    def spmm_slice_t(self, Xt: DeviceStream, plan, Ys: torch.Tensor) -> torch.Tensor:
        """Z = X_S^T Y_S [d x 64] on the row stream of X^T as it is: the slice's cells are <= 16 contiguous pieces of
        every row, found through the transposition's count prefixes (Xt.t4)."""
        import ctypes as C

        d, n = Xt.shape
        t4, rg = Xt.t4, plan["ranges"]
        assert t4 is not None and t4["rpb"] == plan["rpb"] and Ys.shape == (plan["n_s"], 64) and Ys.is_contiguous()
        Z = self.empty((d, 64), torch.float32)
        h = (C.c_int32 * (5 * len(rg)))()
        q_off = 0
        for i, r in enumerate(rg):
            h[5 * i:5 * i + 5] = [r["row0"], r["row1"], q_off, r["g0"], r["g1"]]
            q_off += r["row1"] - r["row0"]
        with self._dev_ctx():
            check(self.lib.mu_spmm_stream_ranges_f32(Xt.n_pos, _p(Xt.sptr), _p(Xt.ent), _p(Xt.perm), Xt.k, _p(Ys),
                                                     plan["n_s"], _p(Z), 0, _p(t4["cnt"]), t4["stride"], len(rg), h,
                                                     len(rg), self._stream()))
        return Z

    def split_streams(self, X: DeviceCSR):
        """(row streams of X, row streams of X^T) for an f64-valued CSR: see SplitStream."""
        assert X.values.dtype == torch.float64
        hi = X.values.to(torch.float32)
        rest = X.values - hi.to(torch.float64)
        Xh = X.with_values(hi)
        s_hi, t_hi = self.stream(Xh), self.transpose_stream(Xh)
        if bool((rest != 0).any().item()):
            Xl = X.with_values(rest.to(torch.float32))
            return SplitStream(s_hi, self.stream(Xl)), SplitStream(t_hi, self.transpose_stream(Xl))
        return SplitStream(s_hi, None), SplitStream(t_hi, None)

    def launch_layout(self, lens: torch.Tensor, K: Optional[int] = None):
        """Where the rows go in a row stream (include/muon_amd.h): sorted by length (descending,
        stable) and dealt round robin - row-set q of the sorted order goes to workgroup q % n_wg,
        inside it to wave (q // n_wg) % 16 and row-set slot (q // n_wg) // 16 - so that the four rows
        a wave advances in lock step have similar lengths and every workgroup and wave gets the same
        mix.  Returns (perm int32[n_pos], inv int32[n], K)."""
        n = int(lens.numel())
        K = max(1, int(K or self.lib.mu_spmm_stream_k(n)))
        W = 16  # waves of a workgroup
        per_wg = 4 * W * K
        n_wg = max(1, (n + per_wg - 1) // per_wg)
        # One workgroup per CU runs at a time (128 KiB of LDS) and the dealt workgroups take equally
        # long, so the launch proceeds in rounds of n_cus workgroups: a last round that is 3/4 empty
        # costs a full one.  Dealing the same rows over a whole number of rounds (the last row-set
        # slots of every wave stay empty instead) makes every round shorter: Xt*Y at 200k rows
        # 447 -> 512 workgroups, X*Q at 1e6 rows 1954 -> 2048.
        n_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        if n_wg > n_cus and not self.__dict__.get("_no_round_fill"):
            n_wg = -(-n_wg // n_cus) * n_cus
        elif 5 * n_wg >= 4 * n_cus and not self.__dict__.get("_no_round_fill"):
            # a little less than one round: every CU takes a workgroup (100 000 rows at K = 7: 224 -> 256
            # workgroups of 391 rows; empty row slots are cheap).  A/B on one box: MOFA c4 -1 %, c3shard -0.3 %;
            # at 157 of 256 workgroups (10k x 30k) it cost 1 %: hence the 4 / 5 threshold.
            n_wg = n_cus
        n_pos = n_wg * per_wg
        order = torch.argsort(lens, descending=True, stable=True)
        i = torch.arange(n, device=lens.device)
        q, j = i // 4, i % 4
        b, t = q % n_wg, q // n_wg
        w, k = t % W, t // W
        pos = ((b * W + w) * K + k) * 4 + j
        perm = torch.full((n_pos,), -1, dtype=torch.int32, device=lens.device)
        perm[pos] = order.to(torch.int32)
        inv = torch.empty((n,), dtype=torch.int32, device=lens.device)
        inv[order] = pos.to(torch.int32)
        return perm, inv, K

    # -- sliced-ELL operand of the narrow-block SpMM (csrc/spmm_ell.hip) ---------------------------
    def ell16(self, X: DeviceCSR, wide: bool = False):
        """Lay a canonical CSR out for mu_spmm_ell16_f32 / _f64 (once per fit: the operand of MOFA's sparse views is
        multiplied hundreds of times).  f32 values -> DeviceEll; ``wide`` (against f64 blocks): 512-column slabs,
        and f64 values -> SplitEll (hi + lo)."""
        waves = int(self.lib.mu_spmm_ell16_waves(X.shape[0]))
        cols = 512 if wide else 1024
        fill = self._ell16_fill
        if X.values.dtype == torch.float32:
            return ell16_layout(X, waves, cols, self.slab_ptr_width, fill)
        assert wide and X.values.dtype == torch.float64
        hi = X.values.to(torch.float32)
        rest = X.values - hi.to(torch.float64)
        e_hi = ell16_layout(X.with_values(hi), waves, cols, self.slab_ptr_width, fill)
        if bool((rest != 0).any().item()):
            return SplitEll(e_hi, ell16_layout(X.with_values(rest.to(torch.float32)), waves, cols, self.slab_ptr_width, fill))
        return SplitEll(e_hi, None)

    def ell16_pair(self, X: DeviceCSR, wide: bool = False):
        """(layout of X, layout of X^T) - both operands of a fit.  The transposed CSR comes from the tile-staged
        transposition (f32 values; an f64-valued matrix is transposed as its hi and lo parts, the second only when
        some value is not exact in f32) instead of the general kernel behind `transpose`."""
        if X.values.dtype == torch.float32:
            return self.ell16(X, wide), self.ell16(self.transpose_csr(X), wide)
        assert wide and X.values.dtype == torch.float64
        hi = X.values.to(torch.float32)
        rest = X.values - hi.to(torch.float64)
        parts = [hi] + ([rest.to(torch.float32)] if bool((rest != 0).any().item()) else [])
        del rest
        fw = [self.ell16(X.with_values(p), True) for p in parts]
        tr = [self.ell16(self.transpose_csr(X.with_values(p)), True) for p in parts]
        return SplitEll(fw[0], fw[1] if len(fw) > 1 else None), SplitEll(tr[0], tr[1] if len(tr) > 1 else None)

    def col_moments(self, Y: torch.Tensor, a: int, b: int):
        """(sum, sum of squares) per column over rows a .. b-1 of a dense row-major f32 / f64 block: f64 sums, one pass."""
        assert Y.dim() == 2 and Y.is_contiguous() and Y.dtype in (torch.float32, torch.float64)
        D = int(Y.shape[1])
        chunks = max(1, int(self.lib.mu_dense_col_moments_chunks(b - a, D)))
        part = self.empty((chunks, 2, D), torch.float64)
        with self._dev_ctx():
            check(self.lib.mu_dense_col_moments(_dt(Y),
                                                int(a), int(b), D, _p(Y), chunks, _p(part), self._stream()))
        s = part.sum(dim=0)
        return s[0], s[1]

    def _ell16_fill(self, X, slab_cols, sp, perm, hdr, win_base, ent):
        with self._dev_ctx():
            check(self.lib.mu_ell16_fill(int(perm.numel()) // 16, X.shape[1], X.nnz, int(slab_cols), _p(X.indices), _p(X.values),
                                         _p(sp), _p(perm), _p(hdr), _p(win_base), _p(ent), self._stream()))

    def slab_ptr_width(self, X: DeviceCSR, width: int) -> torch.Tensor:
        """First entry of every row at or behind every multiple of ``width`` columns (+ the row's end)."""
        n, d = X.shape
        sp = self.empty((n * (-(-d // width) + 1),), torch.int64)
        with self._dev_ctx():
            check(self.lib.mu_csr_slab_ptr_width(n, d, int(width), _p(X.indptr), _p(X.indices), _p(sp), self._stream()))
        return sp

    def spmm_ell(self, E: DeviceEll, Q: torch.Tensor, out=None, accumulate: bool = False) -> torch.Tensor:
        n, d = E.shape
        wide = Q.dtype == torch.float64
        if Q.shape != (d, 16) or Q.dtype not in (torch.float32, torch.float64) or not Q.is_contiguous():
            raise TypeError("the sliced-ELL SpMM needs a contiguous f32 / f64 block of 16 columns")
        if E.slab_cols != (512 if wide else 1024):
            raise TypeError("sliced-ELL operand laid out for the other block type (slab width)")
        if accumulate and not wide:
            raise TypeError("accumulating sliced-ELL products exist for f64 blocks only")
        if out is None:
            assert not accumulate
            out = self.empty((n, 16), Q.dtype)
        # r06: an operand of a few thousand rows (one rank's shard of a sharded fit) is a few dozen workgroups that each
        # pull all of Q through their LDS - its column slabs are split over blockIdx.y, the partial products summed in
        # order (mu_spmm_ell16_parts: one part for every shape that fills a round of the chip by its rows)
        import ctypes as C

        wv, parts = C.c_int(0), C.c_int(1)
        check(self.lib.mu_spmm_ell16_parts(n, d, int(wide), C.byref(wv), C.byref(parts)))
        if parts.value > 1:
            slabs = -(-d // E.slab_cols)
            ny = -(-slabs // -(-slabs // parts.value))
            part = self.empty((ny, n, 16), Q.dtype)
            fn = self.lib.mu_spmm_ell16_parts_f64 if wide else self.lib.mu_spmm_ell16_parts_f32
            with self._dev_ctx():
                check(fn(wv.value, parts.value, int(E.perm.numel()), d, _p(E.hdr), _p(E.wave_base), _p(E.ent), _p(E.perm),
                         _p(Q), _p(part), n * 16, self._stream()))
            if accumulate:
                out += part.sum(dim=0)
            else:
                torch.sum(part, dim=0, out=out)
            return out
        with self._dev_ctx():
            if wide:
                check(self.lib.mu_spmm_ell16_f64(E.waves, int(E.perm.numel()), d, _p(E.hdr), _p(E.wave_base),
                                                 _p(E.ent), _p(E.perm), _p(Q), _p(out), int(bool(accumulate)),
                                                 self._stream()))
            else:
                check(self.lib.mu_spmm_ell16_f32(E.waves, int(E.perm.numel()), d, _p(E.hdr), _p(E.wave_base),
                                                 _p(E.ent), _p(E.perm), _p(Q), _p(out), self._stream()))
        return out

    def stream_both(self, X: DeviceCSR):
        """(row stream of X, row stream of X^T).  The two builders are independent; the streaming
        copy of X (HBM bound) runs on a second stream under the transposition, whose fill is
        instruction bound and leaves half of every CU's wave slots free."""
        have = self._xstream_of(X)
        if have is not None and self._use_tpack4(X):
            # r05: the TF-IDF scale sweep wrote X's stream already; the transposition reads its rows from there
            Xs, row_dst = have
            return Xs, self.transpose_stream(X, src=(Xs, row_dst))
        cur = torch.cuda.current_stream(self.device)
        side = self.__dict__.get("_side_stream")
        if side is None:
            side = self._side_stream = torch.cuda.Stream(self.device)
        got = []

        def start_copy():
            # right before the fill: the count phase of the transposition (binary searches and a
            # histogram, latency bound) slowed down 4x next to the streaming copy, the fill does not
            side.wait_stream(cur)
            wg = 2  # (A/B on one box, c3: 32 -> 451, 4 -> 449, 2 -> 441 ms per step)
            with torch.cuda.stream(side):
                self.tune("pack_wg", wg)  # few workgroups per CU: the fill's 1024-thread groups must fit next to them
                # ... and the unpipelined loop: the pipelined copy is faster alone (22.5 -> 20-21.6 ms at 1e6 x 200k) and
                # costs the fill more next to it (both together 87 -> 95 ms; scripts/probes/stream_pipe_probe.py)
                self.tune("stream_pipe", 1)
                try:
                    got.append(self.stream(X))
                finally:
                    self.tune("pack_wg", 0)
                    self.tune("stream_pipe", 0)

        Xt = self.transpose_stream(X, before_fill=start_copy)
        Xs = got[0]
        cur.wait_stream(side)
        for t in (Xs.sptr, Xs.ent, Xs.perm):
            if t is not None:
                t.record_stream(cur)
        return Xs, Xt

    def fetch_async(self, tensors):
        """Start device -> host copies of small tensors into pinned staging buffers; ``wait()`` on
        the returned handle blocks only until these copies are done - work queued on the stream
        afterwards keeps running (the host-side Ritz step of the LSI hides under the next SpMM)."""
        # a staging buffer belongs to ONE handle until its wait() has copied the data out (several fetches can be in
        # flight: the pipelined LSI loop starts the next block's fetch before it reads the current one)
        pool = self.__dict__.setdefault("_pinned_free", {})
        bufs = []
        for t in tensors:
            free = pool.setdefault((tuple(t.shape), t.dtype), [])
            buf = free.pop() if free else torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            bufs.append(buf)
            buf.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return _Fetch(bufs, ev, pool)

    def tune(self, key: str, value: int) -> None:
        check(self.lib.mu_tune_set(key.encode(), int(value)))

    def spmm(self, X, Q: torch.Tensor, out=None, accumulate: bool = False) -> torch.Tensor:
        n, d = X.shape
        B = Q.shape[1]
        assert Q.shape[0] == d and Q.dtype in (torch.float32, torch.float64) and Q.is_contiguous()
        if isinstance(X, SplitStream):
            # f64 values as two f32 streams: X Q = X_hi Q + X_lo Q, accumulated in f64
            out = self.spmm(X.hi, Q, out=out)
            if X.lo is not None:
                self.spmm(X.lo, Q, out=out, accumulate=True)
            return out
        if isinstance(X, SplitEll):
            out = self.spmm_ell(X.hi, Q, out=out, accumulate=accumulate)
            if X.lo is not None:
                self.spmm_ell(X.lo, Q, out=out, accumulate=True)
            return out
        if isinstance(X, DeviceEll):
            return self.spmm_ell(X, Q, out=out, accumulate=accumulate)
        if isinstance(X, DeviceStream):
            wide = Q.dtype == torch.float64
            if B not in ((16, 32) if wide else (16, 32, 64)):
                raise TypeError("the row-stream SpMM needs a dense block of width 16, 32 or 64 (f64: 16 or 32)")
            if accumulate and not wide:
                raise TypeError("accumulating row-stream products exist for f64 blocks only")
            if out is None:
                assert not accumulate
                out = self.empty((n, B), Q.dtype)
            with self._dev_ctx():
                if wide:
                    check(self.lib.mu_spmm_stream_f64(X.n_pos, d, _p(X.sptr), _p(X.ent), _p(X.perm), X.k,
                                                      _p(Q), B, _p(out), int(bool(accumulate)), self._stream()))
                else:
                    check(self.lib.mu_spmm_stream_f32(X.n_pos, d, _p(X.sptr), _p(X.ent), _p(X.perm), X.k,
                                                      _p(Q), B, _p(out), self._stream()))
            return out
        if X.values.dtype != Q.dtype:
            raise TypeError("spmm needs values and dense block of one dtype")
        if out is None:
            out = self.empty((n, B), Q.dtype)
        fn = self.lib.mu_spmm_f32 if Q.dtype == torch.float32 else self.lib.mu_spmm_f64
        with self._dev_ctx():
            check(fn(n, d, _p(X.indptr), _p(X.indices), _p(X.values), _p(Q), B, _p(out), 0,
                     self._stream()))
        return out

    def gram(self, A: torch.Tensor):
        n, B = A.shape
        assert A.dtype == torch.float32 and A.is_contiguous()
        G = self.empty((B, B), torch.float64)
        cs = self.empty((B,), torch.float64)
        wb = int(self.lib.mu_gram_worksize(n, B))
        work = self.empty((wb,), torch.uint8)
        with self._dev_ctx():
            check(self.lib.mu_gram_f32(n, B, _p(A), _p(G), _p(cs), _p(work), wb, self._stream()))
        return G, cs

    def gram_cross(self, A: torch.Tensor, Bm: torch.Tensor) -> torch.Tensor:
        """C = A^T Bm in f64 (A, Bm n x B f32)."""
        n, B = A.shape
        assert Bm.shape == A.shape and A.dtype == torch.float32 and Bm.dtype == torch.float32
        assert A.is_contiguous() and Bm.is_contiguous()
        Cm = self.empty((B, B), torch.float64)
        wb = int(self.lib.mu_gram_worksize(n, B))
        work = self.empty((wb,), torch.uint8)
        with self._dev_ctx():
            check(self.lib.mu_gram_cross_f32(n, B, _p(A), _p(Bm), _p(Cm), _p(work), wb, self._stream()))
        return Cm

    def apply(self, A: torch.Tensor, M: torch.Tensor, bias=None, out=None) -> torch.Tensor:
        n, B = A.shape
        assert M.shape == (B, B) and M.dtype == torch.float32 and M.is_contiguous()
        if out is None:
            out = torch.empty_like(A)
        with self._dev_ctx():
            check(self.lib.mu_dense_apply_f32(n, B, _p(A), _p(M), _p(bias), _p(out), self._stream()))
        return out

    def project_out_block(self, Q: torch.Tensor, C: torch.Tensor, Z: torch.Tensor) -> torch.Tensor:
        """Z -= Q C in place (C: the f64 B x B coefficients Q^T Z of gram_cross)."""
        n, B = Q.shape
        assert Z.shape == Q.shape and C.shape == (B, B) and C.dtype == torch.float64 and C.is_contiguous()
        assert Q.dtype == torch.float32 and Z.dtype == torch.float32 and Z.is_contiguous() and Q.is_contiguous()
        with self._dev_ctx():
            check(self.lib.mu_dense_project_out_f32(n, B, _p(Q), _p(C), _p(Z), self._stream()))
        return Z

    def chol_rinv(self, G: torch.Tensor, w: int, flag: torch.Tensor) -> torch.Tensor:
        """R^-1 (f32, B x B, upper) of the leading w x w block of the f64 Gram G = R^T R, on the
        device; ``flag`` (int32[1]) is set when a pivot was not safely positive."""
        B = G.shape[0]
        assert G.dtype == torch.float64 and G.is_contiguous() and flag.dtype == torch.int32
        M = self.empty((B, B), torch.float32)
        with self._dev_ctx():
            check(self.lib.mu_chol_rinv_f64(B, int(w), _p(G), _p(M), _p(flag), self._stream()))
        return M

    def randn(self, rows: int, B: int, seed: int) -> torch.Tensor:
        out = self.empty((rows, B), torch.float32)
        with self._dev_ctx():
            check(self.lib.mu_randn_f32(rows * B, int(seed) & (2**64 - 1), _p(out), self._stream()))
        return out

    # -- MOFA+: tall-skinny products of a dense view with 16-column factor blocks ------------------
    skinny_mixed = True  # f64 products of a dense view stored in f32 (mu_skinny_*_f64_f32)

    def skinny_nn(self, Y: torch.Tensor, T16: torch.Tensor) -> torch.Tensor:
        """Y [n x D] (row slice of a row-major matrix) times T16 [D x 16] -> [n x 16].  Y may be stored in f32
        under an f64 block: the product is the f64 one."""
        n, D = Y.shape
        mixed = Y.dtype == torch.float32 and T16.dtype == torch.float64
        assert T16.shape == (D, 16) and (mixed or T16.dtype == Y.dtype) and T16.is_contiguous() and Y.stride(1) == 1
        out = self.empty((n, 16), T16.dtype)
        ld = Y.stride(0) if n > 1 else D
        with self._dev_ctx():
            if mixed:
                check(self.lib.mu_skinny_nn_f64_f32(n, D, ld, _p(Y), _p(T16), _p(out), self._stream()))
            else:
                check(self.lib.mu_skinny_nn(_dt(Y), n, D, ld, _p(Y), _p(T16), _p(out), self._stream()))
        return out

    def skinny_tn(self, Y: torch.Tensor, Z16: torch.Tensor) -> torch.Tensor:
        """Y^T [D x n] times Z16 [n x 16] -> [D x 16] (Y stored in f32 under an f64 block: the f64 product)."""
        n, D = Y.shape
        mixed = Y.dtype == torch.float32 and Z16.dtype == torch.float64
        assert Z16.shape == (n, 16) and (mixed or Z16.dtype == Y.dtype) and Z16.is_contiguous() and Y.stride(1) == 1
        out = self.empty((D, 16), Z16.dtype)
        wb = int(self.lib.mu_skinny_tn_worksize(_dt(Z16), n, D))
        work = self.empty((wb,), torch.uint8)
        ld = Y.stride(0) if n > 1 else D
        with self._dev_ctx():
            if mixed:
                check(self.lib.mu_skinny_tn_f64_f32(n, D, ld, _p(Y), _p(Z16), _p(out), _p(work), wb, self._stream()))
            else:
                check(self.lib.mu_skinny_tn(_dt(Y), n, D, ld, _p(Y), _p(Z16), _p(out), _p(work), wb, self._stream()))
        return out

    # -- MOFA+ coordinate updates (reference tools.py:585 -> mofapy2 node updates) -----------
    def mofa_update_w(self, B, tau, Gz, Z2, alpha, lth, l1mth, spikeslab, EW, EW2, gamma, EWh2, sig2):
        G, D, K = B.shape
        with self._dev_ctx():
            check(self.lib.mu_mofa_update_w(_dt(EW), D, K, G, _p(B), _p(tau), _p(Gz), _p(Z2),
                                            _p(alpha), _p(lth), _p(l1mth), int(bool(spikeslab)),
                                            _p(EW), _p(EW2), _p(gamma), _p(EWh2), _p(sig2),
                                            self._stream()))

    def mofa_update_z(self, A, pres, grp, Gw, dw2, alphaz, EZ, EZ2, sig2, corr=None):
        M, N, K = A.shape
        G = alphaz.shape[0]
        with self._dev_ctx():
            check(self.lib.mu_mofa_update_z(_dt(EZ), N, K, M, G, _p(A), _p(pres), _p(grp), _p(Gw),
                                            _p(dw2), _p(alphaz), _p(corr), _p(EZ), _p(EZ2), _p(sig2),
                                            self._stream()))

    def mofa_rowstats_work(self, K: int) -> torch.Tensor:
        return self.empty((int(self.lib.mu_mofa_rowstats_work_doubles(int(K))),), torch.float64)

    def mofa_rowstats(self, E, E2, r0, r1, work, wgt=None, aux=None, scale_out=False, out_pad=None, col0=0,
                      out_t=None, gram=None, s2=None, s1=None):
        """One pass over rows r0..r1 of the factor / weight block E (include/muon_amd.h)."""
        K = E.shape[1]
        ld = int(out_pad.shape[1]) if out_pad is not None else 0
        ld_t = int(out_t.shape[1]) if out_t is not None else 0
        with self._dev_ctx():
            check(self.lib.mu_mofa_rowstats(_dt(E), int(r0), int(r1), K, _p(E), _p(E2), _p(wgt), _p(aux),
                                            int(bool(scale_out)), _p(out_pad), ld, int(col0), _p(out_t), ld_t,
                                            _p(gram), _p(s2), _p(s1), _p(work), self._stream()))

    def umap_strengths(self, dist, idx, target: float, mean_all: float):
        """Membership strengths of a neighbour table [n, k] (include/muon_amd.h): a thread per row."""
        n, k = dist.shape
        assert dist.dtype == torch.float64 and idx.dtype == torch.int64 and dist.is_contiguous() and idx.is_contiguous()
        out = self.empty((n, k), torch.float64)
        with self._dev_ctx():
            check(self.lib.mu_umap_strengths_f64(int(n), int(k), _p(dist), _p(idx), float(target), float(mean_all),
                                                 _p(out), self._stream()))
        return out

    def wnn_bandwidth(self, X, g_indptr, g_indices, r_indptr, r_indices, n_bw: int, bbox: float):
        """csigma of muon.pp.neighbors, a wave per cell (include/muon_amd.h); returns (csigma, overflowed)."""
        n, p = X.shape
        out = self.empty((n,), torch.float64)
        over = self.zeros((1,), torch.int32)
        with self._dev_ctx():
            check(self.lib.mu_wnn_bandwidth_f64(int(n), int(p), _p(X), _p(g_indptr), _p(g_indices), _p(r_indptr),
                                                _p(r_indices), int(n_bw), float(bbox), _p(out), _p(over),
                                                self._stream()))
        return out, bool(int(over.item()))

    def knn_filter(self, Xq, Xc, sqq, sqc, thr, self_pos, c_lo, c_hi, buf_pos, buf_d, cnt):
        """Candidates of positions [c_lo, c_hi) that beat the queries' thresholds (include/muon_amd.h)."""
        with self._dev_ctx():
            check(self.lib.mu_knn_filter_f64(int(Xq.shape[0]), int(c_lo), int(c_hi), int(Xq.shape[1]), _p(Xq), _p(Xc),
                                             _p(sqq), _p(sqc), _p(thr), _p(self_pos), int(buf_pos.shape[1]),
                                             _p(buf_pos), _p(buf_d), _p(cnt), self._stream()))

    def knn_merge(self, cur_d, cur_p, buf_d, buf_pos, cnt):
        """list ++ buffer -> (the kc smallest distances ascending, their positions, the new thresholds)
        (include/muon_amd.h mu_knn_merge_f64)."""
        n, kc = cur_d.shape
        out_d, out_p = torch.empty_like(cur_d), torch.empty_like(cur_p)
        thr = self.empty((n,), torch.float64)
        with self._dev_ctx():
            check(self.lib.mu_knn_merge_f64(int(n), int(kc), int(buf_d.shape[1]), _p(cur_d), _p(cur_p), _p(buf_d),
                                            _p(buf_pos), _p(cnt), _p(out_d), _p(out_p), _p(thr), self._stream()))
        return out_d, out_p, thr

    def densify_rows(self, X: DeviceCSR, lo: int, hi: int) -> torch.Tensor:
        """Rows [lo, hi) of a device CSR as a dense chunk (include/muon_amd.h)."""
        out = self.empty((hi - lo, X.shape[1]), X.values.dtype)
        with self._dev_ctx():
            check(self.lib.mu_csr_densify_rows(_dt(X.values), int(lo), int(hi), int(X.shape[1]), _p(X.indptr),
                                               _p(X.indices), _p(X.values), _p(out), self._stream()))
        return out

    def mofa_jaakkola(self, zeta, a, b):
        """Bernoulli pseudo-data precision 2 lambda(xi), xi^2 = zeta^2 + a - b, written over ``a``."""
        assert zeta.is_contiguous() and a.is_contiguous() and b.is_contiguous() and a.shape == zeta.shape == b.shape
        with self._dev_ctx():
            check(self.lib.mu_mofa_jaakkola(_dt(zeta), int(zeta.numel()), _p(zeta), _p(a), _p(b), _p(a), self._stream()))
        return a

    def mofa_poisson_pseudo(self, zeta, Y, kappa, mode: int):
        """Poisson pseudo-data (mode 0) / likelihood terms (mode 1) of a dense chunk, in place of zeta."""
        n, D = zeta.shape
        assert zeta.is_contiguous() and Y.is_contiguous() and Y.shape == zeta.shape and Y.dtype == zeta.dtype
        with self._dev_ctx():
            check(self.lib.mu_mofa_poisson_pseudo(_dt(zeta), int(n), int(D), int(mode), _p(zeta), _p(Y), _p(kappa),
                                                  _p(zeta), self._stream()))
        return zeta

    mofa_poisson_lik_with_b = True  # (mode 3 of mofa_poisson_pass)

    def mofa_poisson_pass(self, mode: int, E_own, E_other, kappa, X: DeviceCSR, pads: Optional[dict] = None):
        """One pass of a poisson view without anything N x D (csrc/mofa_poisson.hip, include/muon_amd.h): mode 0 ->
        a = R <W> [N, K] (E_own = <Z>, E_other = <W>, X = the view), mode 1 -> b = R^T <Z> [D, K] (E_own = <W>, E_other =
        <Z>, X = the view's transpose), mode 2 -> per-sample likelihood terms [N], mode 3 -> mode 1 with the per-feature
        likelihood terms as column K: [D, K + 1]."""
        n_own, K = E_own.shape
        n_other = E_other.shape[0]
        assert E_other.shape[1] == K and E_own.dtype == E_other.dtype and 1 <= K <= 32
        assert X.shape == (n_own, n_other) and X.values.dtype == E_own.dtype
        # rows padded with zeros to ld columns (include/muon_amd.h): the padded width 4 / 8 / 12 / 16 / 32, and 16 for
        # 9 <= K <= 12 too - 64-byte rows let the stored-entry pass read a row in one cache-line look-up (r06)
        ld = 16 if 8 < K <= 16 else next(k for k in (4, 8, 32) if k >= K)

        def pad(E, slot):
            if K == ld and E.is_contiguous():
                return E
            # `pads` (a dict the caller owns, e.g. the engine of a fit): a padded buffer per (role, shape) whose columns
            # K .. ld - 1 are zeroed once - a pass copies the K columns in, one kernel instead of a fill and a copy.
            # Calls are ordered on one stream and a pass has consumed its operands when the next one overwrites them;
            # the buffers live as long as the caller (a captured iteration keeps pointing at them).
            if pads is None:
                P = torch.zeros((E.shape[0], ld), dtype=E.dtype, device=E.device)
            else:
                key = (slot, int(E.shape[0]), ld, K, E.dtype)
                P = pads.get(key)
                if P is None:
                    P = pads[key] = torch.zeros((E.shape[0], ld), dtype=E.dtype, device=E.device)
            P[:, :K].copy_(E)
            return P

        E_own, E_other = pad(E_own, "own"), pad(E_other, "other")
        blk = int(self.lib.mu_mofa_poisson_blocks_for(_dt(E_own), int(mode), K, n_own, n_other))
        nb = -(-n_other // blk)
        part = self.empty((nb, n_own) if mode == 2 else (nb, n_own, K + 1 if mode == 3 else K), E_own.dtype)
        with self._dev_ctx():
            check(self.lib.mu_mofa_poisson_dense_ld(_dt(E_own), int(mode), n_own, n_other, K, ld, blk, _p(E_own),
                                                    _p(E_other), _p(kappa), _p(part), self._stream()))
            out = part[0] if nb == 1 else part.sum(dim=0)  # (fixed order: deterministic)
            out = out.contiguous()
            check(self.lib.mu_mofa_poisson_sparse_ld(_dt(E_own), int(mode), n_own, K, ld, _p(X.indptr), _p(X.indices),
                                                     _p(X.values), _p(E_own), _p(E_other), _p(out), self._stream()))
        return out

    def mofa_softplus_sweep(self, E_own, E_other, pads: Optional[dict] = None):
        """out[own] = - sum_other ln(1 + e^zeta), zeta = <e_own> . <e_other>: the dense sweep of mofa_poisson_pass(2, ...)
        alone (the likelihood of a bernoulli view needs it without the poisson correction over the stored entries)"""
        n_own, K = E_own.shape
        n_other = E_other.shape[0]
        assert E_other.shape[1] == K and E_own.dtype == E_other.dtype and 1 <= K <= 32
        ld = 16 if 8 < K <= 16 else next(k for k in (4, 8, 32) if k >= K)

        def pad(E, slot):
            if K == ld and E.is_contiguous():
                return E
            key = (slot, int(E.shape[0]), ld, K, E.dtype)
            P = pads.get(key) if pads is not None else None
            if P is None:
                P = torch.zeros((E.shape[0], ld), dtype=E.dtype, device=E.device)
                if pads is not None:
                    pads[key] = P
            P[:, :K].copy_(E)
            return P

        E_own, E_other = pad(E_own, "own"), pad(E_other, "other")
        blk = int(self.lib.mu_mofa_poisson_blocks_for(_dt(E_own), 2, K, n_own, n_other))
        nb = -(-n_other // blk)
        part = self.empty((nb, n_own), E_own.dtype)
        with self._dev_ctx():
            check(self.lib.mu_mofa_poisson_dense_ld(_dt(E_own), 2, n_own, n_other, K, ld, blk, _p(E_own), _p(E_other),
                                                    None, _p(part), self._stream()))
            return part[0] if nb == 1 else part.sum(dim=0)

    def mofa_jaakkola_sweep(self, E_own, E2_own, E_other, E2_other):
        """The precision-weighted moment sums of a bernoulli view without anything N x D (csrc/mofa_bernoulli.hip,
        include/muon_amd.h): out[own] = sum_other Omega(own, other) <m m^T>_other as [n_own, K, K], Omega the Jaakkola
        precision of the pair.  E / E2: first / second moments [rows, K], K <= 16."""
        n_own, K = E_own.shape
        n_other = E_other.shape[0]
        assert 1 <= K <= 16 and E_other.shape[1] == K and E_own.dtype == E_other.dtype
        args = [t.contiguous() for t in (E_own, E2_own, E_other, E2_other)]
        ldm = int(self.lib.mu_mofa_jaakkola_cols(K))
        pc = K * (K + 1) // 2
        M = self.empty((n_other, ldm), E_own.dtype)
        blk = int(self.lib.mu_mofa_jaakkola_blocks(_dt(E_own), K, n_own, n_other))
        nb = -(-n_other // blk)
        part = self.empty((nb, n_own, pc), E_own.dtype)
        with self._dev_ctx():
            check(self.lib.mu_mofa_pack_moments(_dt(E_own), n_other, K, ldm, _p(args[2]), _p(args[3]), _p(M), self._stream()))
            check(self.lib.mu_mofa_jaakkola_sweep(_dt(E_own), n_own, n_other, K, blk, _p(args[0]), _p(args[1]), _p(args[2]),
                                                  _p(args[3]), _p(M), ldm, _p(part), self._stream()))
            packed = part[0] if nb == 1 else part.sum(dim=0)  # (fixed order: deterministic)
            idx = self._tri_index.get((K, packed.device))
            if idx is None:
                pos = {}
                for k in range(K):
                    for l in range(k, K):
                        pos[(k, l)] = len(pos)
                idx = torch.tensor([pos[(min(k, l), max(k, l))] for k in range(K) for l in range(K)], dtype=torch.int64,
                                   device=packed.device)
                self._tri_index[(K, packed.device)] = idx
            return packed.index_select(1, idx).view(n_own, K, K)

    def mofa_gs_update(self, Tm, b, prior, lth, l1mth, spikeslab, E, E2, gamma, Eh2, sig2):
        """Gauss-Seidel sweep over the factors of every row with row-wise K x K statistics (include/muon_amd.h);
        prior / lth / l1mth: f64 [K]."""
        n, K = E.shape
        assert Tm.is_contiguous() and b.is_contiguous() and E.is_contiguous() and E2.is_contiguous() and sig2.is_contiguous()
        with self._dev_ctx():
            check(self.lib.mu_mofa_gs_update(_dt(E), int(n), int(K), _p(Tm), _p(b), _p(prior), _p(lth), _p(l1mth),
                                             int(bool(spikeslab)), _p(E), _p(E2), _p(gamma), _p(Eh2), _p(sig2),
                                             self._stream()))

    def mofa_elbo_work(self, K: int) -> torch.Tensor:
        return self.empty((int(self.lib.mu_mofa_elbo_work_doubles(int(K))),), torch.float64)

    def mofa_tau_elbo(self, yy, Ngm, EW, EW2, B, Gz, Z2, a0, b0, tau, ltau, elbo, work):
        """tau / <ln tau> of one view from the sufficient statistics; adds the likelihood and tau-node
        terms to the f64 device scalar ``elbo`` (include/muon_amd.h)."""
        G, D, K = B.shape
        with self._dev_ctx():
            check(self.lib.mu_mofa_tau_elbo(_dt(EW), D, K, G, _p(yy), _p(Ngm), _p(EW), _p(EW2), _p(B), _p(Gz),
                                            _p(Z2), float(a0), float(b0), _p(tau), _p(ltau), _p(elbo),
                                            _p(work), self._stream()))

    def mofa_stats_resid(self, yy, EW, EW2, B, Q, S):
        """S[d] = yy[d] - 2 <w_d> . B[d] + sum Q[d] * <w w^T>_d in f64 (include/muon_amd.h): yy, S f64 [D]; EW, EW2, B [D, K];
        Q [D or 1, K^2]"""
        D, K = EW.shape
        assert yy.dtype == torch.float64 and S.dtype == torch.float64 and yy.is_contiguous() and S.is_contiguous()
        assert EW.is_contiguous() and EW2.is_contiguous() and B.is_contiguous() and Q.is_contiguous()
        assert B.shape == (D, K) and Q.shape[1] == K * K and Q.shape[0] in (1, D) and B.dtype == EW.dtype == Q.dtype
        with self._dev_ctx():
            check(self.lib.mu_mofa_stats_resid(_dt(EW), int(D), int(K), int(Q.shape[0]), _p(yy), _p(EW), _p(EW2), _p(B),
                                               _p(Q), _p(S), self._stream()))

    def mofa_tau_finish(self, S, Ngd, a0, b0, tau, ltau, elbo, work):
        """tau / <ln tau> from the expected squared residuals and counts of every (group, feature) (f64, same shape as
        tau); adds the likelihood and tau-node terms to the f64 device scalar ``elbo`` (include/muon_amd.h)."""
        assert S.dtype == torch.float64 and Ngd.dtype == torch.float64 and S.is_contiguous() and Ngd.is_contiguous()
        assert tau.is_contiguous() and ltau.is_contiguous() and tau.numel() == S.numel() == Ngd.numel() == ltau.numel()
        with self._dev_ctx():
            check(self.lib.mu_mofa_tau_finish(_dt(tau), int(S.numel()), _p(S), _p(Ngd), float(a0), float(b0), _p(tau),
                                              _p(ltau), _p(elbo), _p(work), self._stream()))

    def mofa_w_elbo(self, EWh2, gamma, sig2, ard, spikeslab, a_alpha, a0, b0, th_a0, th_b0, alpha, lalpha,
                    lth, l1mth, elbo, work):
        D, K = EWh2.shape
        with self._dev_ctx():
            check(self.lib.mu_mofa_w_elbo(_dt(EWh2), D, K, int(bool(ard)), int(bool(spikeslab)), _p(EWh2),
                                          _p(gamma), _p(sig2), float(a_alpha), float(a0), float(b0),
                                          float(th_a0), float(th_b0), _p(alpha), _p(lalpha), _p(lth),
                                          _p(l1mth), _p(elbo), _p(work), self._stream()))

    def mofa_z_sums(self, EZ2, sig2, n0, n1, out, work):
        K = EZ2.shape[1]
        with self._dev_ctx():
            check(self.lib.mu_mofa_z_sums(_dt(EZ2), int(n0), int(n1), K, _p(EZ2), _p(sig2), _p(out), _p(work),
                                          self._stream()))

    def mofa_z_elbo(self, zs, Ng, ard, a0, b0, alpha_z, lalpha_z, elbo):
        G, _two, K = zs.shape
        with self._dev_ctx():
            check(self.lib.mu_mofa_z_elbo(_dt(alpha_z), K, G, int(bool(ard)), _p(zs), _p(Ng), float(a0),
                                          float(b0), _p(alpha_z), _p(lalpha_z), _p(elbo), self._stream()))

    # -- synthetic data (bench / tests) ---------------------------------------------
    def synth_counts(self, row0: int, n_rows: int, n_cols: int, n_topics: int = 50,
                     density: float = 0.03, seed: int = 0) -> DeviceCSR:
        row_nnz = self.empty((n_rows,), torch.int64)
        indptr = self.empty((n_rows + 1,), torch.int64)
        with self._dev_ctx():
            st = self._stream()
            check(self.lib.mu_synth_row_nnz(row0, n_rows, n_cols, n_topics, density, seed,
                                            _p(row_nnz), st))
            check(self.lib.mu_exclusive_scan_i64(n_rows, _p(row_nnz), _p(indptr), st))
            nnz = int(indptr[-1].item())
            indices = self.empty((nnz,), torch.int32)
            values = self.empty((nnz,), torch.float32)
            check(self.lib.mu_synth_fill(row0, n_rows, n_cols, n_topics, density, seed, _p(indptr),
                                         _p(indices), _p(values), st))
        # (the generator stands where ingest stands: the slab pointers of its index arrays are made here, once)
        return self.with_slab_ptr(DeviceCSR(indptr, indices, values, (n_rows, n_cols)))


class _Fetch:
    def __init__(self, bufs, event, pool=None):
        self.bufs, self.event, self.pool = bufs, event, pool

    def wait(self):
        self.event.synchronize()
        out = [b.numpy().copy() for b in self.bufs]
        if self.pool is not None:  # the staging buffers go back to the pool (an abandoned handle keeps its own)
            for b in self.bufs:
                self.pool.setdefault((tuple(b.shape), b.dtype), []).append(b)
            self.bufs, self.pool = [], None
        return out

    def release(self):
        """Give the staging buffers back without reading them (an abandoned fetch).  Safe without waiting: the next
        copy into a buffer is queued on the same stream behind this one, and its reader waits for its own event."""
        if self.pool is not None:
            for b in self.bufs:
                self.pool.setdefault((tuple(b.shape), b.dtype), []).append(b)
            self.bufs, self.pool = [], None


_default_backend = None


def get_backend():
    """The process-wide HipBackend on the current device (raises without a GPU)."""
    global _default_backend
    if _default_backend is None or _default_backend.device.index != torch.cuda.current_device():
        _default_backend = HipBackend()
    return _default_backend
