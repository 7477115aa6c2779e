"""Collectives for the row(cell)-sharded path: one process per GPU, RCCL over xGMI.

The reference has no distributed code; sharding cells across ranks is this package's
design.  Only three reductions exist on the path:
  * tfidf : all-reduce of the per-peak count sums (d values)
  * lsi   : all-reduce of Z = X^T Y (d x B) per iteration, and of the small B x B Gram
  * mofa  : all-reduce of the D x K sufficient statistics per iteration
``torch.distributed`` with backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import torch


class LocalComm:
    """Single process: every reduction is the identity."""

    world_size = 1
    rank = 0

    def all_reduce_sum(self, *tensors):
        return tensors[0] if len(tensors) == 1 else tensors

    def all_gather_rows(self, t):
        return t

    def all_reduce_sum_big(self, t):
        return t

    def reduce_scatter_rows(self, t):
        return 0, int(t.shape[0])

    def all_gather_rows_into(self, t, rows):
        return t

    def sum_scalar(self, x):
        return x

    def all_reduce_max(self, t):
        return t

    def agree(self, *flags):
        return bool(flags[0]) if len(flags) == 1 else tuple(bool(f) for f in flags)


class TorchDistComm:
    def __init__(self, group=None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_reduce_sum(self, *tensors):
        """In-place sum over the ranks.  Several tensors travel as ONE message per dtype: a Krylov
        step of the LSI reduces up to four small Grams, a MOFA iteration four statistics, and on
        xGMI every collective pays its launch and ring latency whatever its size."""
        groups = {}
        for t in tensors:
            groups.setdefault((t.dtype, t.device), []).append(t)
        for ts in groups.values():
            if len(ts) == 1:
                t = ts[0]
                if t.is_contiguous():
                    self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
                else:
                    c = t.contiguous()
                    self._dist.all_reduce(c, op=self._dist.ReduceOp.SUM, group=self.group)
                    t.copy_(c)
                continue
            flat = torch.cat([t.reshape(-1) for t in ts])
            self._dist.all_reduce(flat, op=self._dist.ReduceOp.SUM, group=self.group)
            o = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[o:o + n].view(t.shape))
                o += n
        return tensors[0] if len(tensors) == 1 else tensors

    def all_reduce_max(self, t):
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
        return t

    def all_reduce_sum_big(self, t, mode=None):
        """In-place sum of ONE large contiguous tensor - Z = X^T Y of an LSI expansion, 51 MB at 200 000 peaks - either
        as a plain all-reduce (default) or, with MUON_AMD_Z_COLLECTIVE=rsag (or mode="rsag"; an "rsqr" caller that does
        not work on the slice in between gets the same), as an explicit
        reduce-scatter + all-gather over row chunks: every rank sums 1 / W of the rows, then the chunks are gathered.
        On xGMI's point-to-point links the two halves are what a ring all-reduce does anyway; having them as separate
        calls is what lets the first 8-GPU lease A/B them (and later overlap the reduce-scatter of one column chunk
        with the product of the next).  Same sums: each element is added over the ranks in one place, in rank order
        for the gloo path.  (r05: built because VERDICT r04 asked for the variant; unmeasured on more than one GPU.)"""
        import os

        mode = mode or os.environ.get("MUON_AMD_Z_COLLECTIVE", "allreduce")
        W = self.world_size
        if mode not in ("rsag", "rsqr") or W == 1 or not t.is_contiguous() or t.numel() < W:
            return self.all_reduce_sum(t)
        t2 = t.view(t.shape[0], -1) if t.dim() >= 2 else t.view(-1, 1)
        rows = self.reduce_scatter_rows(t2)
        self.all_gather_rows_into(t2, rows)
        return t

    def _row_chunks(self, t):
        W = self.world_size
        d = int(t.shape[0])
        width = int(t.numel() // max(d, 1)) if d else 0
        per = -(-d // W) if d else 0
        r0 = min(self.rank * per, d)
        return d, width, per, r0, min(r0 + per, d)

    def reduce_scatter_rows(self, t):
        """SURVEY 8e's first half for a contiguous [rows, width] block: every rank ends up with the SUM over the ranks of
        its own 1 / W of the rows - in place, t[r0:r1] - and returns (r0, r1); the other rows keep this rank's partial
        values (the caller works on its slice and calls all_gather_rows_into).  Chunks are ceil(rows / W) rows; the last
        ranks' chunks may be short or empty."""
        dist = self._dist
        assert t.is_contiguous()
        d, width, per, r0, r1 = self._row_chunks(t)
        W = self.world_size
        if W == 1 or d == 0:
            return 0, d
        flat = t.view(-1)
        n, chunk = d * width, per * width
        if chunk * W != n:
            buf = torch.zeros((chunk * W,), dtype=t.dtype, device=t.device)
            buf[:n] = flat
        else:
            buf = flat
        mine = torch.empty((chunk,), dtype=t.dtype, device=t.device)
        if dist.get_backend(self.group) == "nccl":
            dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM, group=self.group)
        else:
            # gloo has no reduce-scatter: W reductions, one per owner (the CPU tests exercise the chunking and the gather)
            for r in range(W):
                piece = buf[r * chunk:(r + 1) * chunk].clone()
                dst = dist.get_global_rank(self.group, r) if self.group is not None else r
                dist.reduce(piece, dst=dst, op=dist.ReduceOp.SUM, group=self.group)
                if r == self.rank:
                    mine.copy_(piece)
        if r1 > r0:
            t[r0:r1].reshape(-1).copy_(mine[:(r1 - r0) * width])
        return r0, r1

    def all_gather_rows_into(self, t, rows):
        """The second half: every rank's slice t[r0:r1] into every rank's t (in place)."""
        dist = self._dist
        d, width, per, r0, r1 = self._row_chunks(t)
        W = self.world_size
        if W == 1 or d == 0:
            return t
        assert (r0, r1) == tuple(rows)
        flat = t.view(-1)
        n, chunk = d * width, per * width
        mine = torch.zeros((chunk,), dtype=t.dtype, device=t.device)
        if r1 > r0:
            mine[:(r1 - r0) * width] = t[r0:r1].reshape(-1)
        if dist.get_backend(self.group) == "nccl":
            buf = torch.empty((chunk * W,), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(buf, mine, group=self.group)
        else:
            parts = [torch.empty_like(mine) for _ in range(W)]
            dist.all_gather(parts, mine, group=self.group)
            buf = torch.cat(parts)
        flat.copy_(buf[:n])
        return t

    def all_gather_rows(self, t):
        """Concatenate row shards of possibly different length along dim 0."""
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world_size)]
        self._dist.all_gather(sizes, n, group=self.group)
        sizes = [int(s.item()) for s in sizes]
        mx = max(sizes)
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(self.world_size)]
        self._dist.all_gather(bufs, pad, group=self.group)
        return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)

    def sum_scalar(self, x):
        t = torch.tensor([float(x)], dtype=torch.float64)
        if self._dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        self._dist.all_reduce(t, group=self.group)
        return float(t.item())


    def agree(self, *flags):
        """Rank 0's decisions, for every rank (one broadcast for all the flags of a step): stop /
        restart / speculation / convergence tests are taken from host LAPACK results that need not be
        bit-identical across ranks (different BLAS builds or CPUs), and a rank that leaves a loop - or
        queues a product with its collective - alone strands the others in their next collective."""
        t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32)
        if self._dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        self._dist.broadcast(t, src=self._dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                             group=self.group)
        got = [bool(int(v)) for v in t.tolist()]
        return got[0] if len(got) == 1 else tuple(got)


def default_comm(comm=None):
    """``comm=None`` always means single process (the reference's semantics): a caller that
    holds a row shard per rank passes ``TorchDistComm()`` explicitly."""
    return LocalComm() if comm is None else comm
