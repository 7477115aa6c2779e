#!/usr/bin/env python
"""How often the fourth-generation fill's 32-entry windows overflow on the bench matrix, per tile width: rows with more
than 32 / 64 entries in a tile, and waves (32 consecutive rows) with more than 2 / 4 / 6 such rows in one tile (what one,
two or three overflow slots could take without retrying the tile)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from muon_amd._backend import HipBackend

be = HipBackend(0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
X = be.synth_counts(0, cells, 200000, 50, 0.03, 0)
n, d = X.shape
for C in (416, 480, 544, 608, 672, 736):
    sp = be.slab_ptr_width(X, C).view(n, -1)
    cnt = sp[:, 1:] - sp[:, :-1]
    T = cnt.shape[1]
    nw = n // 32
    over = (cnt[: nw * 32] > 32).view(nw, 32, T).sum(dim=1)  # overflow rows per (wave, tile)
    big = (cnt[: nw * 32] > 64).view(nw, 32, T).any(dim=1)
    blocks = nw // 16
    def tiles_hit(mask):  # fraction of (block, tile) with any wave hit
        return float(mask[: blocks * 16].view(blocks, 16, T).any(dim=1).float().mean())
    print(f"C = {C} (mean {float(cnt.float().mean()):.1f} entries per row and tile): rows over 32: {float((cnt > 32).float().mean()):.4f}, "
          f"over 64: {float((cnt > 64).float().mean()):.5f}; tiles with a wave of > 2 overflow rows {tiles_hit(over > 2):.3f}, > 4: {tiles_hit(over > 4):.3f}, "
          f"> 6: {tiles_hit(over > 6):.3f}, > 8: {tiles_hit(over > 8):.3f}; tiles with a row over 64: {tiles_hit(big):.3f}; "
          f"tiles with any overflow {tiles_hit(over > 0):.3f}", flush=True)
