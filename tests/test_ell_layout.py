"""The sliced-ELL layout of csrc/spmm_ell.hip (muon_amd._backend.ell16_layout, tensor operations) against its
definition in include/muon_amd.h: a numpy walk over the windows in the order the kernel consumes them must return
every stored entry exactly once, in stored order per row, and the product it implies must be X Q."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from muon_amd._backend import DeviceCSR, ell16_layout
from tests.synth import planted_topics_csr


def _walk(E):
    slab, rb = E.slab_cols, 65536 // E.slab_cols
    """(row, col, value) of every non-padding slot, per group in consumption order"""
    hdr, base, ent, perm = E.hdr.numpy(), E.wave_base.numpy(), E.ent.numpy(), E.perm.numpy()
    n_groups, S = hdr.shape
    out = []
    for w in range(n_groups):
        p = int(base[w])
        for s in range(S):
            for _ in range(int(hdr[w, s])):
                vals = ent[p, :256].view(np.float32)
                offs = ent[p, 256:].view(np.uint16)
                for j in range(4):
                    for r in range(16):
                        v, off = vals[4 * r + j], int(offs[4 * r + j])
                        row = perm[w * 16 + r]
                        if v != 0 or off != 0:
                            assert row >= 0 and off % rb == 0 and off < 65536
                            out.append((int(row), s * slab + off // rb, float(v)))
                p += 1
    return out


@pytest.mark.parametrize("slab", [1024, 512])
@pytest.mark.parametrize("shape", [(70, 2500), (100, 1024), (37, 3000), (16, 90)])
def test_layout_holds_every_entry_once_in_stored_order(shape, slab):
    m = planted_topics_csr(shape[0], shape[1], n_topics=4, density=0.06, seed=shape[0])
    m.data = (m.data + 0.5).astype(np.float32)  # no zero values: padding is recognisable
    keep = np.ones(shape[0]); keep[3] = 0  # an empty row
    m = sp.csr_matrix(sp.diags(keep) @ m).astype(np.float32)
    m.eliminate_zeros()
    m.sort_indices()
    X = DeviceCSR(torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int32)),
                  torch.from_numpy(m.data.astype(np.float32)), m.shape)
    E = ell16_layout(X, 15, slab)
    got = _walk(E)
    coo = m.tocoo()
    want = sorted(zip(coo.row.tolist(), coo.col.tolist(), coo.data.tolist()))
    assert sorted(got) == want
    # per row the entries come in column order (the kernel adds them in that order: bit-reproducible sums)
    last = {}
    for r, c, _ in got:
        assert last.get(r, -1) < c
        last[r] = c
    # slots / entries: the padding the kernel streams
    assert E.slots >= m.nnz and E.ent.shape == (E.slots // 64 + 8, 384)
