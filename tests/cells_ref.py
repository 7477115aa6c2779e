"""TEST INFRASTRUCTURE: numpy statement of the cell format of csrc/spmm_mfma.hip (include/muon_amd.h,
"matrix-core SpMM operand") - an encoder, a decoder and the arithmetic of the product on it.  The GPU tests
compare mu_cells_cut with `decode` of both sides (a cell is an unordered set of slots) and mu_spmm_cells_f32
with `product`.  Slow loops: small matrices only."""
import numpy as np
import scipy.sparse as sp

STEP_BYTES = 224  # hi[32] f16 | lo[32] f16 | off[32] u16 in gather order | mask[4][8] u8
BAND_ROWS, TILE_ROWS, RING = 32, 8, 2


def _idx(k):
    """u16 indices of (hi, lo, offset) of k-slot k inside a record, and its (mask dword base, bit)"""
    kb, ii = k >> 3, k & 7
    return k, 32 + k, 64 + kb * 8 + 2 * (ii & 3) + (ii >> 2), 48 + 2 * kb, ii


def geometry(nset):
    return dict(slab_rows=512 if nset == 1 else 256, stride=160 if nset == 1 else 288, step_bytes=STEP_BYTES,
                band_rows=BAND_ROWS, ring=RING)


def value_scale(values):
    """power of two that puts the largest |value| into [2^13, 2^14)"""
    vmax = float(np.max(np.abs(values))) if len(values) else 0.0
    e = np.floor(np.log2(vmax)) - 13.0 if vmax > 0 else 0.0
    return float(2.0 ** e)


def split_f16(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def band_bounds(indptr, n_rows, n_slabs):
    n_bands = -(-n_rows // BAND_ROWS)
    edge = np.array([indptr[min(BAND_ROWS * b, n_rows)] for b in range(n_bands + 1)], dtype=np.int64)
    ub = (edge[1:] - edge[:-1]) // 32 + 5 * n_slabs  # (+ an all-zero step where a slab's count is odd)
    base = np.zeros(n_bands + 1, dtype=np.int64)
    np.cumsum(ub, out=base[1:])
    return base


def encode(m, nset=1, vscale=None):
    """scipy CSR (f32, canonical) -> (hdr int32 [bands, slabs], band_base int64, cells uint8, vscale)"""
    m = sp.csr_matrix(m)
    n, d = m.shape
    g = geometry(nset)
    sr, gran = g["slab_rows"], g["stride"] // 8
    n_bands, n_slabs = -(-n // BAND_ROWS), -(-d // sr)
    if vscale is None:
        vscale = value_scale(m.data)
    base = band_bounds(m.indptr.astype(np.int64), n, n_slabs)
    cells = np.zeros((int(base[-1]) + RING + 1) * STEP_BYTES, dtype=np.uint8)
    hdr = np.zeros((n_bands, n_slabs), dtype=np.int32)
    for b in range(n_bands):
        out = int(base[b])
        for s in range(n_slabs):
            c0, c1 = s * sr, min((s + 1) * sr, d)
            steps = 0
            for t in range(4):
                slots = []
                for r in range(TILE_ROWS):
                    row = b * BAND_ROWS + t * TILE_ROWS + r
                    if row >= n:
                        continue
                    lo_, hi_ = m.indptr[row], m.indptr[row + 1]
                    cols = m.indices[lo_:hi_]
                    sel = (cols >= c0) & (cols < c1)
                    for c, v in zip(cols[sel], m.data[lo_:hi_][sel]):
                        slots.append((r, int(c) - c0, np.float32(v)))
                for q in range(0, len(slots), 32):
                    rec = np.zeros(STEP_BYTES, dtype=np.uint8)
                    h16 = rec.view(np.uint16)
                    d32 = rec.view(np.uint32)
                    for k, (r, c, v) in enumerate(slots[q:q + 32]):
                        h, l = split_f16(np.array([v / np.float32(vscale)], dtype=np.float32))
                        ih, il, io, im, ii = _idx(k)
                        h16[ih], h16[il] = h.view(np.uint16)[0], l.view(np.uint16)[0]
                        h16[io] = (c * gran) | ((t << 14) if (ii >> 2) == 0 else 0)
                        d32[im + (r >> 2)] |= np.uint32(1 << (8 * (r & 3) + ii))
                    cells[(out) * STEP_BYTES:(out + 1) * STEP_BYTES] = rec
                    out += 1
                    steps += 1
            if steps & 1:  # the product consumes steps in pairs: an all-zero step completes an odd slab
                out += 1
                steps += 1
            hdr[b, s] = steps
        assert out <= base[b + 1]
    return hdr, base, cells, vscale


def decode(hdr, band_base, cells, shape, nset=1):
    """-> sorted array of (row, col, hi_bits, lo_bits) of every real slot (mask bit set), plus the number of steps"""
    n, d = shape
    g = geometry(nset)
    sr, gran = g["slab_rows"], g["stride"] // 8
    cells = np.asarray(cells, dtype=np.uint8)
    out = []
    n_steps = 0
    for b in range(hdr.shape[0]):
        p = int(band_base[b])
        for s in range(hdr.shape[1]):
            for _ in range(int(hdr[b, s])):
                rec = cells[p * STEP_BYTES:(p + 1) * STEP_BYTES]
                h16 = rec.view(np.uint16)
                d32 = rec.view(np.uint32)
                t = int(h16[_idx(0)[2]]) >> 14
                for k in range(32):
                    ih, il, io, im, ii = _idx(k)
                    rows = [r for r in range(8) if int(d32[im + (r >> 2)]) & (1 << (8 * (r & 3) + ii))]
                    assert len(rows) <= 1, "a slot belongs to at most one tile row"
                    if not rows:
                        continue
                    o = int(h16[io])
                    assert (o >> 14) == (t if (ii >> 2) == 0 else 0)
                    c = (o & 0x3fff) // gran
                    assert (o & 0x3fff) % gran == 0
                    out.append((b * BAND_ROWS + t * TILE_ROWS + rows[0], s * sr + c, int(h16[ih]), int(h16[il])))
                p += 1
                n_steps += 1
        assert p <= band_base[b + 1]
    arr = np.array(sorted(out), dtype=np.int64).reshape(-1, 4)
    return arr, n_steps


def round_block(Q):
    """what mu_dense_to_f16 keeps of an f32 block: (rounded block f32, hi f16 scaled, lo f16 scaled, scale[64])"""
    Q = np.asarray(Q, dtype=np.float32)
    mx = np.abs(Q).max(axis=0)
    e = np.zeros(Q.shape[1])
    nz = mx > 0
    e[nz] = np.frexp(mx[nz])[1] - 14
    scale = (2.0 ** e).astype(np.float32)
    xs = Q / scale
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32) * scale, hi, lo, scale


def product(m, Q, nset=1, vscale=None):
    """f64 statement of mu_spmm_cells_f32: (hi + lo of the scaled values) x (f16 block [+ lo]) with the scales
    put back; also returns the rounded block (what nset = 1 leaves in Q)."""
    m = sp.csr_matrix(m)
    if vscale is None:
        vscale = value_scale(m.data)
    h, l = split_f16((m.data / np.float32(vscale)).astype(np.float32))
    mv = m.copy().astype(np.float64)
    mv.data = (h.astype(np.float64) + l.astype(np.float64)) * vscale
    Qr, hi, lo, scale = round_block(Q)
    B = hi.astype(np.float64) * scale
    if nset == 2:
        B = B + lo.astype(np.float64) * scale
    return mv @ B, Qr
