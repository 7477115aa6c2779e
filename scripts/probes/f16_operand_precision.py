"""CPU experiment (r04): what 16-bit operands in BOTH factors of the LSI products do to the top-k subspace.

The matrix-core SpMM (csrc/spmm_mfma.hip) multiplies f16 stored values with f16 rows of the dense operand
and sums in f32.  This script runs the product's arithmetic on the CPU inside the real block Lanczos host
code (tests/cpu_backend.py) and measures the largest principal angle against f64 ARPACK on the exact matrix.
  modes: exact | q16 (dense operand f16) | x16 (values f16 too) | x16hl (values hi + lo f16, dense f16)
"""
import sys, os, time
import numpy as np, scipy.sparse as sp, scipy.linalg, scipy.sparse.linalg as spla, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tests.cpu_backend import CpuTestBackend
from tests.synth import planted_topics_csr, unstructured_csr
from oracle.tfidf_oracle import tfidf as tfidf_csr
from muon_amd._atac import tools
from muon_amd._backend import DeviceCSR


def q_round(Q):
    """f16 with one power-of-two scale per column (exact scaling)"""
    q = Q.numpy().astype(np.float64)
    mx = np.abs(q).max(axis=0)
    mx[mx == 0] = 1
    sc = 2.0 ** (np.floor(np.log2(mx)) - 13)  # largest entry in [2^13, 2^14)
    return (q / sc).astype(np.float16).astype(np.float64), sc


class B16(CpuTestBackend):
    def __init__(self, mode, n_rows=None):
        self.mode = mode
        self.cache = {}
        self.n_rows = n_rows

    def spmm(self, X, Q, out=None):
        key = id(X)
        if key not in self.cache:
            m = self._sp(X).astype(np.float64)
            hi = m.copy(); hi.data = m.data.astype(np.float16).astype(np.float64)
            lo = m.copy(); lo.data = (m.data - hi.data).astype(np.float16).astype(np.float64)
            self.cache[key] = (m, hi, lo)
        m, hi, lo = self.cache[key]
        if self.mode == "basis":
            # what the matrix-core product does (r04): X Q rounds the block IN PLACE - the rounded block is the
            # basis block from then on - and multiplies hi + lo values (22 bits) with it; X^T Y stays exact f32
            if X.shape[0] == self.n_rows:
                qh, sc = q_round(Q)
                Q.copy_(torch.from_numpy((qh * sc).astype(np.float32)))
                y = (hi + lo) @ (qh * sc)
            else:
                y = m @ Q.numpy().astype(np.float64)
        elif self.mode == "exact":
            y = (m @ Q.numpy().astype(np.float64))
        else:
            qh, sc = q_round(Q)
            if self.mode == "q16":
                y = (m @ qh) * sc
            elif self.mode == "x16":
                y = (hi @ qh) * sc
            elif self.mode == "x16hl":
                y = (hi @ qh + lo @ qh) * sc
        y = torch.from_numpy(y.astype(np.float32))
        if out is not None:
            out.copy_(y); return out
        return y


def run(name, X, k=50):
    T = tfidf_csr(X)
    T = sp.csr_matrix(T); T.sort_indices()
    t0 = time.time()
    u, s, vt = spla.svds(T.astype(np.float64), k=k, tol=1e-12)
    Vref = vt.T
    print(f"{name}: shape {T.shape} nnz {T.nnz} arpack f64 {time.time()-t0:.1f}s  sigma_k/sigma_1 {s.min()/s.max():.3f}", flush=True)
    for mode in MODES:
        b = B16(mode, T.shape[0])
        Xd = DeviceCSR(torch.from_numpy(T.indptr.astype(np.int64)), torch.from_numpy(T.indices.astype(np.int32)),
                       torch.from_numpy(T.data.astype(np.float32)), T.shape)
        U, stdev, V, info = tools.lsi_device(b, Xd, n_comps=k, return_info=True, pack=False)
        ang = scipy.linalg.subspace_angles(Vref, V.numpy().astype(np.float64)).max()
        sv = np.abs(info["svalues"] - np.sort(s)[::-1]).max() / s.max()
        print(f"  {mode:6s} angle {ang:.2e}  sv rel {sv:.1e}  products {info['spmm']} converged {info['converged']} bound {info['angle_bound']:.1e}", flush=True)


MODES = ("exact", "basis", "q16")

if __name__ == "__main__":
    which = sys.argv[1:] or ["c2"]
    if "k100" in which:
        # tests/test_gpu_lsi.py::test_more_components_than_the_block_width
        run("5000 x 6000, k = 100", planted_topics_csr(5000, 6000, n_topics=100, density=0.04, seed=8), k=100)
    if "small" in which:
        run("small 3000x4000", planted_topics_csr(3000, 4000, n_topics=20, density=0.03, seed=3), k=15)
    if "c2" in which:
        run("c2 10000x30000", planted_topics_csr(10000, 30000, n_topics=50, density=0.03, seed=0))
    if "uns" in which:
        run("unstructured 8000x20000", unstructured_csr(8000, 20000, seed=1))
    if "wide" in which:
        run("12000x200000", planted_topics_csr(12000, 200000, n_topics=50, density=0.03, seed=0))
