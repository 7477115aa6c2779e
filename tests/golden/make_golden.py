#!/usr/bin/env python
"""Generate golden input/output vectors by EXECUTING the reference's own code.

muon itself cannot be imported in the build image (anndata / mudata / scanpy /
h5py are absent), but the two hot-path functions only touch those packages for
type dispatch and ``view_to_actual``.  This script installs tiny stub modules
for them, loads the reference source files *where they lie* under
/root/reference (nothing is copied) and runs

  * muon._atac.preproc.tfidf   (/root/reference/muon/_atac/preproc.py:16-129)
  * muon._atac.tools.lsi       (/root/reference/muon/_atac/tools.py:29-71)

on seeded inputs, including the exact inputs of the reference's own tests
(/root/reference/tests/test_atac_preproc.py:11-14, 47-52, 57-58).  Outputs are
written as small ``.npz`` fixtures next to this file.  /root/reference does not
exist on the GPU box: tests read only the fixtures.

Run (in the build container):  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("MUON_REFERENCE", "/root/reference")

from muon_amd._containers import AnnData, MuData, view_to_actual  # noqa: E402
from tests.synth import planted_topics_csr  # noqa: E402


def _install_stubs():
    anndata = types.ModuleType("anndata")
    anndata.AnnData = AnnData
    mudata = types.ModuleType("mudata")
    mudata.MuData = MuData
    scanpy = types.ModuleType("scanpy")
    sc_utils = types.ModuleType("scanpy._utils")
    sc_utils.view_to_actual = view_to_actual
    sc_logging = types.ModuleType("scanpy.logging")
    sc_logging.info = lambda *a, **k: None
    scanpy._utils = sc_utils
    scanpy.logging = sc_logging
    sys.modules.update(
        {
            "anndata": anndata,
            "mudata": mudata,
            "scanpy": scanpy,
            "scanpy._utils": sc_utils,
            "scanpy.logging": sc_logging,
        }
    )
    # fake package skeleton so relative imports in the reference files resolve
    muon = types.ModuleType("muon")
    muon.__path__ = [os.path.join(REF, "muon")]
    muon.MuData = MuData
    atac = types.ModuleType("muon._atac")
    atac.__path__ = [os.path.join(REF, "muon", "_atac")]
    rna = types.ModuleType("muon._rna")
    rna.__path__ = [os.path.join(REF, "muon", "_rna")]
    rna_utils = types.ModuleType("muon._rna.utils")
    rna_utils.get_gene_annotation_from_rna = None
    sys.modules.update(
        {"muon": muon, "muon._atac": atac, "muon._rna": rna, "muon._rna.utils": rna_utils}
    )


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    _install_stubs()
    _load("muon._atac.utils", "muon/_atac/utils.py")
    preproc = _load("muon._atac.preproc", "muon/_atac/preproc.py")
    tools = _load("muon._atac.tools", "muon/_atac/tools.py")
    return preproc, tools


def csr_parts(m, prefix):
    m = m.tocsr()
    return {
        prefix + "_data": m.data,
        prefix + "_indices": m.indices,
        prefix + "_indptr": m.indptr,
        prefix + "_shape": np.asarray(m.shape),
    }


def main():
    preproc, tools = load_reference()
    out = {}

    # ---- reference test inputs: dense 4x5 (test_atac_preproc.py:11-14) ----------
    np.random.seed(2020)
    x = np.abs(np.random.normal(size=(4, 5)))
    ad = AnnData(x.copy())
    preproc.tfidf(ad, log_tf=True, log_idf=True)
    assert "%.3f" % ad.X[0, 0] == "4.659" and "%.3f" % ad.X[3, 0] == "4.770"
    out["dense_in"] = x
    out.update(csr_parts(ad.X, "dense_out"))

    ad = AnnData(x.copy())
    ad.layers["counts"] = ad.X.copy() + 1
    ad.X = None
    preproc.tfidf(ad, from_layer="counts")
    assert "%.3f" % ad.X[0, 0] == "2.856"
    out.update(csr_parts(ad.X, "dense_plus1_out"))

    # ---- reference test inputs: sparse rand(100,10) (test_atac_preproc.py:57-58) --
    np.random.seed(2020)
    xs = sp.rand(100, 10, density=0.2, format="csr")
    ad = AnnData(xs.copy())
    preproc.tfidf(ad, log_tf=True, log_idf=True)
    assert "%.3f" % ad.X[10, 9] == "18.749" and "%.3f" % ad.X[50, 5] == "0.000"
    out.update(csr_parts(xs, "sparse_in"))
    out.update(csr_parts(ad.X, "sparse_out"))

    # ---- option sweep on a seeded count matrix with edge cases -------------------
    rng = np.random.default_rng(7)
    n, d = 257, 131
    dens = sp.random(n, d, density=0.08, format="csr", random_state=rng, dtype=np.float32)
    dens.data = (1 + rng.poisson(0.6, size=dens.nnz)).astype(np.float32)
    dens = dens.tolil()
    dens[5, :] = 0  # empty row
    dens[:, 17] = 0  # empty column
    cnt = dens.tocsr()
    cnt.eliminate_zeros()
    cnt.sort_indices()
    out.update(csr_parts(cnt, "sweep_in"))
    sweeps = {
        "default": dict(),
        "nolog": dict(log_tf=False, log_idf=False),
        "logtfidf": dict(log_tf=False, log_idf=False, log_tfidf=True),
        "noscale": dict(scale_factor=None),
        "scale100": dict(scale_factor=100.0),
        "logtf_only": dict(log_idf=False),
        "logidf_only": dict(log_tf=False),
    }
    for name, kw in sweeps.items():
        for dt in (np.float32, np.float64):
            ad = AnnData(cnt.astype(dt))
            with np.errstate(divide="ignore", invalid="ignore"):
                res = preproc.tfidf(ad, inplace=False, **kw)
            out.update(csr_parts(res, f"sweep_{name}_{np.dtype(dt).name}"))

    # explicit stored zeros are dropped by scipy's SpGEMM
    ez = cnt.copy().astype(np.float32)
    ez.data[::11] = 0.0
    out.update(csr_parts(ez, "ezero_in"))
    ad = AnnData(ez.copy())
    with np.errstate(divide="ignore", invalid="ignore"):
        res = preproc.tfidf(ad, inplace=False)
    out.update(csr_parts(res, "ezero_out"))

    # integer counts are promoted to float64 by the reference
    ad = AnnData(cnt.astype(np.int32))
    with np.errstate(divide="ignore", invalid="ignore"):
        res = preproc.tfidf(ad, inplace=False)
    assert res.dtype == np.float64
    out.update(csr_parts(res, "sweep_int32"))

    np.savez_compressed(os.path.join(HERE, "tfidf_golden.npz"), **out)
    print("wrote tfidf_golden.npz with", len(out), "arrays")

    # ---- LSI: tfidf + lsi on a planted-topic matrix (no reference test exists) ----
    lsi_out = {}
    X = planted_topics_csr(600, 400, n_topics=12, density=0.06, seed=3, dtype=np.float64)
    ad = AnnData(X.copy())
    preproc.tfidf(ad)
    lsi_out.update(csr_parts(X, "counts"))
    lsi_out.update(csr_parts(ad.X, "tfidf"))
    for scale in (True, False):
        a2 = AnnData(ad.X.copy())
        tools.lsi(a2, scale_embeddings=scale, n_comps=12)
        tag = "scaled" if scale else "raw"
        lsi_out[f"X_lsi_{tag}"] = a2.obsm["X_lsi"]
        lsi_out[f"stdev_{tag}"] = a2.uns["lsi"]["stdev"]
        lsi_out[f"LSI_{tag}"] = a2.varm["LSI"]
    np.savez_compressed(os.path.join(HERE, "lsi_golden.npz"), **lsi_out)
    print("wrote lsi_golden.npz with", len(lsi_out), "arrays")


if __name__ == "__main__":
    main()
