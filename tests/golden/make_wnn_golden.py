#!/usr/bin/env python
"""Golden vectors for weighted nearest neighbours by EXECUTING the reference's own code.

/root/reference/muon/_core/preproc.py cannot be imported as it is (numba, umap-learn, pynndescent, scanpy, anndata and
mudata are absent from the build image).  This script installs stub modules for the THIRD-PARTY pieces only and then
loads the reference file where it lies and runs its own `neighbors()` (:264-640) and `l2norm` (:182-262):

  * numba.njit / prange          -> identity decorators (the njit functions of the file are plain Python);
  * pynndescent.distances.euclidean, pynndescent.sparse.sparse_euclidean / sparse_jaccard -> their published
    definitions (Jaccard DISTANCE of two sorted index lists; Euclidean distance);
  * umap.umap_.nearest_neighbors -> an EXHAUSTIVE search with the metric the reference passes (its own
    `_jaccard_euclidean_metric` for the kernel bandwidths, a metric name for the candidates): the exact answer
    NN-descent approximates, ties by index;
  * scanpy: `logging`, `_choose_representation` (X or .obsm[use_rep]) and the UMAP connectivities
    (`scanpy.neighbors._connectivity.umap` -> oracle/wnn_oracle.fuzzy_simplicial_set: OURS - the connectivities of the
    fixture are therefore not a reference statement and are not stored);
  * anndata / mudata -> muon_amd._containers.

Everything else - the bandwidth metric, the affinity ratios, the softmax weights, the candidate union, the affinities,
`_sparse_csr_fast_knn`, the slots and parameters written - is the reference's own statements executing.  Inputs and
outputs go to tests/golden/wnn_golden.npz; /root/reference does not exist on the GPU box: tests read only the fixture.

Run (in the build container):  python tests/golden/make_wnn_golden.py
"""
import importlib.metadata
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.sparse as sp
from scipy.spatial.distance import cdist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("MUON_REFERENCE", "/root/reference")

from muon_amd._containers import AnnData, MuData  # noqa: E402
from oracle import wnn_oracle  # noqa: E402


def _euclidean(x, y):
    return float(np.sqrt(np.sum((np.asarray(x, dtype=np.float64) - np.asarray(y, dtype=np.float64)) ** 2)))


def _sparse_jaccard(ind1, data1, ind2, data2):
    """pynndescent.sparse.sparse_jaccard: 1 - |A & B| / |A | B| of the two index sets (0 when both are empty)"""
    a, b = set(int(i) for i in ind1), set(int(i) for i in ind2)
    union = len(a | b)
    if union == 0:
        return 0.0
    return float(union - len(a & b)) / float(union)


def _sparse_euclidean(ind1, data1, ind2, data2):
    d = {}
    for i, v in zip(ind1, data1):
        d[int(i)] = d.get(int(i), 0.0) + float(v)
    for i, v in zip(ind2, data2):
        d[int(i)] = d.get(int(i), 0.0) - float(v)
    return float(np.sqrt(sum(v * v for v in d.values())))


def _nearest_neighbors(X, n_neighbors, metric, metric_kwds=None, angular=False, random_state=None, low_memory=True,
                       **_ignored):
    """umap.umap_.nearest_neighbors, exhaustively: (indices [n, k], distances [n, k], None), ascending, ties by index"""
    X = np.asarray(X)
    n = X.shape[0]
    if callable(metric):
        D = np.empty((n, n))
        kw = metric_kwds or {}
        for i in range(n):
            for j in range(n):
                D[i, j] = metric(X[i], X[j], **kw)
    else:
        D = cdist(X.astype(np.float64), X.astype(np.float64), metric=metric)
    order = np.argsort(D, axis=1, kind="stable")[:, :n_neighbors]
    return order, np.take_along_axis(D, order, axis=1), None


def _njit(*args, **kwargs):
    if args and callable(args[0]):
        return args[0]
    return lambda f: f


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("numba", njit=_njit, prange=range)
    mod("pynndescent")
    mod("pynndescent.distances", euclidean=_euclidean)
    mod("pynndescent.sparse", sparse_euclidean=_sparse_euclidean, sparse_jaccard=_sparse_jaccard)
    mod("umap")
    mod("umap.umap_", nearest_neighbors=_nearest_neighbors)
    mod("anndata", AnnData=AnnData)
    mod("mudata", MuData=MuData)
    logging = mod("scanpy.logging", info=lambda *a, **k: None, debug=lambda *a, **k: None, warning=lambda *a, **k: None)

    def choose(adata, use_rep=None, n_pcs=None, silent=False):
        if use_rep in (None, "X"):
            return adata.X
        return adata.obsm[use_rep]

    mod("scanpy.tools")
    mod("scanpy.tools._utils", _choose_representation=choose)
    mod("scanpy.neighbors")
    mod("scanpy.neighbors._connectivity",
        umap=lambda knn_indices, knn_dists, n_obs, n_neighbors: wnn_oracle.fuzzy_simplicial_set(knn_indices, knn_dists,
                                                                                               n_obs, n_neighbors))
    sc = mod("scanpy", logging=logging)
    sc.tools = sys.modules["scanpy.tools"]
    real_version = importlib.metadata.version

    def version(name):
        return "1.10.4" if name == "scanpy" else real_version(name)

    importlib.metadata.version = version
    muon = types.ModuleType("muon")
    muon.__path__ = [os.path.join(REF, "muon")]
    sys.modules["muon"] = muon
    core = types.ModuleType("muon._core")
    core.__path__ = [os.path.join(REF, "muon", "_core")]
    sys.modules["muon._core"] = core


def load_reference():
    _install_stubs()
    spec = importlib.util.spec_from_file_location("muon._core.preproc", os.path.join(REF, "muon/_core/preproc.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["muon._core.preproc"] = m
    spec.loader.exec_module(m)
    return m


def _case(seed, n, p1, p2, k1, k2, clusters=6):
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, clusters, n)
    x1 = rng.standard_normal((clusters, p1))[lab] * 2.0 + rng.standard_normal((n, p1))
    x2 = rng.standard_normal((clusters, p2))[lab] * 1.5 + rng.standard_normal((n, p2))
    return x1, x2, k1, k2


def _mdata(x1, x2, k1, k2):
    mods = {}
    for name, x, k in (("rna", x1, k1), ("atac", x2, k2)):
        ad = AnnData(x.copy())
        dist, conn, uns = wnn_oracle.knn_graph(x, n_neighbors=k)  # (the per-modality input sc.pp.neighbors would provide)
        ad.obsp["distances"], ad.obsp["connectivities"] = dist, conn
        ad.uns["neighbors"] = uns
        mods[name] = ad
    return MuData(mods)


def main():
    ref = load_reference()
    out = {}
    cases = {"a": _case(1, 180, 12, 7, 12, 10), "b": _case(2, 120, 5, 9, 8, 8, clusters=4)}
    for tag, (x1, x2, k1, k2) in cases.items():
        md = _mdata(x1, x2, k1, k2)
        kw = dict(n_bandwidth_neighbors=10, n_multineighbors=40) if tag == "a" else dict(n_bandwidth_neighbors=6,
                                                                                          n_multineighbors=30, n_neighbors=9)
        ref.neighbors(md, **kw)
        d = md.obsp["distances"]
        d = sp.csr_matrix((np.asarray(d.data), np.asarray(d.indices), np.asarray(d.indptr)), shape=d.shape)
        out[f"{tag}_x1"], out[f"{tag}_x2"] = x1, x2
        out[f"{tag}_k"] = np.array([k1, k2])
        out[f"{tag}_kw"] = np.array([kw["n_bandwidth_neighbors"], kw["n_multineighbors"], kw.get("n_neighbors", -1)])
        for m in ("rna", "atac"):
            g = md.mod[m].obsp["distances"].tocsr()
            out[f"{tag}_{m}_g_data"], out[f"{tag}_{m}_g_indices"], out[f"{tag}_{m}_g_indptr"] = g.data, g.indices, g.indptr
            out[f"{tag}_{m}_weight"] = np.asarray(md.obs[f"{m}:mod_weight"], dtype=np.float64)
        out[f"{tag}_dist_data"], out[f"{tag}_dist_indices"], out[f"{tag}_dist_indptr"] = d.data, d.indices, d.indptr
        p = md.uns["neighbors"]["params"]
        out[f"{tag}_n_neighbors"] = np.array([p["n_neighbors"]])
        print(tag, "n_neighbors", p["n_neighbors"], "entries per row", np.unique(np.diff(d.indptr)),
              "mean rna weight %.4f" % out[f"{tag}_rna_weight"].mean())
    # l2norm (preproc.py:182-262) on a dense and a CSR matrix
    rng = np.random.default_rng(3)
    xd = rng.standard_normal((20, 6))
    xs = sp.random(20, 30, density=0.3, format="csr", random_state=4, dtype=np.float64)
    ad = AnnData(xd.copy())
    ref.l2norm(ad)
    out["l2_dense_in"], out["l2_dense_out"] = xd, np.asarray(ad.X)
    ad = AnnData(xs.copy())
    ref.l2norm(ad)
    got = ad.X.tocsr()
    out["l2_csr_in_data"], out["l2_csr_in_indices"], out["l2_csr_in_indptr"] = xs.data, xs.indices, xs.indptr
    out["l2_csr_out_data"] = np.asarray(got.data)
    np.savez_compressed(os.path.join(HERE, "wnn_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "wnn_golden.npz"), sum(v.nbytes for v in out.values()), "bytes")


if __name__ == "__main__":
    main()
