#!/usr/bin/env python
"""Golden vectors for MOFA's data assembly by EXECUTING the reference's own code.

`muon.tl.mofa` hands the arithmetic to mofapy2 (absent here, not vendored, not installable: the MOFA oracle stays
unpinned) - but everything BEFORE that call is the reference's own code: `_set_mofa_data_from_mudata`
(/root/reference/muon/_core/tools.py:50-287) picks the data slot, expands the samples for `use_obs="union"`, subsets
features, orders the samples by group, names views / groups / samples and stores the per-group intercepts.  This script
loads the reference file where it lies, with stubs for the third-party modules it imports at the top (scanpy, h5py,
natsort, anndata, mudata, and mofapy2.build_model.utils whose `process_data` is replaced by the IDENTITY: the fixture
holds the matrices exactly as the reference hands them to mofapy2), runs that function on seeded MuData objects with a
recording model object, and writes what it set to tests/golden/mofa_prep_golden.npz.  tests/test_mofa_host.py compares
`muon_amd._core.tools._collect_views` with it.

Second part: the WHOLE `mofa()` (:290-708) around mofapy2 - the keyword routing into `set_data_options /
set_model_options / set_train_options`, and everything after `ent.save()`: reading the model file back, re-ordering the
factors by sample names, NaN rows outside the intersection, zero rows of `.varm["LFs"]` for unused features, the
`.uns["mofa"]` record and the variance table.  mofapy2's entry point is replaced by a recorder whose `save()` puts a
SEEDED model (random Z / W / R2 of the right shapes, mofapy2's dataset layout) into an in-memory stand-in for the HDF5
file that the stubbed `h5py.File` hands back.  The fixture (mofa_writeback_golden.npz) holds that model and what the
reference wrote into the MuData object; tests/test_mofa_host.py feeds the same model through `muon_amd.tl.mofa`'s
write-back.

Run (in the build container):  python tests/golden/make_mofa_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import pandas as pd
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = os.environ.get("MUON_REFERENCE", "/root/reference")

sys.dont_write_bytecode = True  # (no __pycache__ beside the fixtures: tests/test_layout.py audits this directory)
import make_wnn_golden as wnn_stubs  # noqa: E402  (the stubs preproc.py needs: tools.py imports it)
from muon_amd._containers import AnnData, MuData  # noqa: E402


def load_reference_tools():
    wnn_stubs.load_reference()  # muon._core.preproc (for `from .preproc import _sparse_csr_fast_knn`)
    h5 = types.ModuleType("h5py")
    h5.File = lambda name, mode="r": _FILES[name]
    sys.modules["h5py"] = h5
    run = types.ModuleType("mofapy2.run")
    ep = types.ModuleType("mofapy2.run.entry_point")
    ep.entry_point = RecordingEntryPoint
    sys.modules.update({"mofapy2.run": run, "mofapy2.run.entry_point": ep})
    ns = types.ModuleType("natsort")
    ns.natsorted = sorted
    sys.modules["natsort"] = ns
    sc = sys.modules["scanpy"]
    sc.logging = sys.modules["scanpy.logging"]
    mofapy2 = types.ModuleType("mofapy2")
    bm = types.ModuleType("mofapy2.build_model")
    utils = types.ModuleType("mofapy2.build_model.utils")
    utils.process_data = lambda data, likelihoods, data_opts, samples_groups: data  # IDENTITY: what mofapy2 is handed
    utils.guess_likelihoods = lambda data: (_ for _ in ()).throw(AssertionError("likelihoods are given in every case"))
    sys.modules.update({"mofapy2": mofapy2, "mofapy2.build_model": bm, "mofapy2.build_model.utils": utils})
    spec = importlib.util.spec_from_file_location("muon._core.tools", os.path.join(REF, "muon/_core/tools.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["muon._core.tools"] = m
    spec.loader.exec_module(m)
    return m


class _Node(dict):
    """h5py group / dataset stand-in: nested dicts, datasets are numpy arrays (h5py datasets index like arrays)"""

    def close(self):
        pass


_FILES = {}


class RecordingEntryPoint:
    """mofapy2.run.entry_point.entry_point: records the option calls; `save()` writes a seeded model of the right
    shapes in mofapy2's layout (expectations/Z/<group> [K, n_g], expectations/W/<view> [K, D], samples/<group>,
    views/views, groups/groups, model_options/likelihoods, variance_explained/r2_per_factor/<group> [M, K])"""

    last = None

    def __init__(self):
        self.dimensionalities = {}
        self.calls = {}
        RecordingEntryPoint.last = self

    def set_data_options(self, **kw):
        self.calls["data"] = kw
        self.data_opts = dict(kw)

    def set_model_options(self, **kw):
        self.calls["model"] = kw
        self.K = kw["factors"]

    def set_train_options(self, **kw):
        self.calls["train"] = kw

    def build(self):
        self.calls["built"] = True

    def run(self):
        self.calls["ran"] = True

    def save(self, outfile, save_data=True, save_parameters=False, expectations=None):
        self.calls["save"] = dict(save_data=save_data, save_parameters=save_parameters, expectations=expectations)
        rng = np.random.default_rng(11)
        K = self.K
        groups = self.data_opts["groups_names"]
        views = self.data_opts["views_names"]
        f = _Node()
        f["expectations"] = _Node(Z=_Node(), W=_Node())
        f["samples"] = _Node()
        f["variance_explained"] = _Node(r2_per_factor=_Node())
        for g, names in zip(groups, self.data_opts["samples_names"]):
            f["expectations"]["Z"][g] = rng.standard_normal((K, len(names)))
            f["samples"][g] = np.asarray(names, dtype="S")
            f["variance_explained"]["r2_per_factor"][g] = rng.random((len(views), K)) * 10
        for m, d in zip(views, self.dimensionalities["D"]):
            f["expectations"]["W"][m] = rng.standard_normal((K, d))
        f["views"] = _Node(views=np.asarray(views, dtype="S"))
        f["groups"] = _Node(groups=np.asarray(groups, dtype="S"))
        f["model_options"] = _Node(likelihoods=np.asarray(self.likelihoods, dtype="S"))
        _FILES[outfile] = f


class RecordingModel:
    """what `_set_mofa_data_from_mudata` reads and writes on mofapy2's entry point object"""

    def __init__(self):
        self.dimensionalities = {}

    def set_data_options(self):
        self.data_opts = {}


def cases():
    rng = np.random.default_rng(7)
    out = {}
    # (1) two views, same cells in a shuffled order per modality? no: same cells, groups interleaved (like the
    #     reference's own test set-up, tests/test_muon_tools.py: samples of two groups shuffled)
    n = 30
    names = np.array([f"c{i}" for i in range(n)])
    y1 = rng.standard_normal((n, 6))
    y2 = sp.random(n, 9, density=0.35, format="csr", random_state=1, dtype=np.float64)
    a1, a2 = AnnData(y1.copy()), AnnData(y2.copy())
    a1.obs_names, a2.obs_names = names, names
    md = MuData({"rna": a1, "atac": a2})
    md.obs["grp"] = rng.choice(["B", "A", "C"], size=n)
    out["groups"] = (md, dict(groups_label="grp", likelihoods=["gaussian", "gaussian"]))
    # (2) use_obs = "union": the modalities share only part of the cells
    b1, b2 = AnnData(y1[:22].copy()), AnnData(y2[8:].copy())
    b1.obs_names, b2.obs_names = names[:22], names[8:]
    out["union"] = (MuData({"rna": b1, "atac": b2}), dict(use_obs="union", likelihoods=["gaussian", "gaussian"]))
    # (3) use_obs = "intersection"
    c1, c2 = AnnData(y1[:22].copy()), AnnData(y2[8:].copy())
    c1.obs_names, c2.obs_names = names[:22], names[8:]
    out["intersection"] = (MuData({"rna": c1, "atac": c2}), dict(use_obs="intersection", likelihoods=["gaussian", "gaussian"]))
    # (4) features_subset + use_layer
    d1, d2 = AnnData(y1.copy()), AnnData(y2.copy())
    d1.obs_names, d2.obs_names = names, names
    d1.layers["lognorm"] = y1 * 2.0 + 1.0
    d2.layers["lognorm"] = (y2 * 3.0).tocsr()
    d1.var["highly_variable"] = rng.random(6) < 0.6
    d2.var["highly_variable"] = rng.random(9) < 0.6
    out["subset_layer"] = (MuData({"rna": d1, "atac": d2}),
                           dict(use_layer="lognorm", features_subset="highly_variable", likelihoods=["gaussian", "gaussian"]))
    return out


def pack_inputs(tag, md, out):
    for m, a in md.mod.items():
        x = a.X
        if sp.issparse(x):
            x = x.tocsr()
            out[f"{tag}_{m}_X_data"], out[f"{tag}_{m}_X_indices"], out[f"{tag}_{m}_X_indptr"] = x.data, x.indices, x.indptr
            out[f"{tag}_{m}_X_shape"] = np.asarray(x.shape)
        else:
            out[f"{tag}_{m}_X"] = np.asarray(x)
        out[f"{tag}_{m}_obs_names"] = np.asarray(a.obs_names, dtype="U")
        for k, v in a.layers.items():
            out[f"{tag}_{m}_layer_{k}"] = v.toarray() if sp.issparse(v) else np.asarray(v)
            out[f"{tag}_{m}_layer_{k}_sparse"] = np.array([int(sp.issparse(v))])
        if "highly_variable" in a.var.columns:
            out[f"{tag}_{m}_hv"] = np.asarray(a.var["highly_variable"].values, dtype=bool)
    if "grp" in md.obs.columns:
        out[f"{tag}_grp"] = np.asarray(md.obs["grp"].values, dtype="U")
    out[f"{tag}_obs_names"] = np.asarray(md.obs.index.values, dtype="U")


def main():
    tools = load_reference_tools()
    out = {}
    for tag, (md, kw) in cases().items():
        pack_inputs(tag, md, out)
        model = RecordingModel()
        tools._set_mofa_data_from_mudata(model, md, **kw)
        for m, x in enumerate(model.data):
            out[f"{tag}_data{m}"] = np.asarray(x, dtype=np.float64)
        out[f"{tag}_dims"] = np.array([model.dimensionalities["M"], model.dimensionalities["G"], model.dimensionalities["N"]]
                                      + list(model.dimensionalities["D"]))
        out[f"{tag}_views_names"] = np.asarray(model.data_opts["views_names"], dtype="U")
        out[f"{tag}_groups_names"] = np.asarray(model.data_opts["groups_names"], dtype="U")
        out[f"{tag}_samples_groups"] = np.asarray(model.data_opts["samples_groups"], dtype="U")
        out[f"{tag}_samples_names"] = np.asarray(np.concatenate([np.asarray(s, dtype="U") for s in model.data_opts["samples_names"]]))
        out[f"{tag}_samples_per_group"] = np.array([len(s) for s in model.data_opts["samples_names"]])
        for m in range(len(model.data)):
            out[f"{tag}_intercepts{m}"] = np.stack(model.intercepts[m])
        out[f"{tag}_likelihoods"] = np.asarray(model.likelihoods, dtype="U")
        print(tag, "dims", out[f"{tag}_dims"], "groups", list(out[f"{tag}_groups_names"]))
    np.savez_compressed(os.path.join(HERE, "mofa_prep_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "mofa_prep_golden.npz"), sum(v.nbytes for v in out.values()), "bytes")
    writeback(tools)


def writeback_cases():
    """MuData objects + mofa() keywords: groups in shuffled order with a feature subset; an intersection; a union.
    (Cell names are zero-padded, i.e. already in sorted order: for unsorted names the reference's mask assignment of the
    intersection case gives cells each other's factors, DESIGN.md 6 - nothing to pin there.)"""
    rng = np.random.default_rng(21)
    n = 24
    names = np.array([f"cell{i:02d}" for i in range(n)])
    y1 = rng.standard_normal((n, 7))
    y2 = sp.random(n, 5, density=0.5, format="csr", random_state=2, dtype=np.float64)
    out = {}
    a1, a2 = AnnData(y1.copy()), AnnData(y2.copy())
    a1.obs_names, a2.obs_names = names, names
    a1.var["highly_variable"] = np.array([True, False, True, True, False, True, True])
    a2.var["highly_variable"] = np.array([True, True, False, True, True])
    md = MuData({"rna": a1, "atac": a2})
    md.obs["batch"] = rng.choice(["b2", "b1"], size=n)
    md.var["highly_variable"] = np.concatenate([a1.var["highly_variable"].values, a2.var["highly_variable"].values])
    out["groups_subset"] = (md, dict(groups_label="batch", use_var="highly_variable", n_factors=4, likelihoods="gaussian",
                                     n_iterations=7, convergence_mode="medium", seed=3, scale_views=True, quiet=True,
                                     outfile="/tmp/golden_a.hdf5"))
    b1, b2 = AnnData(y1[:18].copy()), AnnData(y2[5:].copy())
    b1.obs_names, b2.obs_names = names[:18], names[5:]
    out["intersection"] = (MuData({"rna": b1, "atac": b2}),
                           dict(use_obs="intersection", use_var=None, n_factors=3, likelihoods=["gaussian", "gaussian"],
                                quiet=True, outfile="/tmp/golden_b.hdf5"))
    c1, c2 = AnnData(y1[:18].copy()), AnnData(y2[5:].copy())
    c1.obs_names, c2.obs_names = names[:18], names[5:]
    out["union"] = (MuData({"rna": c1, "atac": c2}),
                    dict(use_obs="union", use_var=None, n_factors=3, likelihoods=["gaussian", "gaussian"], quiet=True,
                         use_float32=True, outfile="/tmp/golden_c.hdf5"))
    return out


def writeback(tools):
    out = {}
    for tag, (md, kw) in writeback_cases().items():
        pack_inputs(tag, md, out)
        if "batch" in md.obs.columns:
            out[f"{tag}_batch"] = np.asarray(md.obs["batch"].values, dtype="U")
        tools.mofa(md, **kw)
        ent = RecordingEntryPoint.last
        f = _FILES[kw["outfile"]]
        for g in f["expectations"]["Z"]:
            out[f"{tag}_model_Z_{g}"] = f["expectations"]["Z"][g]
            out[f"{tag}_model_samples_{g}"] = f["samples"][g].astype("U")
            out[f"{tag}_model_r2_{g}"] = f["variance_explained"]["r2_per_factor"][g]
        for m in f["expectations"]["W"]:
            out[f"{tag}_model_W_{m}"] = f["expectations"]["W"][m]
        out[f"{tag}_model_groups"] = f["groups"]["groups"].astype("U")
        out[f"{tag}_X_mofa"] = np.asarray(md.obsm["X_mofa"])
        out[f"{tag}_LFs"] = np.asarray(md.varm["LFs"])
        u = md.uns["mofa"]
        for sect in ("data", "model", "training"):
            for k, v in u["params"][sect].items():
                out[f"{tag}_param_{sect}_{k}"] = np.asarray("None" if v is None else v)
        for view, v in u["variance"].items():
            if isinstance(v, dict):
                for g, arr in v.items():
                    out[f"{tag}_variance_{view}_{g}"] = np.asarray(arr)
            else:
                out[f"{tag}_variance_{view}"] = np.asarray(v)
        for sect in ("data", "model", "train"):
            for k, v in ent.calls[sect].items():
                out[f"{tag}_call_{sect}_{k}"] = np.asarray("None" if v is None else v)
        print(tag, "X_mofa", out[f"{tag}_X_mofa"].shape, "LFs", out[f"{tag}_LFs"].shape, "calls", sorted(ent.calls))
    np.savez_compressed(os.path.join(HERE, "mofa_writeback_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "mofa_writeback_golden.npz"), sum(v.nbytes for v in out.values()), "bytes")


if __name__ == "__main__":
    main()
