"""muon.tl.mofa on MI355X.

Host side mirrors /root/reference/muon/_core/tools.py:52-287 (`_set_mofa_data_from_mudata`)
and :290-708 (`mofa`): same 40-keyword signature, same validation errors, same write-back
into ``.obsm["X_mofa"]``, ``.varm["LFs"]`` and ``.uns["mofa"]``.  The training itself
(`ent.build(); ent.run()`, :583-585, third-party mofapy2) is replaced by
``mofa_engine.MofaEngine``: HIP kernels for the factor / weight sweeps and the sparse
products, PyTorch-ROCm for the dense blocks.

Deliberate deviations, all explicit:
  * sparse modalities are NOT densified (tools.py:117-141 does ``.todense()``);
  * only the Gaussian likelihood is implemented: views whose likelihood the reference's default
    would GUESS as bernoulli / poisson are modelled as gaussian with a warning; asking for them
    explicitly, SVI, MEFISTO (smooth_*) and ``spikeslab_factors`` raise NotImplementedError;
  * a bad ``groups_label`` raises ValueError where the reference calls ``sys.exit()`` (:106-113);
  * the model file is HDF5 (mofapy2's layout) when h5py is importable; otherwise ``outfile`` +
    ".npz" (NumPy archive, with a warning); results reach the write-back directly, not through it.
"""
from __future__ import annotations

import logging
import os
from functools import reduce
from time import strftime
from typing import Any, Iterable, List, Mapping, Optional, Union
from warnings import warn

import numpy as np
import pandas as pd
import torch
from scipy.sparse import csr_matrix, issparse

from .._containers import MuData, is_anndata, is_mudata

logger = logging.getLogger("muon_amd")


def _guess_likelihood(x) -> str:
    """mofapy2.build_model.utils.guess_likelihoods semantics (tools.py:272-273): binary ->
    bernoulli, integer -> poisson, otherwise gaussian."""
    v = x.data if issparse(x) else np.asarray(x)
    v = v[~np.isnan(v)] if v.dtype.kind == "f" else v
    if v.size and np.all(np.isin(v, (0, 1))):
        return "bernoulli"
    if v.size and np.all(v == np.round(v)):
        return "poisson"
    return "gaussian"


def _collect_views(mdata, groups_label, use_raw, use_layer, likelihoods, features_subset, use_obs):
    """`_set_mofa_data_from_mudata` (tools.py:52-287) without densification."""
    obs_names = mdata.obs.index.values
    mods = list(mdata.mod.keys())
    if use_obs == "intersection":
        common = reduce(np.intersect1d, [v.obs_names.values for v in mdata.mod.values()])
        # `mdata = mdata[common_obs]` in the reference (:99-101): np.intersect1d SORTS, so the samples of the model
        # are in sorted name order - found by executing the reference (tests/golden/make_mofa_golden.py); r03 kept
        # mdata's order, which changes which sample gets which row of the seeded initialisation
        obs_names = np.asarray(common)

    if groups_label is not None:
        if not isinstance(groups_label, str):
            raise ValueError("groups_label should be a string present in the observations column names")
        if groups_label not in mdata.obs.columns:
            raise ValueError("{} is not in observations names".format(groups_label))

    views = []
    for m in mods:
        adata = mdata.mod[m]
        if use_layer:
            if use_layer not in adata.layers:
                raise ValueError("Layer {} does not exist".format(use_layer))
            x = adata.layers[use_layer]
        elif use_raw:
            if getattr(adata, "raw", None) is None:
                raise ValueError(f"modality {m} has no .raw")
            x = adata.raw[:, adata.var_names].X
        else:
            x = adata.X
        # place this modality's samples into the common sample axis (union expands with missing)
        pos = pd.Index(obs_names).get_indexer(adata.obs_names)
        have = pos >= 0
        n = len(obs_names)
        if issparse(x):
            x = x.tocsr()[np.nonzero(have)[0]]
            coo = x.tocoo()
            rows = pos[have][coo.row]
            full = csr_matrix((coo.data, (rows, coo.col)), shape=(n, x.shape[1]))
            missing = np.ones(n, dtype=bool)
            missing[pos[have]] = False
            full._missing_rows = missing if missing.any() else None
            x = full
        else:
            full = np.full((n, np.asarray(x).shape[1]), np.nan)
            full[pos[have]] = np.asarray(x, dtype=np.float64)[have]
            x = full
        if features_subset is not None:
            if features_subset not in adata.var.columns:
                raise KeyError(f"There is no column {features_subset} in .var for modality {m}")
            sel = np.asarray(adata.var[features_subset].values).astype(bool)
            miss = getattr(x, "_missing_rows", None)
            x = x[:, sel] if not issparse(x) else x[:, np.nonzero(sel)[0]]
            if issparse(x):
                x._missing_rows = miss
        views.append(x)

    guessed = likelihoods is None
    if guessed:
        likelihoods = [_guess_likelihood(v) for v in views]
    assert len(likelihoods) == len(views), "Please specify one likelihood for each view"
    assert set(likelihoods).issubset({"gaussian", "bernoulli", "poisson"}), \
        "Available likelihoods are 'gaussian', 'bernoulli', 'poisson'"
    # (r03: poisson / bernoulli views and element-wise NaN run through _core/mofa_general.py - the
    #  guessed likelihoods are the ones the model is fitted with, as in the reference; r02 fitted
    #  guessed count likelihoods as gaussian with a warning)

    obs = mdata.obs.loc[obs_names]
    if groups_label is None:
        groups = np.zeros(len(obs_names), dtype=np.int64)
        group_names = ["group1"]
    else:
        labels = obs[groups_label].astype(str).values
        group_names = list(pd.unique(labels))  # order of first appearance (groupby sort=False)
        groups = pd.Index(group_names).get_indexer(labels).astype(np.int64)
    return views, groups, group_names, obs_names, likelihoods


def _has_elementwise_nan(v) -> bool:
    """NaN entries that are not whole missing samples (those are row masks: use_obs='union')."""
    if issparse(v):
        return bool(np.isnan(v.data).any())
    a = np.isnan(np.asarray(v))
    return bool((a.any(axis=1) & ~a.all(axis=1)).any())


def mofa(
    data,
    groups_label: bool = None,
    use_raw: bool = False,
    use_layer: str = None,
    use_var: Optional[str] = "highly_variable",
    use_obs: Optional[str] = None,
    likelihoods: Optional[Union[str, List[str]]] = None,
    n_factors: int = 10,
    scale_views: bool = False,
    scale_groups: bool = False,
    center_groups: bool = True,
    ard_weights: bool = True,
    ard_factors: bool = True,
    spikeslab_weights: bool = True,
    spikeslab_factors: bool = False,
    n_iterations: int = 1000,
    convergence_mode: str = "fast",
    use_float32: bool = False,
    gpu_mode: bool = False,
    gpu_device: Optional[bool] = None,
    svi_mode: bool = False,
    svi_batch_size: float = 0.5,
    svi_learning_rate: float = 1.0,
    svi_forgetting_rate: float = 0.5,
    svi_start_stochastic: int = 1,
    smooth_covariate: Optional[str] = None,
    smooth_warping: bool = False,
    smooth_kwargs: Optional[Mapping[str, Any]] = None,
    save_parameters: bool = False,
    save_data: bool = True,
    save_metadata: bool = True,
    seed: int = 1,
    outfile: Optional[str] = None,
    expectations: Optional[List[str]] = None,
    save_interrupted: bool = True,
    verbose: bool = False,
    quiet: bool = True,
    copy: bool = False,
    *,
    backend=None,
    comm=None,
):
    """
    Run Multi-Omics Factor Analysis (MOFA+, Gaussian likelihood) on the GPU.

    Same parameters as ``muon.tl.mofa`` (reference tools.py:290-416).  ``gpu_mode`` /
    ``gpu_device`` are accepted and ignored (this implementation always runs on the GPU).
    Keyword-only extras: ``backend`` (operator set) and ``comm`` (samples sharded over ranks).
    """
    if is_anndata(data):
        logger.info("Wrapping an AnnData object into an MuData container")
        mdata = MuData({"data": data})
        # Modality name is used as a prefix by default
        mdata.obs = data.obs
    elif is_mudata(data):
        mdata = data
    else:
        raise TypeError("Expected an MuData object")

    if outfile is None:
        outfile = os.path.join("/tmp", "mofa_{}.hdf5".format(strftime("%Y%m%d-%H%M%S")))

    if use_var and use_var not in data.var.columns:
        warn(f"There is no column {use_var} in the provided object")
        use_var = None
    common_obs = None
    if is_mudata(data):
        common_obs = reduce(np.intersect1d, [v.obs_names.values for k, v in mdata.mod.items()])
        if len(common_obs) != mdata.n_obs:
            if not use_obs:
                raise IndexError(
                    "Not all the observations are the same across modalities. Please run `mdata.intersect_obs()` to subset the data or devise a strategy with `use_obs` ('union' or 'intersection')"
                )
            elif use_obs not in ["union", "intersection"]:
                raise ValueError(
                    f"Expected `use_obs` argument to be 'union' or 'intersection', not '{use_obs}'"
                )
        else:
            use_obs = None

    if svi_mode:
        raise NotImplementedError("stochastic variational inference (svi_mode) is not implemented")
    if smooth_covariate is not None or smooth_warping or smooth_kwargs:
        raise NotImplementedError("MEFISTO (smooth_covariate / smooth_warping) is not implemented")
    if convergence_mode not in ("fast", "medium", "slow"):
        raise ValueError("convergence_mode must be 'fast', 'medium' or 'slow'")

    lik = likelihoods
    if lik is not None and (isinstance(lik, str) and isinstance(lik, Iterable)):
        lik = [lik for _ in range(len(mdata.mod))]

    logger.info("Setting data from MuData object...")
    views, groups, group_names, obs_used, lik = _collect_views(
        mdata, groups_label, use_raw, use_layer, lik, use_var, use_obs
    )

    if backend is None:
        from .._backend import get_backend

        backend = get_backend()  # raises without a GPU: no CPU fallback
    from .mofa_engine import MofaEngine

    logger.info("Building the model...")
    kw = dict(dtype=torch.float32 if use_float32 else torch.float64,
              center_groups=center_groups, scale_views=scale_views, scale_groups=scale_groups,
              ard_weights=ard_weights, ard_factors=ard_factors, spikeslab_weights=spikeslab_weights,
              seed=seed, comm=comm)
    n_groups = int(np.max(groups)) + 1 if len(groups) else 1
    wide = int(n_factors) > 32 or (n_groups * int(n_factors) > 64 and any(issparse(v) for v in views))
    if any(l != "gaussian" for l in lik) or any(_has_elementwise_nan(v) for v in views) or spikeslab_factors or wide:
        # pseudo-data likelihoods / element-wise missing values: element-wise precisions, walked in
        # row chunks (the sparse modalities stay CSR on the device).  r06: also the model options the two-pass engine's
        # kernels are not instantiated for - more than 32 factors (tools.py:298 takes any int), more than 64 stacked
        # (group, factor) columns against a sparse view, spikeslab_factors=True (tools.py:305,486) - the same model on
        # the general engine's chunk passes (slower: those are rare settings; n_factors defaults to 10)
        from .mofa_general import GeneralMofaEngine

        eng = GeneralMofaEngine(backend, views, list(lik), groups, n_factors, spikeslab_factors=bool(spikeslab_factors), **kw)
    else:
        eng = MofaEngine(backend, views, groups, n_factors, **kw)
    logger.info("Running the model...")
    eng.run(n_iterations=n_iterations, convergence_mode=convergence_mode)
    res = eng.results(sort_factors=True)

    logger.info("Saving the model...")
    try:
        written = _save_model(outfile, res, list(mdata.mod.keys()), group_names, obs_used, groups, expectations)
    except Exception as e:  # noqa: BLE001  (a failing save must not lose the fit: the slots below are still written)
        warn(f"Cannot save the model to {outfile}: {e!r}")
        written = None

    if copy:
        data = data.copy()

    # Factors: rows follow the order of the samples in data.obs (tools.py:604-627)
    z = res["Z"]
    if use_obs == "intersection":
        # the model's samples are the common cells in SORTED name order (np.intersect1d, like the reference :99-101);
        # every cell gets its own row back, by name.  (The reference assigns the sorted-order rows through a boolean
        # mask in data.obs order, :617-621 - the same thing only when the names happen to be sorted; otherwise each
        # common cell receives another cell's factors: not reproduced.)
        xm = np.full((data.n_obs, z.shape[1]), np.nan)
        if os.environ.get("MUON_AMD_MOFA_INTERSECTION_MASK", "0") == "1":
            # the reference's statement, bit for bit (:615-621): the model's rows - in sorted-name order - land in the
            # True positions of the mask in data.obs order.  Identical to the default when the names are sorted.
            xm[data.obs.index.isin(pd.Index(obs_used))] = z
        else:
            xm[data.obs.index.get_indexer(pd.Index(obs_used))] = z
        data.obsm["X_mofa"] = xm
    else:
        data.obsm["X_mofa"] = z

    # Weights (tools.py:629-641)
    w = np.concatenate(res["W"], axis=0)
    if use_var:
        lfs = np.zeros(shape=(data.n_vars, w.shape[1]))
        lfs[np.asarray(data.var[use_var]).astype(bool)] = w
        data.varm["LFs"] = lfs
    else:
        data.varm["LFs"] = w

    # Parameters (tools.py:653-678)
    data.uns["mofa"] = {
        "params": {
            "data": {
                "groups_label": groups_label,
                "use_raw": use_raw,
                "use_layer": use_layer,
                "likelihoods": np.asarray(lik).astype(str),
                "features_subset": use_var,
                "use_obs": use_obs,
                "scale_views": scale_views,
                "scale_groups": scale_groups,
                "center_groups": center_groups,
                "use_float32": use_float32,
            },
            "model": {
                "ard_factors": ard_factors,
                "ard_weights": ard_weights,
                "spikeslab_weights": spikeslab_weights,
                "spikeslab_factors": spikeslab_factors,
                "n_factors": n_factors,
            },
            "training": {
                "n_iterations": n_iterations,
                "convergence_mode": convergence_mode,
                "gpu_mode": gpu_mode,
                "seed": seed,
            },
        },
        "elbo": np.asarray(res["elbo"]),
        # (not in the reference) the file actually written: without h5py the archive is `outfile`.npz
        "model_file": str(written),
    }
    # Variance explained, R2 in % per factor (tools.py:681-697)
    variance = {m: {} for m in mdata.mod}
    for i, m in enumerate(mdata.mod):
        if len(group_names) > 1:
            for gi, g in enumerate(group_names):
                variance[m][g] = res["r2"][i, gi, :]
        else:
            variance[m] = res["r2"][i, 0, :]
    data.uns["mofa"]["variance"] = variance

    if copy:
        return data
    else:
        if not quiet:
            print("Saved MOFA embeddings in .obsm['X_mofa'] slot and their loadings in .varm['LFs'].")

    return None


def _model_datasets(res, view_names, group_names, obs_names, groups):
    """The model as {dataset path: array} in mofapy2's HDF5 layout - the paths and shapes the reference reads back
    (/root/reference/muon/_core/tools.py:604-641: expectations/Z/<group> [factors, samples], expectations/W/<view>
    [factors, features], samples/<group>) plus views/views, groups/groups, variance_explained/r2_per_factor/<group>
    [views, factors] and training_stats/elbo.  Both writers below store exactly this mapping."""
    def _bytes(names):  # UTF-8 byte strings (h5py / mofapy2's convention; `.astype("S")` raises on non-ASCII names)
        return np.char.encode(np.asarray(names).astype(str), "utf-8") if len(names) else np.asarray([], dtype="S1")

    d = {}
    for gi, g in enumerate(group_names):
        d[f"expectations/Z/{g}"] = np.ascontiguousarray(res["Z"][groups == gi].T)
        d[f"samples/{g}"] = _bytes(np.asarray(obs_names)[groups == gi])
        d[f"variance_explained/r2_per_factor/{g}"] = np.ascontiguousarray(res["r2"][:, gi, :])
    for m, w in zip(view_names, res["W"]):
        d[f"expectations/W/{m}"] = np.ascontiguousarray(np.asarray(w).T)
    d["views/views"] = _bytes(view_names)
    d["groups/groups"] = _bytes(group_names)
    d["training_stats/elbo"] = np.asarray(res["elbo"])
    return d


def _save_model(outfile, res, view_names, group_names, obs_names, groups, expectations):
    """Model file (tools.py:600-602 ent.save): HDF5 with mofapy2's group layout when h5py is
    importable.  Without h5py (this image) the SAME datasets are written as a NumPy .npz whose keys are the HDF5
    paths - under a name that says so (``outfile`` + ".npz" unless it already ends in .npz): a file called *.hdf5
    that no MOFA tool can open helps nobody, and a converter is one loop
    (``for k in z.files: h5.create_dataset(k, data=z[k])``).  Returns the path written."""
    data = _model_datasets(res, view_names, group_names, obs_names, groups)
    try:
        import h5py  # noqa: F401
    except Exception:  # noqa: BLE001
        if not str(outfile).endswith(".npz"):
            warn(f"h5py is not importable: the model is saved as NumPy archive {outfile}.npz, not as HDF5")
            outfile = str(outfile) + ".npz"
        try:
            with open(outfile, "wb") as f:
                np.savez(f, **data)
        except OSError as e:  # pragma: no cover
            warn(f"Cannot save the model to {outfile}: {e}")
        return outfile
    import h5py

    with h5py.File(outfile, "w") as f:  # pragma: no cover - h5py absent in the build image
        for k, v in data.items():
            f.create_dataset(k, data=v)
    return outfile
