"""Minimal AnnData / MuData stand-ins.

The reference operates on ``anndata.AnnData`` and ``mudata.MuData`` objects
(/root/reference/muon/_atac/preproc.py:62-67, muon/_core/tools.py:425-433).
Neither package is installable in the build image, so the hot-path functions
accept *either* the real classes (when importable) *or* these small
duck-typed containers that expose exactly the slots the hot path reads and
writes: ``X``, ``layers``, ``obs``, ``var``, ``obsm``, ``varm``, ``uns``,
``shape``, ``copy()``, ``is_view`` and ``[...]`` views.

They are host-side plumbing only; no arithmetic lives here.
"""
from __future__ import annotations

import copy as _copy
from typing import Dict, Optional

import numpy as np
import pandas as pd
from scipy.sparse import issparse

try:  # pragma: no cover - not available in the build image
    from anndata import AnnData as _RealAnnData
except Exception:  # noqa: BLE001
    _RealAnnData = None
try:  # pragma: no cover
    from mudata import MuData as _RealMuData
except Exception:  # noqa: BLE001
    _RealMuData = None


def _as_index(names, n, prefix):
    if names is None:
        return pd.Index([f"{prefix}{i}" for i in range(n)], dtype=object)
    return pd.Index(names)


class AnnData:
    """Duck-typed subset of ``anndata.AnnData``."""

    def __init__(
        self,
        X=None,
        obs: Optional[pd.DataFrame] = None,
        var: Optional[pd.DataFrame] = None,
        layers: Optional[Dict] = None,
        obsm: Optional[Dict] = None,
        varm: Optional[Dict] = None,
        uns: Optional[Dict] = None,
        shape=None,
    ):
        if X is not None and not issparse(X):
            X = np.asarray(X)
        self._X = X
        if X is not None:
            shape = tuple(X.shape)
        elif shape is None:
            if obs is not None and var is not None:
                shape = (len(obs), len(var))
            else:
                raise ValueError("shape is required when X is None")
        self._shape = (int(shape[0]), int(shape[1]))
        n, d = self._shape
        if obs is None:
            obs = pd.DataFrame(index=_as_index(None, n, "obs"))
        if var is None:
            var = pd.DataFrame(index=_as_index(None, d, "var"))
        if len(obs) != n or len(var) != d:
            raise ValueError("obs/var lengths do not match X")
        self.obs = obs
        self.var = var
        self.layers = dict(layers) if layers else {}
        self.obsm = dict(obsm) if obsm else {}
        self.varm = dict(varm) if varm else {}
        self.uns = dict(uns) if uns else {}
        self.obsp = {}
        self._is_view = False
        self.raw = None

    # -- basic geometry ---------------------------------------------------
    @property
    def shape(self):
        return self._shape

    @property
    def n_obs(self):
        return self._shape[0]

    @property
    def n_vars(self):
        return self._shape[1]

    @property
    def obs_names(self):
        return self.obs.index

    @obs_names.setter
    def obs_names(self, names):
        self.obs.index = pd.Index(names)

    @property
    def var_names(self):
        return self.var.index

    @var_names.setter
    def var_names(self, names):
        self.var.index = pd.Index(names)

    @property
    def is_view(self):
        return self._is_view

    @property
    def X(self):
        return self._X

    @X.setter
    def X(self, value):
        if value is not None:
            if not issparse(value):
                value = np.asarray(value)
            if tuple(value.shape) != self._shape:
                raise ValueError(
                    f"X has shape {tuple(value.shape)}, expected {self._shape}"
                )
        self._X = value

    # -- copy / views -------------------------------------------------------
    def copy(self):
        new = AnnData(
            None if self._X is None else self._X.copy(),
            obs=self.obs.copy(),
            var=self.var.copy(),
            layers={k: v.copy() for k, v in self.layers.items()},
            obsm={k: _copy.copy(v) for k, v in self.obsm.items()},
            varm={k: _copy.copy(v) for k, v in self.varm.items()},
            uns=_copy.deepcopy(self.uns),
            shape=self._shape,
        )
        new.obsp = {k: v.copy() for k, v in self.obsp.items()}
        return new

    @staticmethod
    def _norm_index(idx, names, n):
        if isinstance(idx, slice):
            return np.arange(n)[idx]
        arr = np.asarray(idx)
        if arr.dtype == bool:
            return np.nonzero(arr)[0]
        if arr.dtype.kind in "iu":
            return arr.reshape(-1)
        # label based
        return names.get_indexer(pd.Index(arr.reshape(-1)))

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key, slice(None))
        ri = self._norm_index(key[0], self.obs.index, self.n_obs)
        ci = self._norm_index(key[1], self.var.index, self.n_vars)

        def sub2(m):
            if m is None:
                return None
            if issparse(m):
                return m.tocsr()[ri][:, ci]
            return np.asarray(m)[np.ix_(ri, ci)]

        view = AnnData(
            sub2(self._X),
            obs=self.obs.iloc[ri].copy(),
            var=self.var.iloc[ci].copy(),
            layers={k: sub2(v) for k, v in self.layers.items()},
            obsm={k: np.asarray(v)[ri] for k, v in self.obsm.items()},
            varm={k: np.asarray(v)[ci] for k, v in self.varm.items()},
            uns=self.uns,
            shape=(len(ri), len(ci)),
        )
        view._is_view = True
        return view

    def _init_as_actual(self):
        """Materialise a view in place (what scanpy's view_to_actual does)."""
        self._is_view = False
        self.uns = _copy.deepcopy(self.uns)

    def __repr__(self):
        return f"AnnData(duck) n_obs × n_vars = {self.n_obs} × {self.n_vars}"


class MuData:
    """Duck-typed subset of ``mudata.MuData``: a dict of modalities sharing obs."""

    def __init__(self, mod: Dict[str, AnnData]):
        self.mod = dict(mod)
        self.obsm = {}
        self.varm = {}
        self.uns = {}
        self.obsp = {}
        self.update()

    def update(self):
        obs_index = None
        for a in self.mod.values():
            obs_index = a.obs.index if obs_index is None else obs_index.union(a.obs.index, sort=False)
        var_index = None
        for a in self.mod.values():
            var_index = a.var.index if var_index is None else var_index.append(a.var.index)
        old_obs = getattr(self, "obs", None)
        self.obs = pd.DataFrame(index=obs_index)
        if old_obs is not None:
            for c in old_obs.columns:
                self.obs[c] = old_obs[c].reindex(obs_index)
        # propagate per-modality obs columns that are shared by name
        self.var = pd.DataFrame(index=var_index)
        for c in set().union(*[set(a.var.columns) for a in self.mod.values()]) if self.mod else []:
            vals = []
            for a in self.mod.values():
                if c in a.var.columns:
                    vals.append(a.var[c])
                else:
                    vals.append(pd.Series([np.nan] * a.n_vars, index=a.var.index))
            self.var[c] = pd.concat(vals)

    def update_obs(self):
        """mudata's MuData.update_obs(): bring .obs in line with the modalities (muon.pp.neighbors calls it last,
        /root/reference/muon/_core/preproc.py:638); global columns are kept."""
        self.update()

    @property
    def shape(self):
        return (len(self.obs), len(self.var))

    @property
    def n_obs(self):
        return len(self.obs)

    @property
    def n_vars(self):
        return len(self.var)

    @property
    def obs_names(self):
        return self.obs.index

    @property
    def var_names(self):
        return self.var.index

    def __getitem__(self, key):
        if isinstance(key, str) and key in self.mod:
            return self.mod[key]
        names = pd.Index(np.asarray(key).reshape(-1))
        sub = {}
        for m, a in self.mod.items():
            keep = names[names.isin(a.obs.index)]
            v = a[keep.values]
            v._init_as_actual()
            sub[m] = v
        out = MuData(sub)
        out.obs = self.obs.loc[out.obs.index].copy()
        return out

    def copy(self):
        new = MuData({k: v.copy() for k, v in self.mod.items()})
        new.obs = self.obs.copy()
        new.var = self.var.copy()
        new.obsm = {k: _copy.copy(v) for k, v in self.obsm.items()}
        new.varm = {k: _copy.copy(v) for k, v in self.varm.items()}
        new.uns = _copy.deepcopy(self.uns)
        new.obsp = {k: v.copy() for k, v in self.obsp.items()}
        return new

    def __repr__(self):
        mods = ", ".join(f"{k}: {v.n_obs}×{v.n_vars}" for k, v in self.mod.items())
        return f"MuData(duck) n_obs={self.n_obs} [{mods}]"


def is_anndata(obj) -> bool:
    if isinstance(obj, AnnData):
        return True
    return _RealAnnData is not None and isinstance(obj, _RealAnnData)


def is_mudata(obj) -> bool:
    if isinstance(obj, MuData):
        return True
    return _RealMuData is not None and isinstance(obj, _RealMuData)


def view_to_actual(adata) -> None:
    """scanpy._utils.view_to_actual equivalent (reference preproc.py:84)."""
    if getattr(adata, "is_view", False):
        if isinstance(adata, AnnData):
            adata._init_as_actual()
        else:  # real anndata
            adata._init_as_actual(adata.copy())
