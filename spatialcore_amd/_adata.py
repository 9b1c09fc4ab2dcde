"""Minimal in-memory AnnData look-alike.

The hot path only touches a handful of AnnData attributes:
``X, layers, obs, obsm, obsp, uns, var_names, n_obs, copy()`` and column slicing
``adata[:, names]``. ``anndata`` is not installed in the build image nor on the GPU box, so
tests, ``bench.py`` and users without ``anndata`` use this class; every public function in
``spatialcore_amd.spatial`` duck-types and works with a real ``anndata.AnnData`` as well.
"""

from __future__ import annotations

import copy as _copy
from typing import Any, Dict, Optional, Sequence

import numpy as np
import pandas as pd
from scipy import sparse


class SimpleAnnData:
    """Cells x genes container with the AnnData attribute names the hot path reads/writes."""

    def __init__(
        self,
        X,
        obs: Optional[pd.DataFrame] = None,
        var_names: Optional[Sequence[str]] = None,
        var: Optional[pd.DataFrame] = None,
        obsm: Optional[Dict[str, Any]] = None,
        obsp: Optional[Dict[str, Any]] = None,
        layers: Optional[Dict[str, Any]] = None,
        uns: Optional[Dict[str, Any]] = None,
    ) -> None:
        if not sparse.issparse(X):
            X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError(f"X must be 2-D (cells x genes), got shape {X.shape}")
        self.X = X
        n_obs, n_vars = X.shape
        if var_names is None:
            var_names = [f"gene_{i}" for i in range(n_vars)]
        self.var_names = pd.Index([str(v) for v in var_names])
        if len(self.var_names) != n_vars:
            raise ValueError("len(var_names) must equal X.shape[1]")
        self.var = pd.DataFrame(index=self.var_names) if var is None else var
        if len(self.var) != n_vars:
            raise ValueError("len(var) must equal X.shape[1]")
        if obs is None:
            obs = pd.DataFrame(index=pd.RangeIndex(n_obs).astype(str))
        if len(obs) != n_obs:
            raise ValueError("len(obs) must equal X.shape[0]")
        self.obs = obs
        self.obsm = dict(obsm or {})
        self.obsp = dict(obsp or {})
        self.layers = dict(layers or {})
        self.uns = dict(uns or {})

    # -- shape --------------------------------------------------------------------------------
    @property
    def n_obs(self) -> int:
        return self.X.shape[0]

    @property
    def n_vars(self) -> int:
        return self.X.shape[1]

    @property
    def shape(self):
        return self.X.shape

    @property
    def obs_names(self) -> pd.Index:
        return self.obs.index

    # -- copy / slicing -----------------------------------------------------------------------
    def copy(self) -> "SimpleAnnData":
        return SimpleAnnData(
            self.X.copy(),
            obs=self.obs.copy(),
            var_names=list(self.var_names),
            var=self.var.copy(),
            obsm={k: _copy.deepcopy(v) for k, v in self.obsm.items()},
            obsp={k: v.copy() for k, v in self.obsp.items()},
            layers={k: v.copy() for k, v in self.layers.items()},
            uns=_copy.deepcopy(self.uns),
        )

    def __getitem__(self, key) -> "SimpleAnnData":
        """Only the ``adata[:, gene_names]`` form used by the reference (AC:573) is supported."""
        if not (isinstance(key, tuple) and len(key) == 2 and key[0] == slice(None)):
            raise NotImplementedError("SimpleAnnData supports only adata[:, names] slicing")
        names = key[1]
        if isinstance(names, str):
            names = [names]
        cols = np.array([self.var_names.get_loc(n) for n in names], dtype=np.intp)
        X = self.X.tocsc()[:, cols].tocsr() if sparse.issparse(self.X) else self.X[:, cols]
        layers = {}
        for k, v in self.layers.items():
            layers[k] = v.tocsc()[:, cols].tocsr() if sparse.issparse(v) else np.asarray(v)[:, cols]
        return SimpleAnnData(
            X,
            obs=self.obs,
            var_names=[self.var_names[c] for c in cols],
            var=self.var.iloc[cols],
            obsm=self.obsm,
            obsp=self.obsp,
            layers=layers,
            uns=self.uns,
        )

    def __repr__(self) -> str:
        return f"SimpleAnnData object with n_obs x n_vars = {self.n_obs} x {self.n_vars}"
