"""Minimal AnnData stand-in.

`anndata` is not installed in the build image, and the memento API only touches a
handful of AnnData attributes (counted over /root/reference/memento/*.py):
``X, obs, var, uns, shape, copy(), _inplace_subset_var(mask)``.  A real
``anndata.AnnData`` works with this package unchanged (duck typing); this class
exists so tests, the bench and users without anndata have something to pass in.
"""

import copy as _copy

import numpy as np
import pandas as pd
import scipy.sparse as sp


class AnnDataLite:
    """Cells x genes container: ``X`` (scipy CSR), ``obs``/``var`` DataFrames, ``uns`` dict."""

    def __init__(self, X, obs=None, var=None, uns=None):
        if not sp.isspmatrix_csr(X):
            raise TypeError("AnnDataLite.X must be a scipy.sparse.csr_matrix")
        n, g = X.shape
        self.X = X
        self.obs = obs if obs is not None else pd.DataFrame(index=[str(i) for i in range(n)])
        self.var = var if var is not None else pd.DataFrame(index=[f"gene{i}" for i in range(g)])
        if len(self.obs) != n or len(self.var) != g:
            raise ValueError("obs/var length does not match X")
        self.uns = uns if uns is not None else {}
        self.device_csr = None      # engine.DeviceCSR already resident in HBM (h5ad.read_h5ad): setup_memento takes it instead of X

    @property
    def shape(self):
        return self.X.shape

    @property
    def n_obs(self):
        return self.X.shape[0]

    @property
    def n_vars(self):
        return self.X.shape[1]

    @property
    def var_names(self):
        return self.var.index

    @property
    def obs_names(self):
        return self.obs.index

    def copy(self):
        return AnnDataLite(self.X.copy(), self.obs.copy(), self.var.copy(), _copy.deepcopy(self.uns))

    def _inplace_subset_var(self, mask):
        mask = np.asarray(mask)
        self.X = self.X[:, mask]
        if not sp.isspmatrix_csr(self.X):
            self.X = sp.csr_matrix(self.X)
        self.var = self.var.iloc[np.flatnonzero(mask) if mask.dtype == bool else mask].copy()

    def __repr__(self):
        return f"AnnDataLite(n_obs={self.shape[0]}, n_vars={self.shape[1]}, uns={list(self.uns)})"
