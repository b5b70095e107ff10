"""Synthetic sparse UMI-count matrices of the shapes BASELINE.json names.

Host (numpy) generator for tests / fixtures / small benches.  Counts are gamma-Poisson:
per-gene base mean ``mu_g ~ LogNormal(-2.2, 1.2)``, per-cell depth ``d_c ~ LogNormal(0, 0.35)``,
``x_cg ~ Poisson(mu_g d_c gamma_cg)``, ``gamma ~ Gamma(2, 0.5)``; ``mu`` is rescaled by bisection
so the realised non-zero fraction hits ``density``.  This is input data only -- no reference code.
"""

import numpy as np
import pandas as pd
import scipy.sparse as sp

from .anndata_lite import AnnDataLite


def _expected_density(mu, scale, depth_nodes, depth_w):
    # P(x>0) for gamma(2, 0.5)-Poisson: 1 - (1 + 0.5*m)^-2, averaged over a depth quadrature
    m = scale * mu[None, :] * depth_nodes[:, None]
    return float((depth_w[:, None] * (1.0 - (1.0 + 0.5 * m) ** -2.0)).sum(axis=0).mean())


def synth_counts(n_cells, n_genes, density, seed, dtype=np.float32, row_block=8192):
    """Return a CSR ``n_cells x n_genes`` matrix of integer-valued counts with ~``density`` nnz."""
    rng = np.random.default_rng(seed)
    mu = rng.lognormal(-2.2, 1.2, size=n_genes)
    depth = rng.lognormal(0.0, 0.35, size=n_cells)
    # quadrature over depth for the density calibration
    qs = (np.arange(64) + 0.5) / 64
    nodes = np.quantile(depth, qs)
    w = np.full(64, 1.0 / 64)
    lo, hi = 1e-4, 1e4
    for _ in range(60):
        mid = np.sqrt(lo * hi)
        if _expected_density(mu, mid, nodes, w) < density:
            lo = mid
        else:
            hi = mid
    mu = mu * np.sqrt(lo * hi)
    blocks = []
    for r0 in range(0, n_cells, row_block):
        r1 = min(n_cells, r0 + row_block)
        lam = depth[r0:r1, None] * mu[None, :]
        lam = lam * rng.gamma(2.0, 0.5, size=lam.shape)
        x = rng.poisson(lam)
        blocks.append(sp.csr_matrix(x.astype(dtype)))
    X = sp.vstack(blocks, format="csr")
    X.sort_indices()
    return X


def synth_adata(n_cells, n_genes, density, n_cond, n_rep, seed, q=0.07, dtype=np.float32):
    """AnnDataLite with obs columns ``cond`` (0..n_cond-1), ``rep`` (0..n_rep-1) and ``q``."""
    X = synth_counts(n_cells, n_genes, density, seed, dtype=dtype)
    rng = np.random.default_rng(seed + 1)
    grp = rng.integers(0, n_cond * n_rep, size=n_cells)
    obs = pd.DataFrame(
        {
            "cond": (grp // n_rep).astype(np.int64),
            "rep": (grp % n_rep).astype(np.int64),
            "q": np.full(n_cells, q),
        },
        index=[f"c{i}" for i in range(n_cells)],
    )
    var = pd.DataFrame(index=[f"g{i}" for i in range(n_genes)])
    return AnnDataLite(X, obs, var)
