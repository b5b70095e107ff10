"""Post-hoc helpers the callers of memento use next (reference: memento/util.py:16-29), statsmodels-free."""

import numpy as np


def _get_gene_idx(adata, gene_list):
    """Indices of the genes in ``gene_list`` (reference: memento/util.py:16-19)."""
    pos = {g: i for i, g in enumerate(adata.var.index)}
    return np.array([pos[g] for g in gene_list])


def _fdrcorrect(pvals):
    """Benjamini-Hochberg FDR over the non-NaN p-values; NaN entries get 1 (reference: memento/util.py:22-29,
    which calls statsmodels.stats.multitest.fdrcorrection)."""
    pvals = np.asarray(pvals, dtype=np.float64)
    fdr = np.ones(pvals.shape[0])
    ok = ~np.isnan(pvals)
    p = pvals[ok]
    n = p.shape[0]
    if n == 0:
        return fdr
    order = np.argsort(p)
    ranked = p[order] * n / np.arange(1, n + 1)
    ranked = np.minimum.accumulate(ranked[::-1])[::-1]
    out = np.empty(n)
    out[order] = np.minimum(ranked, 1.0)
    fdr[ok] = out
    return fdr
