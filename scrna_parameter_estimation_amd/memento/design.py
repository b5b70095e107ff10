"""Host side of K9: fold the per-gene meta-regression into one weight row per treatment column.

Everything ``_regress_1d`` / ``_regress_2d`` do before ``_compute_asl`` is LINEAR in the per-group
replicate vector (/root/reference/memento/hypothesis_test.py:262-271, :290-291, :218-228), so for a given
set of valid groups it is a fixed matrix ``W`` (T x n_groups):  coef[t, b] = sum_j W[t, j] * y[j, b].
The HIP kernel mm_contract_stats applies it; this module only builds W (tiny dense algebra, numpy).
"""

import numpy as np


def weight_rows(cov, trt, Nc, good):
    """W (T x n_groups, zero on groups that are not ``good``) for covariates ``cov`` (n x C), treatment
    ``trt`` (n x T) and cell-count weights ``Nc`` (n,).

    all-ones treatment  -> Nc-weighted average               (hypothesis_test.py:262-265)
    otherwise           -> residualise response and treatment on [1, cov] by weighted least squares
                           (what sklearn LinearRegression(sample_weight=Nc) predicts, :269-271), then the
                           weighted slope of _cross_coef (:218-228).
    """
    cov = np.asarray(cov, dtype=np.float64)
    trt = np.asarray(trt, dtype=np.float64)
    Nc = np.asarray(Nc, dtype=np.float64)
    good = np.asarray(good, dtype=bool)
    idx = np.flatnonzero(good)
    n, T = len(idx), trt.shape[1]
    W = np.zeros((T, len(good)))
    if n == 0:
        return W
    c, t, w = cov[idx], trt[idx], Nc[idx]
    wbar = w / w.sum()
    if (t == 1).mean() == 1:
        W[:, idx] = wbar[None, :]
        return W
    Xa = np.column_stack([np.ones(n), c])
    sw = np.sqrt(w)
    # hat matrix of the weighted fit: H = Xa (sw Xa)^+ sw
    H = Xa @ (np.linalg.pinv(Xa * sw[:, None]) * sw[None, :])
    M = np.eye(n) - H
    tt = M @ t                                  # residualised treatment (n x T)
    Ac = tt - wbar @ tt                         # weighted centring
    ss = wbar @ (Ac ** 2)                       # weighted sum of squares per treatment column
    center = np.eye(n) - np.outer(np.ones(n), wbar)
    with np.errstate(divide="ignore", invalid="ignore"):
        Wg = ((Ac * wbar[:, None]).T @ center @ M) / ss[:, None]
    W[:, idx] = Wg
    return W


def residual_parts(cov, trt, Nc, good):
    """For resample_rep=True (hypothesis_test.py:269-286): the residual maker M = I - H of the weighted fit on
    [1, cov] embedded in an n_groups x n_groups matrix (zero rows/columns on groups that are not ``good``), and
    the residualised treatment (T x n_groups, zero on bad groups)."""
    cov = np.asarray(cov, dtype=np.float64)
    trt = np.asarray(trt, dtype=np.float64)
    Nc = np.asarray(Nc, dtype=np.float64)
    good = np.asarray(good, dtype=bool)
    idx = np.flatnonzero(good)
    n = len(idx)
    ng = len(good)
    M = np.zeros((ng, ng))
    tt = np.zeros((trt.shape[1], ng))
    if n == 0:
        return M, tt
    Xa = np.column_stack([np.ones(n), cov[idx]])
    w = Nc[idx]
    sw = np.sqrt(w)
    H = Xa @ (np.linalg.pinv(Xa * sw[:, None]) * sw[None, :])
    Mg = np.eye(n) - H
    M[np.ix_(idx, idx)] = Mg
    # The residualised TREATMENT follows sklearn's own arithmetic (centre by the weighted means, min-norm lstsq on the
    # sqrt-weighted centred design, intercept from the offsets): groups with equal treatment then get bit-identical
    # residuals, which the degenerate resampled columns (every drawn group has the same treatment -> 0/0 -> NaN,
    # ignored by nanstd) rely on.
    c, t = cov[idx], trt[idx]
    c_off, t_off = np.average(c, axis=0, weights=w), np.average(t, axis=0, weights=w)
    coef, *_ = np.linalg.lstsq((c - c_off) * sw[:, None], (t - t_off) * sw[:, None], rcond=None)
    pred = c @ coef + (t_off - c_off @ coef)
    tt[:, idx] = (t - pred).T
    return M, tt
