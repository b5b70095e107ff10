"""CPU ORACLE for the memento hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The shipped path (``scrna_parameter_estimation_amd``) never does: it calls the HIP library.

What it is: an array-level numpy restatement of the reference algorithm
(atarashansky/scrna-parameter-estimation, ``memento`` 0.0.9), each function citing the reference
file:line it follows.  Third-party arithmetic on the path is taken from the same libraries the
reference calls (numpy ``Generator(PCG64).multinomial``, ``np.polyfit``, scipy.stats), all of which
are present both in the build container and on the GPU box.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz``, which ``tests/golden/make_golden.py`` produced by importing and running the
real reference from /root/reference in the build container (float64 X, seeded ``np.random``,
``num_cpus=1``).  The reference has no tests or golden vectors of its own.
"""

import warnings

import numpy as np
import scipy.sparse as sp
import scipy.stats as stats

# ----------------------------------------------------------------------------------------------
# size factors and their binning
# ----------------------------------------------------------------------------------------------


def poly_mv_fit(mean, var):
    """Quadratic fit of log(var) on log(mean) over entries with both > 0 (estimator.py:84-93)."""
    ok = (mean > 0) & (var > 0)
    return np.polyfit(np.log(mean[ok]), np.log(var[ok]), 2)


def residual_variance(mean, var, fit):
    """exp(log v - poly(log m)); NaN where m<=0 or v<=0 (estimator.py:103-111)."""
    ok = (mean > 0) & (var > 0)
    out = np.full(np.shape(mean), np.nan)
    lm = np.log(mean[ok])
    pred = np.zeros_like(lm)
    for c in fit:  # Horner, same order as np.poly1d.__call__
        pred = pred * lm + c
    out[ok] = np.exp(np.log(var[ok]) - pred)
    return out


def moments_1d_sparse(X, size_factor, q):
    """Hypergeometric mean / variance from a sparse cells x genes block (estimator.py:177-185).

    mean_g = sum_c x/sf / n ;  var_g = sum x^2/sf^2 / n - (1-q) sum x/sf^2 / n - mean^2.
    """
    X = sp.csc_matrix(X, dtype=np.float64)
    n = X.shape[0]
    w = 1.0 / size_factor
    w2 = 1.0 / size_factor ** 2
    m1 = np.asarray(X.T.dot(w)).ravel() / n
    m2 = np.asarray(X.power(2).T.dot(w2)).ravel() / n - (1 - q) * np.asarray(X.T.dot(w2)).ravel() / n
    return m1, m2 - m1 ** 2


def setup_size_factors(X, q_all, filter_mean_thresh=0.07, trim_percent=0.1, shrinkage=0.5):
    """The setup_memento pipeline (main.py:54-91, estimator.py:49-81).

    Returns (size_factor, least_variable_mask, all_mean, all_var).
    """
    X = sp.csr_matrix(X, dtype=np.float64)
    n = X.shape[0]
    naive = np.asarray(X.sum(axis=1)).ravel()                      # estimator.py:65-69 (total=True)
    m, v = moments_1d_sparse(X, naive, q_all)                      # main.py:62-66
    m = m.copy()
    m[np.asarray(X.mean(axis=0)).ravel() < filter_mean_thresh] = 0  # main.py:67
    rv = residual_variance(m, v, poly_mv_fit(m, v))                # main.py:68
    ulim = np.quantile(rv[np.isfinite(rv)], trim_percent)          # main.py:71
    rv[~np.isfinite(rv)] = np.inf
    mask = rv < ulim                                               # main.py:73
    nrc = np.asarray(X.multiply(mask).sum(axis=1)).ravel()         # estimator.py:73
    nrc = nrc + np.quantile(nrc, shrinkage)                        # estimator.py:74
    sf = nrc / nrc.mean()                                          # estimator.py:75-76
    am, av = moments_1d_sparse(X, sf, q_all)                       # main.py:86-91
    return sf, mask, am, av


def bin_size_factor(size_factor, num_bins=30):
    """Equal-width binning of size factors; each cell gets its bin's mean, the max keeps its value
    (main.py:138-147).  Returns (approx_sf, bin_index in 0..num_bins-1, bin_means)."""
    means, _, idx = stats.binned_statistic(size_factor, size_factor, bins=num_bins, statistic="mean")
    idx = np.clip(idx, 1, means.shape[0])
    approx = means[idx - 1]
    approx[size_factor == size_factor.max()] = size_factor.max()
    return approx, idx - 1, means


# ----------------------------------------------------------------------------------------------
# compute_1d_moments
# ----------------------------------------------------------------------------------------------


def compute_1d_moments(X, group_id, n_groups, size_factor, group_q, filter_mean_thresh=0.07, min_perc_group=0.7):
    """Per-group moments, gene filters, pooled mean-variance fit, residual variance (main.py:171-255).

    ``group_id`` is a per-cell int array (order of groups = reference's first-appearance order).
    Returns dict with mean/var/res_var (n_groups x G_kept), gene_filter, gene_rv_filter (n_groups x G /
    n_groups x G_kept), overall mask and the fit.
    """
    X = sp.csr_matrix(X, dtype=np.float64)
    G = X.shape[1]
    mean = np.zeros((n_groups, G))
    var = np.zeros((n_groups, G))
    gf = np.zeros((n_groups, G), dtype=bool)
    rvf = np.zeros((n_groups, G), dtype=bool)
    for g in range(n_groups):
        sel = np.flatnonzero(group_id == g)
        Xg = X[sel]
        mean[g], var[g] = moments_1d_sparse(Xg, size_factor[sel], group_q[g])   # main.py:190-194
        obs_mean = np.asarray(Xg.mean(axis=0)).ravel()                          # main.py:201
        gf[g] = (obs_mean > filter_mean_thresh) & (var[g] > 0)                  # main.py:202-203
        rvf[g] = np.asarray(Xg.max(axis=0).todense()).ravel() >= 2              # main.py:206-207
    overall = gf.mean(axis=0) > min_perc_group                                  # main.py:210-212
    mean, var, rvf_k = mean[:, overall], var[:, overall], rvf[:, overall]
    fit = poly_mv_fit(np.concatenate([mean[g][rvf_k[g]] for g in range(n_groups)]),
                      np.concatenate([var[g][rvf_k[g]] for g in range(n_groups)]))  # main.py:232-245
    res_var = np.stack([residual_variance(mean[g], var[g], fit) for g in range(n_groups)])  # main.py:248-255
    return dict(mean=mean, var=var, res_var=res_var, gene_filter=gf, gene_rv_filter=rvf_k,
                overall_gene_filter=overall, mv_fit=fit)


# ----------------------------------------------------------------------------------------------
# unique-value bootstrap (1D)
# ----------------------------------------------------------------------------------------------


def unique_bins_1d(values, approx_sf, r, r0):
    """Collapse one gene's cells in one group into unique (count, approx_sf) bins (bootstrap.py:62-71).

    ``values``: dense per-cell counts (N_g,), ``approx_sf``: per-cell binned size factor, ``r``/``r0``:
    the two uniforms the reference draws from the global ``np.random`` stream.  Bin order is ascending
    ``count*r + r0*approx_sf`` exactly as ``np.unique`` returns it.
    Returns (inv_sf, inv_sf_sq, expr, mult) each of length K.
    """
    code = values * r
    code = code + r0 * approx_sf
    _, first, mult = np.unique(code, return_index=True, return_counts=True)
    sf = approx_sf[first]
    return 1.0 / sf, 1.0 / sf ** 2, values[first].astype(np.float64), mult


def multinomial_weights(n_obs, mult, num_boot):
    """K x B int64 bin weights: Generator(PCG64(5)).multinomial, re-seeded per call (bootstrap.py:102-103)."""
    gen = np.random.Generator(np.random.PCG64(5))
    return gen.multinomial(n_obs, mult / mult.sum(), size=num_boot).T


def replicate_moments_1d(expr, inv_sf, inv_sf_sq, weights, n_obs, q):
    """Replicate mean/var from bins and weights -- the tuple branch of the estimator
    (estimator.py:171-174, 182-183), same operation order."""
    e = expr.reshape(-1, 1)
    a = inv_sf.reshape(-1, 1)
    b = inv_sf_sq.reshape(-1, 1)
    m1 = (e * weights * a).sum(axis=0) / n_obs
    m2 = (e ** 2 * weights * b - (1 - q) * e * weights * b).sum(axis=0) / n_obs
    return m1, m2 - m1 ** 2


def bootstrap_1d(values, approx_sf, q, num_boot, r, r0):
    """_bootstrap_1d (bootstrap.py:74-116) for dense per-cell ``values``; all-NaN if K <= 1."""
    inv_sf, inv_sf_sq, expr, mult = unique_bins_1d(values, approx_sf, r, r0)
    if expr.shape[0] <= 1:
        return np.full(num_boot, np.nan), np.full(num_boot, np.nan)
    w = multinomial_weights(values.shape[0], mult, num_boot)
    return replicate_moments_1d(expr, inv_sf, inv_sf_sq, w, values.shape[0], q)


# ----------------------------------------------------------------------------------------------
# hypothesis test: fill, regression, achieved significance level
# ----------------------------------------------------------------------------------------------


def fill_invalid(val):
    """Replace <=0 / NaN entries by random draws (global np.random) from the valid ones; None if no
    valid entry (hypothesis_test.py:23-33)."""
    bad = np.isnan(val)
    bad[~bad] = val[~bad] <= 0
    nbad = int(bad.sum())
    if nbad == val.shape[0]:
        return None
    val = val.copy()
    val[bad] = np.random.choice(val[~bad], nbad)
    return val


def fill_invalid_corr(val):
    """NaN entries replaced by random valid ones (hypothesis_test.py:35-40)."""
    bad = np.isnan(val)
    val = val.copy()
    val[bad] = np.random.choice(val[~bad], int(bad.sum()))
    return val


def _weighted_residualize(Z, cov, w):
    """Z minus its weighted least-squares projection on [1, cov] -- what
    ``Z - LinearRegression().fit(cov, Z, w).predict(cov)`` returns (hypothesis_test.py:269-271)."""
    Xa = np.column_stack([np.ones(cov.shape[0]), cov])
    sw = np.sqrt(w)[:, None]
    beta, *_ = np.linalg.lstsq(Xa * sw, Z * sw, rcond=None)
    return Z - Xa @ beta


def cross_coef(A, B, w):
    """Weighted regression slope of every column of B on every column of A (hypothesis_test.py:218-228)."""
    Ac = A - np.average(A, axis=0, weights=w)
    Bc = B - np.average(B, axis=0, weights=w)
    ss = np.average(Ac ** 2, axis=0, weights=w)
    return (Ac.T * w) @ Bc / w.sum() / ss[:, None]


def compute_asl(perm_diff, resampling="bootstrap", approx=False):
    """Achieved significance level incl. the extreme-value tail fit (hypothesis_test.py:57-141)."""
    if np.all(perm_diff == perm_diff.mean()):
        return np.nan
    null = perm_diff[1:] - perm_diff[0] if resampling == "bootstrap" else perm_diff[1:]
    null = null[np.isfinite(null)]
    stat = perm_diff[0]
    if approx:
        mu, sd = stats.norm.fit(null)
        a = np.abs(stat)
        return stats.norm.sf(a, mu, sd) + stats.norm.cdf(-a, mu, sd)
    a = abs(stat)
    extreme = int((null > a).sum() + (null < -a).sum())
    fallback = (extreme + 1) / (null.shape[0] + 1)
    if extreme > 10:
        return fallback
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            srt = np.sort(null)
            left = None
            for n_exec in range(300, 50, -30):
                tail = srt[:n_exec]
                params = stats.genextreme.fit(tail)
                if stats.kstest(tail, "genextreme", args=params)[1] > 0.05:
                    left = (n_exec / srt.shape[0]) * stats.genextreme.cdf(-a, *params)
                    break
            if left is None:
                return fallback
            for n_exec in range(300, 50, -30):
                tail = srt[-n_exec:]
                params = stats.genextreme.fit(tail)
                if stats.kstest(tail, "genextreme", args=params)[1] > 0.05:
                    return (n_exec / srt.shape[0]) * stats.genextreme.sf(a, *params) + left
            return fallback
        except Exception:
            return fallback


def cross_coef_resampled(A, B, w, drop_degenerate=False):
    """Per-column weighted slope with per-column rows/weights (hypothesis_test.py:231-239).
    A: (n, nb, T) residualised treatment of the drawn groups, B: (n, nb), w: (n, nb).
    ``drop_degenerate``: NaN for columns whose drawn groups all share one treatment value -- there the reference's
    value is 0/0 or a ratio of round-off residues (noise); the HIP kernel always reports NaN for them."""
    Bc = B - np.average(B, axis=0, weights=w)
    Ac = A - (A * w[:, :, None]).sum(axis=0) / w.sum(axis=0)[:, None]
    ss = (Ac ** 2 * w[:, :, None]).sum(axis=0) / w.sum(axis=0)[:, None]
    with np.errstate(invalid="ignore", divide="ignore"):
        out = np.einsum("ijk,ij->jk", Ac * w[:, :, None], Bc).T / w.sum(axis=0) / ss.T
    if drop_degenerate:
        out[(ss <= 1e-24 * np.abs(A).max(axis=0) ** 2).T] = np.nan
    return out


def _sklearn_like_residualize(Z, cov, w):
    """Z - LinearRegression().fit(cov, Z, w).predict(cov), following sklearn's own steps (weighted centring, min-norm
    lstsq on the sqrt-weighted centred design) so equal inputs give bit-identical residuals."""
    c_off, z_off = np.average(cov, axis=0, weights=w), np.average(Z, axis=0, weights=w)
    sw = np.sqrt(w)[:, None]
    coef, *_ = np.linalg.lstsq((cov - c_off) * sw, (Z - z_off) * sw, rcond=None)
    return Z - (cov @ coef + (z_off - c_off @ coef))


def regress_1d(cov, trt, boot_mean, boot_var, Nc, resampling="bootstrap", approx=False, resample_rep=False,
               drop_degenerate=False):
    """_regress_1d (hypothesis_test.py:242-300); ``resample_rep`` draws from the global np.random stream like the
    reference (:273-286)."""
    ok = np.all(np.isfinite(boot_mean), axis=0) & np.all(np.isfinite(boot_var), axis=0)
    bm, bv = boot_mean[:, ok], boot_var[:, ok]
    Nc = np.asarray(Nc, dtype=np.float64)
    if (trt == 1).mean() == 1:
        mc = np.average(bm, axis=0, weights=Nc).reshape(1, -1)
        vc = np.average(bv, axis=0, weights=Nc).reshape(1, -1)
    elif resample_rep:
        n, nb = bm.shape[0], bm.shape[1] - 1
        bmt, bvt = _sklearn_like_residualize(bm, cov, Nc), _sklearn_like_residualize(bv, cov, Nc)
        tt = _sklearn_like_residualize(trt, cov, Nc)
        ra = np.random.choice(n, size=(n, nb))
        ra[:, 0] = np.arange(n)
        ba = np.random.choice(nb, (n, nb)) + 1
        ba[:, 0] = 0
        mc = cross_coef_resampled(tt[ra], bmt[(ra, ba)], Nc[ra], drop_degenerate)
        vc = cross_coef_resampled(tt[ra], bvt[(ra, ba)], Nc[ra], drop_degenerate)
    else:
        tt = _weighted_residualize(trt, cov, Nc)
        mc = cross_coef(tt, _weighted_residualize(bm, cov, Nc), Nc)
        vc = cross_coef(tt, _weighted_residualize(bv, cov, Nc), Nc)
    masl = np.array([compute_asl(row, resampling, approx) for row in mc])
    vasl = np.array([compute_asl(row, resampling, approx) for row in vc])
    return mc[:, 0], np.nanstd(mc[:, 1:], axis=1), masl, vc[:, 0], np.nanstd(vc[:, 1:], axis=1), vasl


def ht_1d_gene(true_mean, true_res_var, cols, approx_sf, cov, trt, Nc, num_boot, mv_fit, q, **kw):
    """_ht_1d for one gene (hypothesis_test.py:144-215).  ``cols``: list over groups of dense per-cell
    count vectors.  Consumes the global ``np.random`` stream exactly like the reference."""
    ng = trt.shape[0]
    good = np.zeros(ng, dtype=bool)
    bm = np.full((ng, num_boot + 1), np.nan)
    bv = np.full((ng, num_boot + 1), np.nan)
    for j in range(len(true_mean)):
        if np.isnan(true_mean[j]) or np.isnan(true_res_var[j]) or true_mean[j] == 0 or true_res_var[j] < 0:
            continue
        bm[j, 0], bv[j, 0] = np.log(true_mean[j]), np.log(true_res_var[j])
        r = np.random.random(1)[0]
        r0 = np.random.random()
        mean, var = bootstrap_1d(cols[j], approx_sf[j], q[j], num_boot, r, r0)
        rv = residual_variance(mean, var, mv_fit)
        fm, fv = fill_invalid(mean), fill_invalid(rv)
        if fm is None or fv is None:
            continue
        bm[j, 1:], bv[j, 1:] = np.log(fm), np.log(fv)
        good[j] = True
    if good.sum() == 0:
        return (np.nan,) * 6
    return regress_1d(cov[good], trt[good], bm[good], bv[good], np.asarray(Nc)[good], **kw)


def ht_1d(X, group_id, n_groups, approx_sf, moments, cov, trt, num_boot, group_q, **kw):
    """ht_1d_moments over all kept genes, gene-major x treatment flat outputs (main.py:341-412).
    ``X`` must already be subset to the kept genes."""
    X = sp.csc_matrix(X, dtype=np.float64)
    sel = [np.flatnonzero(group_id == g) for g in range(n_groups)]
    Nc = np.array([len(s) for s in sel], dtype=np.float64)
    asf = [approx_sf[s] for s in sel]
    outs = [[] for _ in range(6)]
    for gi in range(X.shape[1]):
        col = np.asarray(X[:, gi].todense()).ravel()
        res = ht_1d_gene(moments["mean"][:, gi], moments["res_var"][:, gi], [col[s] for s in sel], asf,
                         cov, trt, Nc, num_boot, moments["mv_fit"], group_q, **kw)
        for o, r in zip(outs, res):
            o.append(np.atleast_1d(r) * np.ones(trt.shape[1]))
    return [np.concatenate(o) for o in outs]


# ----------------------------------------------------------------------------------------------
# 2D: covariance / correlation, bootstrap over (x_i, x_j, sf) bins
# ----------------------------------------------------------------------------------------------


def cov_2d_sparse(X, size_factor, q, idx1, idx2):
    """Hypergeometric covariance for gene pairs from a sparse block (estimator.py:220-233)."""
    X = sp.csc_matrix(X, dtype=np.float64)
    n = X.shape[0]
    w = 1.0 / size_factor
    A = X[:, idx1].multiply(w[:, None]).tocsc()
    Bm = X[:, idx2].multiply(w[:, None]).tocsc()
    prod = np.asarray(A.multiply(Bm).sum(axis=0)).ravel() / n
    same = np.asarray(idx1) == np.asarray(idx2)
    if same.any():
        s3 = np.asarray(X[:, np.asarray(idx1)[same]].T.dot(w ** 2)).ravel() / n
        prod[same] = prod[same] - (1 - q) * s3
    return prod - np.asarray(A.mean(axis=0)).ravel() * np.asarray(Bm.mean(axis=0)).ravel()


def corr_from_cov(cov, var1, var2):
    """cov / sqrt(v1 v2) clipped to [-1, 1]; 5.0-sentinel -> stays 5 -> clipped to 1 where a variance is
    <= 0?  No: the reference leaves 5.0 then clips to 1 (estimator.py:281-292) -- reproduced as is."""
    corr = np.full(cov.shape, 5.0)
    v1 = np.where(var1 <= 0, np.nan, var1)
    v2 = np.where(var2 <= 0, np.nan, var2)
    vp = np.sqrt(v1 * v2)
    ok = np.isfinite(vp)
    corr[ok] = cov[ok] / vp[ok]
    return np.clip(corr, -1, 1)


def corr_matrix(X, size_factor, q, var):
    """All-by-all correlation matrix (estimator.py:236-270)."""
    X = sp.csc_matrix(X, dtype=np.float64)
    n = X.shape[0]
    w = 1.0 / size_factor
    Xw = sp.csr_matrix(X.multiply(w[:, None]))
    prod = np.asarray((Xw.T @ Xw).todense()) / n
    d = np.arange(X.shape[1])
    prod[d, d] -= (1 - q) * np.asarray(X.T.dot(w ** 2)).ravel() / n
    mu = np.asarray(Xw.mean(axis=0)).ravel()
    cov = prod - np.outer(mu, mu)
    # estimator.py:259-263: the NaN assignments go to fancy-index COPIES (var_1, var_2); var_prod is built from the
    # untouched ``var``, so two negative variances give a finite product, a negative x positive one sqrt(<0) = NaN
    with np.errstate(invalid="ignore"):
        vp = np.sqrt(np.outer(var, var))
    corr = np.full(cov.shape, 5.0)
    ok = np.isfinite(vp)
    with np.errstate(invalid="ignore", divide="ignore"):
        corr[ok] = cov[ok] / vp[ok]
    inside = (corr < 1.05) & (corr > -1.05)
    corr[inside] = np.clip(corr[inside], -1, 1)
    corr[(corr > 1) | (corr < -1)] = np.nan
    return corr


def unique_bins_2d(v1, v2, approx_sf, r, r0):
    """2-column version of unique_bins_1d: code = v1*r[0] + v2*r[1] + r0*sf (bootstrap.py:62-71)."""
    code = v1 * r[0] + v2 * r[1]
    code = code + r0 * approx_sf
    _, first, mult = np.unique(code, return_index=True, return_counts=True)
    sf = approx_sf[first]
    return 1.0 / sf, 1.0 / sf ** 2, v1[first].astype(np.float64), v2[first].astype(np.float64), mult


def bootstrap_2d(v1, v2, approx_sf, q, num_boot, r, r0):
    """_bootstrap_2d (bootstrap.py:119-157): replicate cov and the two variances."""
    n = v1.shape[0]
    a, b, e1, e2, mult = unique_bins_2d(v1, v2, approx_sf, r, r0)
    w = multinomial_weights(n, mult, num_boot)
    A, Bq = a.reshape(-1, 1), b.reshape(-1, 1)
    E1, E2 = e1.reshape(-1, 1), e2.reshape(-1, 1)
    m1 = (E1 * w * A).sum(axis=0) / n
    m2 = (E2 * w * A).sum(axis=0) / n
    mx = (E1 * E2 * w * Bq).sum(axis=0) / n
    cov = mx - m1 * m2                                                   # estimator.py:214-218
    _, var1 = replicate_moments_1d(e1, a, b, w, n, q)
    _, var2 = replicate_moments_1d(e2, a, b, w, n, q)
    return cov, var1, var2


def regress_2d(cov_, trt, boot_corr, Nc, resampling="bootstrap", approx=False, resample_rep=False, drop_degenerate=False):
    """_regress_2d (hypothesis_test.py:367-414); ``resample_rep`` draws from the global np.random stream like the
    reference (:393-404)."""
    ok = np.all(np.isfinite(boot_corr), axis=0)
    bc = boot_corr[:, ok]
    Nc = np.asarray(Nc, dtype=np.float64)
    if (trt == 1).mean() == 1:
        cc = np.average(bc, axis=0, weights=Nc).reshape(1, -1)
    elif resample_rep:
        n, nb = bc.shape[0], bc.shape[1] - 1
        bct = _sklearn_like_residualize(bc, cov_, Nc)
        tt = _sklearn_like_residualize(trt, cov_, Nc)
        ra = np.random.choice(n, size=(n, nb))
        ra[:, 0] = np.arange(n)
        ba = np.random.choice(nb, (n, nb)) + 1
        ba[:, 0] = 0
        cc = cross_coef_resampled(tt[ra], bct[(ra, ba)], Nc[ra], drop_degenerate)
    else:
        cc = cross_coef(_weighted_residualize(trt, cov_, Nc), _weighted_residualize(bc, cov_, Nc), Nc)
    asl = np.array([compute_asl(row, resampling, approx) for row in cc])
    return cc[:, 0], np.nanstd(cc[:, 1:], axis=1), asl


def ht_2d_pair(true_corr, cols1, cols2, approx_sf, cov_, trt, Nc, num_boot, q, **kw):
    """_ht_2d for one gene pair (hypothesis_test.py:303-364)."""
    ng = trt.shape[0]
    good = np.zeros(ng, dtype=bool)
    bc = np.full((ng, num_boot + 1), np.nan)
    for j in range(ng):
        if np.isnan(true_corr[j]) or abs(true_corr[j]) == 1:
            continue
        bc[j, 0] = true_corr[j]
        r = np.random.random(2)
        r0 = np.random.random()
        c, v1, v2 = bootstrap_2d(cols1[j], cols2[j], approx_sf[j], q[j], int(num_boot), r, r0)
        corr = corr_from_cov(c, v1, v2)
        vals = fill_invalid_corr(corr)
        if np.all(np.isnan(vals)):
            continue
        good[j] = True
        bc[j, 1:] = vals
    if good.sum() == 0:
        return np.nan, np.nan, np.nan
    return regress_2d(cov_[good], trt[good], bc[good], np.asarray(Nc)[good], **kw)
