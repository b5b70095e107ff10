"""Device simulator (memento.simulate, csrc/simulate.hip; reference memento/simulate.py:52-68, :91-115).  Draw-level parity with
numpy / scipy is unpinned by construction (own counter-based generators): the checks are exact structural invariants of the
capture process, statistical agreement with the distributions the reference samples -- incl. seeded samples of the REAL
reference's simulate.py (fixture simulate_ref, bottom of this file), whose deterministic extract_parameters is matched to 1e-9 -- and -- the reference's own acceptance
test (analysis/simulation/estimator_validation.ipynb) -- recovery of the simulated moments through the HIP estimators."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dense(csr):
    from scrna_parameter_estimation_amd.memento.simulate import device_csr_to_scipy

    return np.asarray(device_csr_to_scipy(csr).todense())


def test_negative_binomial_transcriptomes_match_their_distribution():
    from scrna_parameter_estimation_amd.memento import simulate

    n = 40_000
    means = np.array([0.05, 0.4, 2.0, 9.0, 25.0, 140.0, 0.0, 3.0])
    variances = np.array([0.06, 0.9, 3.0, 40.0, 400.0, 900.0, 0.0, 2.0])        # the last one is under-dispersed: floored (simulate.py:63)
    t = simulate.simulate_transcriptomes(n, means, variances, None, norm_cov='indep', seed=3)
    Z = _dense(t.to_device_csr())
    assert Z.shape == (n, 8) and (Z >= 0).all() and (Z == np.round(Z)).all() and (Z[:, 6] == 0).all()
    disp = (variances - means) / np.where(means > 0, means, 1) ** 2
    disp[~(disp > 0)] = 1e-5
    want_var = means + disp * means ** 2
    for g in (0, 1, 2, 3, 4, 5, 7):
        se_mean = np.sqrt(want_var[g] / n)
        assert abs(Z[:, g].mean() - means[g]) < 5 * se_mean, (g, Z[:, g].mean(), means[g])
        assert abs(Z[:, g].var() / want_var[g] - 1) < 0.08, (g, Z[:, g].var(), want_var[g])
    # same seed -> same matrix; totals kernel == row sums of the written matrix
    np.testing.assert_array_equal(_dense(t.to_device_csr()), Z)
    from scrna_parameter_estimation_amd import engine

    np.testing.assert_array_equal(engine.host(t.totals())[:n], Z.sum(axis=1).astype(np.int64))
    # zero-probability of the gamma-Poisson mixture: P(z = 0) = (theta / (theta + mu))^theta
    th = 1.0 / disp
    for g in (0, 1, 2):
        p0 = (th[g] / (th[g] + means[g])) ** th[g]
        assert abs((Z[:, g] == 0).mean() - p0) < 5 * np.sqrt(p0 * (1 - p0) / n)


@pytest.mark.parametrize("process", ["hyper", "poisson"])
def test_capture_sampling_invariants(process):
    from scrna_parameter_estimation_amd import engine
    from scrna_parameter_estimation_amd.memento import simulate

    rng = np.random.default_rng(0)
    n, G = 6_000, 60
    means = rng.lognormal(0.5, 1.2, size=G)
    variances = means + 0.3 * means ** 2
    t = simulate.simulate_transcriptomes(n, means, variances, None, norm_cov='indep', seed=11)
    Z = _dense(t.to_device_csr())
    q = 0.1
    qs, cap = simulate.capture_sampling(t, q, process=process)
    X = _dense(cap)
    assert X.shape == Z.shape and (X >= 0).all() and (qs == q).all()
    idx, ptr = engine.host(cap.indices), engine.host(cap.indptr)
    assert all((np.diff(idx[ptr[i]:ptr[i + 1]]) > 0).all() for i in range(0, n, 500))        # canonical CSR rows
    if process == "hyper":
        assert (X <= Z).all()                                                                # molecules are drawn WITHOUT replacement
        np.testing.assert_array_equal(X.sum(axis=1), np.rint(q * Z.sum(axis=1)))             # exactly round(q * total) per cell (simulate.py:107)
        # conditional on z, x ~ hypergeometric: E[x] = n_cap * z / total
        expect = (np.rint(q * Z.sum(axis=1)) / np.maximum(Z.sum(axis=1), 1))[:, None] * Z
    else:
        expect = q * Z                                                                       # x ~ Poisson(q z)
        var_ratio = ((X - expect) ** 2).sum() / expect.sum()
        assert abs(var_ratio - 1) < 0.05                                                     # Poisson: variance == mean
    assert abs(X.sum() / expect.sum() - 1) < 0.01
    g_tot = X.sum(axis=0) / expect.sum(axis=0)
    assert (np.abs(g_tot - 1) < 6 / np.sqrt(np.maximum(expect.sum(axis=0), 1))).all()
    # per-cell capture rates from a Beta law (q_sq given): mean q, second moment q_sq; hypergeometric totals follow them
    np.random.seed(5)
    qs2, cap2 = simulate.capture_sampling(t, 0.1, q_sq=0.011, process=process)
    assert abs(qs2.mean() - 0.1) < 0.002 and abs((qs2 ** 2).mean() - 0.011) < 0.0005
    if process == "hyper":
        np.testing.assert_array_equal(_dense(cap2).sum(axis=1), np.rint(qs2 * Z.sum(axis=1)))


def test_simulated_moments_are_recovered_through_the_hip_estimators():
    """Simulate -> hypergeometric capture at q = 0.1 -> extract_parameters (hypergeometric-corrected relative moments, the K1 kernel)
    recovers the simulated absolute means and variances: the estimator-validation loop of the reference."""
    from scrna_parameter_estimation_amd.memento import simulate

    rng = np.random.default_rng(2)
    n, G, q = 60_000, 400, 0.1
    z_mean = rng.lognormal(1.0, 1.3, size=G)
    z_var = z_mean + rng.uniform(0.1, 0.6, size=G) * z_mean ** 2
    t = simulate.simulate_transcriptomes(n, z_mean, z_var, None, norm_cov='indep', seed=21)
    qs, cap = simulate.capture_sampling(t, q, process='hyper')
    (x_mean, x_var), (zm, zv), Nc, good = simulate.extract_parameters(cap, q=q, min_mean=0.001)
    assert len(good) > 0.9 * G and abs(Nc.mean() / z_mean.sum() - 1) < 0.02
    rel = zm / z_mean[good] - 1
    assert abs(np.median(rel)) < 0.03 and np.percentile(np.abs(rel), 90) < 0.08
    big = z_mean[good] > 5                                   # variances of well-expressed genes: within ~20 %, no systematic bias
    relv = zv[big] / z_var[good][big] - 1
    assert abs(np.median(relv)) < 0.08 and np.percentile(np.abs(relv), 90) < 0.3
    assert np.corrcoef(np.log(zv[big]), np.log(z_var[good][big]))[0, 1] > 0.99
    # feeds the whole API: setup_memento straight from the device CSR
    import pandas as pd
    import scipy.sparse as sp

    from scrna_parameter_estimation_amd import AnnDataLite, memento

    obs = pd.DataFrame({"q": np.full(n, q), "grp": rng.integers(0, 2, size=n)})
    adata = AnnDataLite(sp.csr_matrix((n, G), dtype=np.float32), obs, pd.DataFrame(index=[f"g{i}" for i in range(G)]))
    memento.setup_memento(adata, q_column="q", device_csr=cap)
    memento.create_groups(adata, label_columns=["grp"])
    memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
    assert len(adata.uns["memento"]["gene_list"]) > 0.5 * G


def test_copula_quantiles_equal_scipy_on_the_same_scores():
    """Gaussian-copula branch, kernel level: nb = nbinom.ppf(Phi(score)) (reference simulate.py:80-81).  The scores live on the
    device, so scipy can be asked for the quantiles of exactly the same uniforms: the counts must agree element by element (a
    uniform within rounding of a CDF step may fall on either side: at most a handful of +-1 differences in 240,000 draws)."""
    import scipy.stats as stats

    from scrna_parameter_estimation_amd import engine
    from scrna_parameter_estimation_amd.memento import simulate

    n = 30_000
    means = np.array([0.05, 0.4, 2.0, 9.0, 25.0, 140.0, 2500.0, 0.0])
    variances = np.array([0.06, 0.9, 3.0, 40.0, 400.0, 900.0, 2600.0, 0.0])     # (2500, 2600): nearly Poisson, theta = 62,500
    rng = np.random.default_rng(4)
    A = rng.normal(size=(8, 8))
    cov = A @ A.T + 0.5 * np.eye(8)
    sd = np.sqrt(np.diag(cov))
    scores = simulate.correlated_scores(n, cov / np.outer(sd, sd), seed=9)
    Y = engine.host(scores).astype(np.float64)                                  # [genes][cells]
    assert Y.shape == (8, n) and abs(Y.mean()) < 0.02 and abs(Y.std() - 1) < 0.02
    np.testing.assert_allclose(np.corrcoef(Y), cov / np.outer(sd, sd), atol=0.03)
    disp = (variances - means) / np.where(means > 0, means, 1) ** 2
    disp[~(disp > 0)] = 1e-5
    t = simulate.Transcriptomes(n, means, 1.0 / disp, seed=1, scores=scores)    # no cell sizes: the raw quantiles
    Z = _dense(t.to_device_csr())
    want = np.zeros_like(Z)
    for g in range(7):
        want[:, g] = stats.nbinom.ppf(stats.norm.cdf(Y[g]), *simulate.convert_params_nb(means[g], 1.0 / disp[g]))
    diff = np.abs(Z - want)
    assert diff.max() <= 1 and (diff != 0).sum() <= 5, ((diff != 0).sum(), diff.max())
    assert (Z[:, 7] == 0).all()


def test_copula_transcriptomes_cell_sizes_and_dependence():
    """Gaussian-copula branch through the API (simulate.py:70-89): every cell sums to its drawn size up to the rounding of its
    genes, correlated gene pairs come out rank-correlated as a Gaussian copula predicts (Spearman = 6/pi asin(rho/2)),
    uncorrelated ones do not, and the capture step runs on top."""
    import scipy.stats as stats

    from scrna_parameter_estimation_amd.memento import simulate

    rng = np.random.default_rng(7)
    n, G = 30_000, 40
    means = rng.uniform(8, 60, size=G)
    variances = means + 0.2 * means ** 2
    cov = np.eye(G)
    cov[0, 1] = cov[1, 0] = 0.8
    cov[2, 3] = cov[3, 2] = -0.6
    cov *= np.outer(np.linspace(1, 3, G), np.linspace(1, 3, G))                 # a covariance, not a correlation matrix
    Nc = rng.integers(1200, 1500, size=500)
    np.random.seed(3)
    t = simulate.simulate_transcriptomes(n, means, variances, Nc, norm_cov=cov, seed=5)
    Z = _dense(t.to_device_csr())
    np.random.seed(3)
    sizes = np.random.choice(Nc, size=n)                                        # the draw the call made (global numpy stream)
    assert np.abs(Z.sum(axis=1) - sizes).max() <= G / 2 + 1 and abs((Z.sum(axis=1) - sizes).mean()) < 0.5
    sp01 = stats.spearmanr(Z[:, 0], Z[:, 1])[0]
    sp23 = stats.spearmanr(Z[:, 2], Z[:, 3])[0]
    sp45 = stats.spearmanr(Z[:, 4], Z[:, 5])[0]
    assert abs(sp01 - 6 / np.pi * np.arcsin(0.4)) < 0.05, sp01
    assert abs(sp23 + 6 / np.pi * np.arcsin(0.3)) < 0.05, sp23
    assert abs(sp45) < 0.03, sp45
    rel = Z.mean(axis=0) / Z.mean(axis=0).sum() / (means / means.sum()) - 1     # relative abundances follow the NB means
    assert np.abs(rel).max() < 0.03
    qs, cap = simulate.capture_sampling(t, 0.1, process='hyper')
    X = _dense(cap)
    assert (X <= Z).all()
    np.testing.assert_array_equal(X.sum(axis=1), np.rint(0.1 * Z.sum(axis=1)))
    # default covariance (norm_cov=None): a random SPD matrix, as the reference draws with sklearn's make_spd_matrix
    S = simulate.make_spd_matrix(12, np.random.default_rng(0))
    assert np.allclose(S, S.T) and np.linalg.eigvalsh(S).min() > 0
    t2 = simulate.simulate_transcriptomes(2000, means[:12], variances[:12], Nc, seed=6)
    assert _dense(t2.to_device_csr()).shape == (2000, 12)


# ---- against the REAL reference's simulate.py (fixture tests/golden/simulate_ref.npz, written by tests/golden/make_golden.py) --------
def _ref():
    import os

    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simulate_ref.npz"), allow_pickle=False))
    return g


def _same_distribution(a, b, what):
    """Per-gene two-sample checks of integer samples a, b [cells][genes]: Kolmogorov-Smirnov (conservative on discrete data), means
    within 4.5 pooled standard errors, variances within 30 %."""
    from scipy import stats

    G = a.shape[1]
    for j in range(G):
        x, y = a[:, j].astype(np.float64), b[:, j].astype(np.float64)
        assert stats.ks_2samp(x, y).pvalue > 1e-4 / G * 10, (what, j, "KS")
        se = np.sqrt(x.var() / len(x) + y.var() / len(y))
        assert abs(x.mean() - y.mean()) <= 4.5 * se + 1e-9, (what, j, x.mean(), y.mean())
        if x.var() > 0.05 and y.var() > 0.05:
            assert 0.7 < x.var() / y.var() < 1.0 / 0.7, (what, j, x.var(), y.var())


def test_extract_parameters_matches_the_reference():
    """simulate.extract_parameters (reference simulate.py:13-33; deterministic) on the fixture's CSR: every output to 1e-9."""
    import scipy.sparse as sp

    from scrna_parameter_estimation_amd.memento import simulate

    g = _ref()
    X = sp.csr_matrix((g["in_data"].astype(np.float32), g["in_indices"], g["in_indptr"]), shape=tuple(g["in_shape"]))
    (x_mean, x_var), (z_mean, z_var), Nc, good_idx = simulate.extract_parameters(X, q=float(g["q"]), min_mean=float(g["min_mean"]))
    np.testing.assert_array_equal(good_idx, g["good_idx"])
    for got, k in ((x_mean, "x_mean"), (x_var, "x_var"), (z_mean, "z_mean"), (z_var, "z_var"), (Nc, "Nc")):
        np.testing.assert_allclose(got, g[k], rtol=1e-9, atol=1e-300, err_msg=k)


def test_simulated_transcriptomes_are_distributed_like_the_references():
    """simulate_transcriptomes -- independent negative binomials and the Gaussian copula with a given covariance (reference
    simulate.py:52-89) -- against seeded samples of the real reference with the same parameters: per-gene marginals
    (KS / mean / variance) and, for the copula, the Spearman correlation of neighbouring genes."""
    from scipy import stats

    from scrna_parameter_estimation_amd.memento import simulate

    g = _ref()
    n, means, variances, Nc = int(g["n_cells"]), g["sim_means"], g["sim_variances"], g["Nc"]
    t = simulate.simulate_transcriptomes(n, means, variances, Nc, norm_cov="indep", seed=17)
    _same_distribution(_dense(t.to_device_csr()), g["indep"], "indep")
    np.random.seed(5)
    tc = simulate.simulate_transcriptomes(n, means, variances, Nc, norm_cov=g["norm_cov"], seed=18)
    mine, ref = _dense(tc.to_device_csr()), g["copula"]
    _same_distribution(mine, ref, "copula")
    np.testing.assert_allclose(mine.sum(axis=1).mean(), ref.sum(axis=1).mean(), rtol=0.02)        # cells rescaled to sizes drawn from Nc
    for j in range(0, mine.shape[1] - 1, 3):
        r_mine = stats.spearmanr(mine[:, j], mine[:, j + 1])[0]
        r_ref = stats.spearmanr(ref[:, j], ref[:, j + 1])[0]
        assert abs(r_mine - r_ref) < 0.07, (j, r_mine, r_ref)
        k = (j + 24) % mine.shape[1]          # far apart in the copula (0.7^24): what is left is the shared cell size
        far, far_ref = stats.spearmanr(mine[:, j], mine[:, k])[0], stats.spearmanr(ref[:, j], ref[:, k])[0]
        assert abs(far - far_ref) < 0.07 and abs(far) < 0.25, (j, far, far_ref)


@pytest.mark.parametrize("process,key,q_sq", [("hyper", "cap_hyper", None), ("poisson", "cap_poisson", None), ("hyper", "cap_beta", 0.012)])
def test_capture_sampling_is_distributed_like_the_references(process, key, q_sq):
    """capture_sampling (reference simulate.py:91-115; hypergeometric, Poisson, Beta-distributed capture rates) of independent NB
    transcriptomes against the real reference's captured counts from the same parameters: per-gene marginals and the per-cell
    capture rates."""
    from scipy import stats

    from scrna_parameter_estimation_amd.memento import simulate

    g = _ref()
    n, means, variances, Nc = int(g["n_cells"]), g["sim_means"], g["sim_variances"], g["Nc"]
    t = simulate.simulate_transcriptomes(n, means, variances, Nc, norm_cov="indep", seed=23)
    np.random.seed(9)
    qs, cap = simulate.capture_sampling(t, 0.1, q_sq=q_sq, process=process)
    _same_distribution(_dense(cap), g[key], key)
    ref_qs = g["qs_" + key.split("_")[1]]
    if q_sq is None:
        np.testing.assert_array_equal(qs, ref_qs)
    else:
        assert stats.ks_2samp(qs, ref_qs).pvalue > 1e-3 and abs(qs.mean() - ref_qs.mean()) < 0.003
