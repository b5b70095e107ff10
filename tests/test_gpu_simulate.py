"""Device simulator (memento.simulate, csrc/simulate.hip; reference memento/simulate.py:52-68, :91-115).  Draw-level parity with
numpy / scipy is unpinned by construction (own counter-based generators): the checks are exact structural invariants of the
capture process plus statistical agreement with the distributions the reference samples, and -- the reference's own acceptance
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
