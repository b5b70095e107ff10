"""Pin the CPU oracle (oracle/memento_oracle.py) against fixtures produced by the REAL reference
(tests/golden/make_golden.py).  CPU only."""

import numpy as np
import pytest

from conftest import golden_inputs
from oracle import memento_oracle as orc

RTOL = 1e-9  # oracle vs reference: same libraries, same operation order -> far tighter than the 1e-5 bar


@pytest.mark.parametrize("fx", ["api_small", "api_approx"])
def test_size_factors_and_moments(fx, request):
    g = request.getfixturevalue(fx)
    X, gid, ng, q = golden_inputs(g)
    sf, lv_mask, am, av = orc.setup_size_factors(X, q.mean())
    np.testing.assert_allclose(sf, g["size_factor"], rtol=RTOL)
    np.testing.assert_allclose(am, g["all_m"], rtol=RTOL)
    np.testing.assert_allclose(av, g["all_v"], rtol=RTOL, atol=1e-14)
    assert list(g["in_gene_names"][lv_mask]) == list(g["least_variable_genes"])
    approx, bidx, means = orc.bin_size_factor(sf)
    np.testing.assert_array_equal(approx, g["approx_sf"])
    gq = np.array([q[gid == k].mean() for k in range(ng)])
    np.testing.assert_allclose(gq, g["group_q"], rtol=1e-15)
    mom = orc.compute_1d_moments(X, gid, ng, sf, gq)
    np.testing.assert_array_equal(mom["overall_gene_filter"], g["overall_gene_filter"])
    np.testing.assert_array_equal(mom["gene_filter"], g["gene_filter"])
    np.testing.assert_array_equal(mom["gene_rv_filter"], g["gene_rv_filter"])
    np.testing.assert_allclose(mom["mean"], g["mean"], rtol=RTOL)
    np.testing.assert_allclose(mom["var"], g["var"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(mom["res_var"], g["res_var"], rtol=1e-8, equal_nan=True)
    np.testing.assert_allclose(mom["mv_fit"], g["mv_regressor"], rtol=1e-8)


def test_unique_bins_and_bootstrap(api_small, internals_small):
    g, it = api_small, internals_small
    X, gid, ng, q = golden_inputs(g)
    Xk = X[:, g["overall_gene_filter"]].tocsc()
    B = int(it["num_boot"])
    for n in range(int(it["n_picks"])):
        p = f"p{n}_"
        gi, grp = int(it[p + "gene"]), int(it[p + "group"])
        sel = np.flatnonzero(gid == grp)
        vals = np.asarray(Xk[:, gi].todense()).ravel()[sel]
        asf = g["approx_sf"][sel]
        inv_sf, inv_sf_sq, expr, mult = orc.unique_bins_1d(vals, asf, float(it[p + "r"]), float(it[p + "r0"]))
        np.testing.assert_array_equal(expr, it[p + "expr"])
        np.testing.assert_array_equal(mult, it[p + "counts"])
        np.testing.assert_array_equal(inv_sf, it[p + "inv_sf"])
        w = orc.multinomial_weights(len(sel), mult, B)
        np.testing.assert_array_equal(w, it[p + "weights"])
        m, v = orc.replicate_moments_1d(expr, inv_sf, inv_sf_sq, w, len(sel), float(it[p + "q"]))
        np.testing.assert_array_equal(m, it[p + "mean"])      # same op order -> bit-exact
        np.testing.assert_array_equal(v, it[p + "var"])


def test_regress_and_asl(regress_asl):
    r = regress_asl
    out = orc.regress_1d(r["rg_cov"], r["rg_trt"], r["rg_bm"], r["rg_bv"], r["rg_Nc"], resampling="bootstrap", approx=False)
    for k, v in zip(["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"], out):
        np.testing.assert_allclose(v, r["rg_" + k], rtol=1e-9, err_msg=k)
    out1 = orc.regress_1d(r["rg_cov"][:, :1], np.ones((8, 1)), r["rg_bm"], r["rg_bv"], r["rg_Nc"], resampling="bootstrap", approx=True)
    for k, v in zip(["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"], out1):
        np.testing.assert_allclose(v, r["rg1_" + k], rtol=1e-9, err_msg=k)
    for tag, approx in [("count", False), ("tail", False), ("approx", True), ("negtail", False)]:
        got = orc.compute_asl(r["asl_in_" + tag], "bootstrap", approx)
        np.testing.assert_allclose(got, r["asl_out_" + tag], rtol=1e-9, err_msg=tag)


@pytest.mark.parametrize("fx", ["api_small", "api_approx"])  # (api_meanonly uses another estimator: GPU tests pin it directly)
def test_ht_1d_end_to_end(fx, request):
    """Full ht_1d_moments replay with the seeded global stream (num_cpus=1 semantics)."""
    g = request.getfixturevalue(fx)
    X, gid, ng, q = golden_inputs(g)
    keep = g["overall_gene_filter"]
    mom = dict(mean=g["mean"], res_var=g["res_var"], mv_fit=g["mv_regressor"])
    np.random.seed(int(g["ht_seed"]))
    out = orc.ht_1d(X[:, keep], gid, ng, g["approx_sf"], mom, g["covariate"], g["treatment"], int(g["num_boot"]),
                    g["group_q"], resampling="bootstrap", approx=bool(g["approx"]))
    for k, v in zip(["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"], out):
        np.testing.assert_allclose(v, g["ht_" + k], rtol=1e-7, equal_nan=True, err_msg=k)


def test_2d(api_small):
    g = api_small
    X, gid, ng, q = golden_inputs(g)
    keep = g["overall_gene_filter"]
    Xk = X[:, keep].tocsc()
    i1, i2 = g["pair_idx1"], g["pair_idx2"]
    sel = [np.flatnonzero(gid == k) for k in range(ng)]
    for k in range(ng):
        cov = orc.cov_2d_sparse(Xk[sel[k]], g["size_factor"][sel[k]], g["group_q"][k], i1, i2)
        np.testing.assert_allclose(cov, g["cov2d"][k], rtol=1e-9, atol=1e-15)
        corr = orc.corr_from_cov(cov, g["var"][k][i1], g["var"][k][i2])
        np.testing.assert_allclose(corr, g["corr2d"][k], rtol=1e-9, equal_nan=True)
    cm = orc.corr_matrix(Xk[sel[0]], g["size_factor"][sel[0]], g["group_q"][0], g["var"][0])
    np.testing.assert_allclose(cm, g["corr_matrix_g0"], rtol=1e-8, atol=1e-12, equal_nan=True)
    # ht_2d replay: de-duplicated unordered pairs, self pairs skipped (main.py:467-482)
    np.random.seed(int(g["ht_seed"]) + 1)
    Nc = np.array([len(s) for s in sel], dtype=float)
    asf = [g["approx_sf"][s] for s in sel]
    P = len(i1)
    coef, se, asl = (np.full(P, np.nan) for _ in range(3))
    seen = {}
    for p in range(P):
        a, b = int(i1[p]), int(i2[p])
        if a == b:
            continue
        key = frozenset((a, b))
        if key in seen:
            seen[key].append(p)
            continue
        seen[key] = [p]
    for key, plist in seen.items():
        p = plist[0]
        a, b = int(i1[p]), int(i2[p])
        ca = np.asarray(Xk[:, a].todense()).ravel()
        cb = np.asarray(Xk[:, b].todense()).ravel()
        res = orc.ht_2d_pair(g["corr2d"][:, p], [ca[s] for s in sel], [cb[s] for s in sel], asf, g["covariate"],
                             g["treatment"], Nc, int(g["num_boot"]), g["group_q"], resampling="bootstrap", approx=False)
        for pp in plist:
            coef[pp], se[pp], asl[pp] = [np.atleast_1d(x)[0] for x in res]
    np.testing.assert_allclose(coef, g["ht2_corr_coef"], rtol=1e-7, equal_nan=True)
    np.testing.assert_allclose(se, g["ht2_corr_se"], rtol=1e-7, equal_nan=True)
    np.testing.assert_allclose(asl, g["ht2_corr_asl"], rtol=1e-7, equal_nan=True)


def test_ht_1d_resample_rep(api_small):
    """resample_rep=True: the oracle replays the reference's np.random.choice draws (hypothesis_test.py:273-286)."""
    g = api_small
    X, gid, ng, q = golden_inputs(g)
    keep = g["overall_gene_filter"]
    mom = dict(mean=g["mean"], res_var=g["res_var"], mv_fit=g["mv_regressor"])
    np.random.seed(int(g["ht_seed"]) + 2)
    out = orc.ht_1d(X[:, keep], gid, ng, g["approx_sf"], mom, g["covariate"], g["treatment"], int(g["num_boot"]),
                    g["group_q"], resampling="bootstrap", approx=False, resample_rep=True)
    for k, v in zip(["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"], out):
        np.testing.assert_allclose(v, g["htrr_" + k], rtol=1e-7, equal_nan=True, err_msg=k)


@pytest.mark.parametrize("tag,approx,off", [("exact", False, 0), ("approx", True, 1)])
def test_ht_1d_permutation_resampling(api_small, api_perm, tag, approx, off):
    """resampling='permutation': the null is not centred on the observed value (hypothesis_test.py:66-70)."""
    g, gp = api_small, api_perm
    X, gid, ng, q = golden_inputs(g)
    keep = g["overall_gene_filter"]
    mom = dict(mean=g["mean"], res_var=g["res_var"], mv_fit=g["mv_regressor"])
    np.random.seed(int(gp["ht_seed"]) + off)
    out = orc.ht_1d(X[:, keep], gid, ng, g["approx_sf"], mom, g["covariate"], g["treatment"], int(gp["num_boot"]),
                    g["group_q"], resampling="permutation", approx=approx)
    for k, v in zip(["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"], out):
        np.testing.assert_allclose(v, gp[f"ht_{tag}_{k}"], rtol=1e-7, equal_nan=True, err_msg=k)


@pytest.mark.parametrize("tag,approx", [("exact", False), ("approx", True)])
def test_regress_2d_resample_rep(regress2d_rr, tag, approx):
    """_regress_2d with resample_rep=True (hypothesis_test.py:393-404), replaying the reference's np.random.choice draws
    (6 groups: no resampled column is degenerate in this fixture, so the agreement is exact)."""
    r = regress2d_rr
    np.random.seed(int(r["np_seed"]))
    coef, se, asl = orc.regress_2d(r["cov"], r["trt"], r["boot_corr"], r["Nc"], resampling="bootstrap", approx=approx, resample_rep=True)
    np.testing.assert_allclose(coef, r[f"coef_{tag}"], rtol=1e-9)
    np.testing.assert_allclose(se, r[f"se_{tag}"], rtol=1e-7)
    np.testing.assert_allclose(asl, r[f"asl_{tag}"], rtol=1e-7)


def test_corr_matrix_with_nonpositive_variances(corrmat_negvar):
    """estimator._hyper_corr_symmetric on groups holding genes whose variance estimate is <= 0 (estimator.py:259-268): the
    reference NaNs only copies, so two negative variances give a finite correlation and the diagonal of such a gene is -1."""
    from conftest import golden_inputs

    g = corrmat_negvar
    X, gid, ng, q = golden_inputs(g)
    Xk = X[:, g["overall_gene_filter"]]
    seen_neg_pair = False
    for k in range(ng):
        sel = np.flatnonzero(gid == k)
        want = g[f"corr_matrix_{k}"]
        var = g["var"][k].copy()
        cm = orc.corr_matrix(Xk[sel], g["size_factor"][sel], g["group_q"][k], var)
        np.testing.assert_array_equal(var, g["var"][k])                     # no side effect on the stored moments
        np.testing.assert_array_equal(np.isnan(cm), np.isnan(want))
        np.testing.assert_allclose(cm, want, rtol=1e-8, atol=1e-12, equal_nan=True)
        neg = np.flatnonzero(g["var"][k] < 0)
        if len(neg) >= 2:
            seen_neg_pair = True
            assert np.isfinite(want[np.ix_(neg, neg)]).any()                # the case the round-1 oracle got wrong
    assert seen_neg_pair


@pytest.mark.parametrize("tag,approx", [("exact", False), ("approx", True)])
def test_ht_1d_resample_rep_16_groups(api_rr16, tag, approx):
    """resample_rep=True through the whole 1D path with 2 x 8 groups (no degenerate resampled column can be expected:
    P = 3e-5 per column), numeric covariate besides the intercept: the oracle must reproduce the real reference's
    coefficients, standard errors and p-values WITHOUT any special rule for degenerate columns."""
    g = api_rr16
    X, gid, ng, q = golden_inputs(g)
    keep = g["overall_gene_filter"]
    mom = dict(mean=g["mean"], res_var=g["res_var"], mv_fit=g["mv_regressor"])
    np.random.seed(int(g[f"seed_{tag}"]))
    out = orc.ht_1d(X[:, keep], gid, ng, g["approx_sf"], mom, g["covariate"], g["treatment"], int(g["num_boot"]),
                    g["group_q"], resampling="bootstrap", approx=approx, resample_rep=True)
    for k, v in zip(["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"], out):
        assert np.isfinite(g[f"htrr_{tag}_{k}"]).all()
        np.testing.assert_allclose(v, g[f"htrr_{tag}_{k}"], rtol=1e-7, equal_nan=True, err_msg=k)


def test_ht_2d_resample_rep_16_groups(api_rr16):
    """ht_2d_moments(resample_rep=True) of the real reference, 16 groups: per pair the three hash uniforms of every live
    group, then the two np.random.choice draws of _regress_2d (hypothesis_test.py:395-398)."""
    g = api_rr16
    X, gid, ng, q = golden_inputs(g)
    Xk = X[:, g["overall_gene_filter"]].tocsc()
    i1, i2 = g["pair_idx1"], g["pair_idx2"]
    sel = [np.flatnonzero(gid == k) for k in range(ng)]
    Nc = np.array([len(s_) for s_ in sel], dtype=float)
    asf = [g["approx_sf"][s_] for s_ in sel]
    np.random.seed(int(g["seed_2d"]))
    coef, se, asl = (np.full(len(i1), np.nan) for _ in range(3))
    seen = {}
    for p in range(len(i1)):
        a, b = int(i1[p]), int(i2[p])
        assert a != b
        key = frozenset((a, b))
        seen.setdefault(key, []).append(p)
    for key, plist in seen.items():
        p = plist[0]
        a, b = int(i1[p]), int(i2[p])
        ca = np.asarray(Xk[:, a].todense()).ravel()
        cb = np.asarray(Xk[:, b].todense()).ravel()
        tc = g["true_corr"][:, p]
        res = orc.ht_2d_pair(tc, [ca[s_] for s_ in sel], [cb[s_] for s_ in sel], asf, g["covariate"], g["treatment"], Nc,
                             int(g["num_boot"]), g["group_q"], resampling="bootstrap", approx=False, resample_rep=True)
        for pp in plist:
            coef[pp], se[pp], asl[pp] = [np.atleast_1d(x)[0] for x in res]
    np.testing.assert_allclose(coef, g["ht2rr_corr_coef"], rtol=1e-7, equal_nan=True)
    np.testing.assert_allclose(se, g["ht2rr_corr_se"], rtol=1e-7, equal_nan=True)
    np.testing.assert_allclose(asl, g["ht2rr_corr_asl"], rtol=1e-6, equal_nan=True)


def test_c1_pbmc3k_shape_end_to_end(api_c1):
    """BASELINE.json configs[0]: 2.7k cells x 1.8k genes, 2 groups, 100 bootstraps -- the whole 1D path of the real reference
    (size factors, moments, filters, exact-ASL hypothesis test) reproduced by the oracle."""
    from conftest import c1_inputs

    g = api_c1
    X, gid, ng, q = c1_inputs(g)
    sf, _, _, _ = orc.setup_size_factors(X, q)
    np.testing.assert_allclose(sf, g["size_factor"], rtol=1e-12)
    gq = np.full(ng, q)
    mom = orc.compute_1d_moments(X, gid, ng, sf, gq)
    np.testing.assert_array_equal(mom["overall_gene_filter"], g["overall_gene_filter"])
    np.testing.assert_allclose(mom["mean"], g["mean"], rtol=1e-11)
    np.testing.assert_allclose(mom["res_var"], g["res_var"], rtol=1e-8, equal_nan=True)
    approx_sf, _, _ = orc.bin_size_factor(sf)
    np.random.seed(71)
    out = orc.ht_1d(X[:, mom["overall_gene_filter"]], gid, ng, approx_sf, mom, g["covariate"], g["treatment"], 100, gq,
                    resampling="bootstrap", approx=False)
    for k, v in zip(["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"], out):
        np.testing.assert_allclose(v, g["ht_" + k], rtol=1e-7, equal_nan=True, err_msg=k)
