"""End-to-end GPU parity: the memento.* API on the HIP path against fixtures produced by the REAL
reference (tests/golden/api_*.npz).  Tolerances: integer masks / gene lists bit-exact; floating point
1e-9 (the north-star bar is 1e-5 relative)."""

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _adata_from_golden(g):
    from scrna_parameter_estimation_amd import AnnDataLite

    X = sp.csr_matrix((g["in_data"].astype(np.float32), g["in_indices"], g["in_indptr"]), shape=tuple(g["in_shape"]))
    obs = pd.DataFrame({"cond": g["in_cond"], "rep": g["in_rep"], "q": g["in_q"]}, index=[f"c{i}" for i in range(X.shape[0])])
    var = pd.DataFrame(index=g["in_gene_names"].tolist())
    return AnnDataLite(X, obs, var)


def _run_to_moments(g):
    from scrna_parameter_estimation_amd import memento

    adata = _adata_from_golden(g)
    memento.setup_memento(adata, q_column="q", estimator_type=str(g["estimator_type"]) if "estimator_type" in g else "hyper_relative")
    memento.create_groups(adata, label_columns=["cond", "rep"])
    memento.compute_1d_moments(adata, min_perc_group=0.7)
    return memento, adata


@pytest.mark.parametrize("fx", ["api_small", "api_approx", "api_meanonly"])
def test_setup_and_moments(fx, request):
    g = request.getfixturevalue(fx)
    memento, adata = _run_to_moments(g)
    m = adata.uns["memento"]
    np.testing.assert_allclose(adata.obs["memento_size_factor"].values, g["size_factor"], rtol=1e-12)
    assert m["least_variable_genes"] == list(g["least_variable_genes"])
    np.testing.assert_allclose(m["all_1d_moments"][0], g["all_m"], rtol=1e-11)
    np.testing.assert_allclose(m["all_1d_moments"][1], g["all_v"], rtol=1e-9, atol=1e-13)
    groups = list(g["groups"])
    assert m["groups"] == groups
    np.testing.assert_allclose([m["group_q"][k] for k in groups], g["group_q"], rtol=1e-14)
    assert [m["group_cells"][k].shape[0] for k in groups] == list(g["group_ncells"])
    np.testing.assert_array_equal(m["all_approx_size_factor"], g["approx_sf"])
    np.testing.assert_array_equal(m["overall_gene_filter"], g["overall_gene_filter"])      # masks: bit-exact
    assert m["gene_list"] == list(g["gene_list"])
    assert list(adata.var.index) == list(g["gene_list"])
    for i, k in enumerate(groups):
        np.testing.assert_array_equal(m["gene_filter"][k], g["gene_filter"][i])
        np.testing.assert_array_equal(m["gene_rv_filter"][k], g["gene_rv_filter"][i])
        np.testing.assert_allclose(m["1d_moments"][k][0], g["mean"][i], rtol=1e-11)
        np.testing.assert_allclose(m["1d_moments"][k][1], g["var"][i], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(m["1d_moments"][k][2], g["res_var"][i], rtol=1e-8, atol=1e-13, equal_nan=True)
        np.testing.assert_allclose(m["mv_regressor"][k], g["mv_regressor"], rtol=1e-8, atol=1e-12)
    gm, gv = memento.get_1d_moments(adata, groupby="cond")          # cell-count weighted groupby aggregation
    assert [c for c in gm.columns if c != "gene"] == list(g["groupby_cols"])
    np.testing.assert_allclose(gm[list(g["groupby_cols"])].values, g["groupby_mean"], rtol=1e-9, equal_nan=True)
    np.testing.assert_allclose(gv[list(g["groupby_cols"])].values, g["groupby_var"], rtol=1e-7, atol=1e-12, equal_nan=True)


def _design(memento, adata, g):
    gdf = memento.get_groups(adata)
    cov = pd.DataFrame(g["covariate"], index=gdf.index, columns=["intercept"])
    trt = pd.DataFrame(g["treatment"], index=gdf.index, columns=["cond"])
    return cov, trt


@pytest.mark.parametrize("chain_all_max", [8192, 0])
@pytest.mark.parametrize("fx", ["api_small", "api_approx", "api_meanonly"])
def test_ht_1d_strict_replay_matches_reference(fx, request, chain_all_max, monkeypatch):
    """strict=True replays the reference's global np.random stream (num_cpus=1 semantics): coefficients,
    standard errors and p-values must match the real reference's output -- with every chain in a wave of its own (the default for
    launches of few chains: mm_boot1d_chain) and with the lane-per-chain tile kernel (engine.CHAIN_ALL_MAX = 0)."""
    from scrna_parameter_estimation_amd import engine

    monkeypatch.setattr(engine, "CHAIN_ALL_MAX", chain_all_max)
    g = request.getfixturevalue(fx)
    memento, adata = _run_to_moments(g)
    cov, trt = _design(memento, adata, g)
    np.random.seed(int(g["ht_seed"]))
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=bool(g["approx"]), strict=True)
    ht = adata.uns["memento"]["1d_ht"]
    for k in ["mean_coef", "mean_se", "var_coef", "var_se"]:
        np.testing.assert_allclose(ht[k], g["ht_" + k], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg=k)
    # mean_only fixes every variance at 10, so its "variability" statistics are ~1e-16 rounding noise in the
    # reference as well (p-values of noise): only the DE p-values are meaningful there.
    for k in (["mean_asl"] if fx == "api_meanonly" else ["mean_asl", "var_asl"]):
        np.testing.assert_allclose(ht[k], g["ht_" + k], rtol=1e-5, atol=1e-12, equal_nan=True, err_msg=k)
    bs = adata.uns["memento"]["_hip"].last_bootstrap
    assert (bs.n_tiles == 0) == (chain_all_max > 0) and bs.n_chain > 0
    df = memento.get_1d_ht_result(adata)
    assert list(df.columns) == ["gene", "tx", "de_coef", "de_se", "de_pval", "dv_coef", "dv_se", "dv_pval"]
    assert len(df) == len(g["gene_list"])


@pytest.mark.parametrize("tag,approx,off", [("exact", False, 0), ("approx", True, 1)])
def test_ht_1d_permutation_resampling_matches_reference(api_small, api_perm, tag, approx, off):
    """resampling='permutation' (as common as 'bootstrap' in the reference's analyses): same replicates, un-centred null."""
    g, gp = api_small, api_perm
    memento, adata = _run_to_moments(g)
    cov, trt = _design(memento, adata, g)
    np.random.seed(int(gp["ht_seed"]) + off)
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=int(gp["num_boot"]), num_cpus=1, verbose=0,
                          resampling="permutation", approx=approx, strict=True)
    ht = adata.uns["memento"]["1d_ht"]
    for k in ["mean_coef", "mean_se", "var_coef", "var_se"]:
        np.testing.assert_allclose(ht[k], gp[f"ht_{tag}_{k}"], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg=k)
    for k in ["mean_asl", "var_asl"]:
        np.testing.assert_allclose(ht[k], gp[f"ht_{tag}_{k}"], rtol=1e-5, atol=1e-12, equal_nan=True, err_msg=k)


def test_ht_2d_treatment_for_gene_matches_reference(api_small, api_tfg2d):
    """ht_2d_moments(treatment_for_gene=...) as the reference behaves (main.py:492): columns looked up under the pair's FIRST gene,
    one column per pair; fixture from the real reference with two treatment columns alternating over the first genes."""
    import pandas as pd

    g, gt = api_small, api_tfg2d
    memento, adata = _run_to_moments(g)
    names = np.asarray(adata.var.index)
    pairs = list(zip(names[g["pair_idx1"]].tolist(), names[g["pair_idx2"]].tolist()))
    memento.compute_2d_moments(adata, pairs)
    cov, _ = _design(memento, adata, g)
    gdf = memento.get_groups(adata)
    trt = pd.DataFrame({"cond": (gdf["cond"].astype(int) == 1).astype(float), "rep": (gdf["rep"].astype(int) == 1).astype(float)}, index=gdf.index)
    tfg = {frozenset({a}): [c] for a, c in zip(gt["first_genes"].tolist(), gt["first_gene_column"].tolist())}
    np.random.seed(int(gt["ht_seed"]))
    memento.ht_2d_moments(adata, covariate=cov, treatment=trt, treatment_for_gene=tfg, num_boot=int(gt["num_boot"]), num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=False)
    ht = adata.uns["memento"]["2d_ht"]
    assert ht["treatment_for_gene"] is tfg
    np.testing.assert_allclose(ht["corr_coef"], gt["ht2_corr_coef"], rtol=1e-8, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(ht["corr_se"], gt["ht2_corr_se"], rtol=1e-8, equal_nan=True)
    # Pair 5 (g37, g59) is TIE-PRONE: in ~11 of its 300 replicates the replicate correlations of all good groups are clipped to
    # the same value, the replicate coefficient is 0 up to the last bit and sits exactly ON the threshold of the extreme count
    # (null = -stat): whether such a replicate counts is decided by LAPACK round-off inside the reference's LinearRegression
    # (+-1e-16), here by the fixed weight row.  Coefficient and SE agree to 1e-8; the p-value may differ by those ties / 301.
    tie = np.zeros(len(ht["corr_asl"]), dtype=bool)
    tie[5] = True
    np.testing.assert_allclose(ht["corr_asl"][~tie], gt["ht2_corr_asl"][~tie], rtol=1e-5, equal_nan=True)
    assert abs(ht["corr_asl"][5] - gt["ht2_corr_asl"][5]) <= 12 / 301
    assert np.isfinite(ht["corr_coef"]).sum() >= 8
    with pytest.raises(ValueError):                           # two columns for one pair: the reference cannot store them either
        memento.ht_2d_moments(adata, covariate=cov, treatment=trt, treatment_for_gene={k: ["cond", "rep"] for k in tfg}, num_boot=50,
                              num_cpus=1, verbose=0, resampling="bootstrap")


def test_ht_2d_permutation_resampling_matches_reference(api_small, api_perm):
    g, gp = api_small, api_perm
    memento, adata = _run_to_moments(g)
    names = np.asarray(adata.var.index)
    pairs = list(zip(names[g["pair_idx1"]].tolist(), names[g["pair_idx2"]].tolist()))
    memento.compute_2d_moments(adata, pairs)
    cov, trt = _design(memento, adata, g)
    np.random.seed(int(gp["ht_seed"]) + 2)
    memento.ht_2d_moments(adata, covariate=cov, treatment=trt, num_boot=int(gp["num_boot"]), num_cpus=1, verbose=0,
                          resampling="permutation", approx=False)
    ht = adata.uns["memento"]["2d_ht"]
    np.testing.assert_allclose(ht["corr_coef"], gp["ht2_corr_coef"], rtol=1e-8, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(ht["corr_se"], gp["ht2_corr_se"], rtol=1e-8, equal_nan=True)
    np.testing.assert_allclose(ht["corr_asl"], gp["ht2_corr_asl"], rtol=1e-5, equal_nan=True)


def test_setup_and_moments_options_match_reference(api_small, api_opts):
    """Options the reference's analyses use: setup_memento(filter_mean_thresh, trim_percent, shrinkage, num_bins),
    compute_1d_moments(filter_genes=False) and compute_1d_moments(gene_list=[...]) -- against the real reference."""
    from scrna_parameter_estimation_amd import memento

    g, go = api_small, api_opts
    # (a) other setup parameters, then the whole 1D path
    adata = _adata_from_golden(g)
    memento.setup_memento(adata, q_column="q", filter_mean_thresh=0.05, trim_percent=0.2, shrinkage=0.4, num_bins=20)
    memento.create_groups(adata, label_columns=["cond", "rep"])
    m = adata.uns["memento"]
    np.testing.assert_allclose(adata.obs["memento_size_factor"].values, go["a_size_factor"], rtol=1e-12)
    assert m["least_variable_genes"] == list(go["a_least_variable_genes"])
    memento.compute_1d_moments(adata, min_perc_group=0.5)
    np.testing.assert_array_equal(m["all_approx_size_factor"], go["a_approx_sf"])
    assert m["gene_list"] == list(go["a_gene_list"])
    groups = m["groups"]
    np.testing.assert_allclose(np.stack([m["1d_moments"][k][0] for k in groups]), go["a_mean"], rtol=1e-11)
    np.testing.assert_allclose(np.stack([m["1d_moments"][k][2] for k in groups]), go["a_res_var"], rtol=1e-8, equal_nan=True)
    cov, trt = _design(memento, adata, g)
    np.random.seed(41)
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=200, num_cpus=1, verbose=0, resampling="bootstrap", approx=True,
                          strict=True)
    for k in ["mean_coef", "mean_se", "var_coef", "var_se"]:
        np.testing.assert_allclose(m["1d_ht"][k], go["a_ht_" + k], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg=k)
    for k in ["mean_asl", "var_asl"]:
        np.testing.assert_allclose(m["1d_ht"][k], go["a_ht_" + k], rtol=1e-5, atol=1e-12, equal_nan=True, err_msg=k)
    # (b) filter_genes=False: nothing is dropped, masks and fit as in the reference
    adata = _adata_from_golden(g)
    memento.setup_memento(adata, q_column="q")
    memento.create_groups(adata, label_columns=["cond", "rep"])
    memento.compute_1d_moments(adata, min_perc_group=0.7, filter_genes=False)
    m = adata.uns["memento"]
    assert adata.shape[1] == int(go["b_n_vars"]) and m["gene_list"] == list(go["b_gene_list"])
    np.testing.assert_allclose(np.stack([m["1d_moments"][k][0] for k in groups]), go["b_mean"], rtol=1e-11)
    np.testing.assert_allclose(np.stack([m["1d_moments"][k][1] for k in groups]), go["b_var"], rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(np.stack([m["1d_moments"][k][2] for k in groups]), go["b_res_var"], rtol=1e-8, equal_nan=True)
    np.testing.assert_allclose(m["mv_regressor"]["all"], go["b_mv_regressor"], rtol=1e-8)
    np.testing.assert_array_equal(np.stack([m["gene_rv_filter"][k] for k in groups]), go["b_gene_rv_filter"])
    # (c) gene_list: a further subset of the kept genes (unknown names are ignored), then the test on that subset
    adata = _adata_from_golden(g)
    memento.setup_memento(adata, q_column="q")
    memento.create_groups(adata, label_columns=["cond", "rep"])
    memento.compute_1d_moments(adata, min_perc_group=0.7, gene_list=[str(x) for x in go["c_chosen"]])
    m = adata.uns["memento"]
    assert list(adata.var.index) == list(go["c_var_names"])
    np.testing.assert_allclose(np.stack([m["1d_moments"][k][0] for k in groups]), go["c_mean"], rtol=1e-11)
    np.testing.assert_allclose(np.stack([m["1d_moments"][k][2] for k in groups]), go["c_res_var"], rtol=1e-8, equal_nan=True)
    np.random.seed(43)
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=200, num_cpus=1, verbose=0, resampling="bootstrap", approx=False,
                          strict=True)
    for k in ["mean_coef", "mean_se", "var_coef", "var_se"]:
        np.testing.assert_allclose(m["1d_ht"][k], go["c_ht_" + k], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg=k)
    for k in ["mean_asl", "var_asl"]:
        np.testing.assert_allclose(m["1d_ht"][k], go["c_ht_" + k], rtol=1e-5, atol=1e-12, equal_nan=True, err_msg=k)


def test_ht_2d_is_independent_of_the_packing(api_small, monkeypatch):
    """All 1,891 gene pairs of the fixture, packed as wide resident tiles and as > 2048 single-chain tiles (many-tile regime,
    3-waves-per-SIMD variant of the 2D kernel): identical coefficients, standard errors and p-values."""
    from scrna_parameter_estimation_amd import engine

    g = api_small
    memento, adata = _run_to_moments(g)
    names = list(adata.var.index)
    pairs = [(a, b) for i, a in enumerate(names) for b in names[i + 1:]]
    memento.compute_2d_moments(adata, pairs)
    cov, trt = _design(memento, adata, g)
    res = []
    for waves, resident in ((engine.PACK_WAVES, 2048), (10 ** 7, 10 ** 9)):
        monkeypatch.setattr(engine, "PACK_WAVES", waves)
        monkeypatch.setattr(engine, "PACK_MAX_RESIDENT", resident)
        np.random.seed(5)
        memento.ht_2d_moments(adata, covariate=cov, treatment=trt, num_boot=120, num_cpus=1, verbose=0, resampling="bootstrap",
                              approx=True)
        ht = adata.uns["memento"]["2d_ht"]
        res.append((adata.uns["memento"]["_hip"].last_bootstrap2d.n_tiles, ht["corr_coef"].copy(), ht["corr_se"].copy(), ht["corr_asl"].copy()))
    assert res[0][0] <= 2048 < res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        np.testing.assert_array_equal(a, b)
    assert np.isfinite(res[0][3]).mean() > 0.9


def test_2d_resample_rep_kernels_match_reference(regress2d_rr):
    """resample_rep=True on correlation rows: mm_residualize + mm_cross_resampled fed with the reference's own np.random.choice
    draws reproduce _regress_2d(resample_rep=True) of the REAL reference (fixture regress2d_rr: coefficient, SE, p-values)."""
    import torch

    from scrna_parameter_estimation_amd import engine
    from scrna_parameter_estimation_amd.memento import asl, design

    r = regress2d_rr
    bc, cov, trt, Nc = r["boot_corr"], r["cov"], r["trt"], r["Nc"]
    ng, B = bc.shape[0], bc.shape[1] - 1
    np.random.seed(int(r["np_seed"]))                     # the two draws of hypothesis_test.py:395-398
    ra = np.random.choice(ng, size=(ng, B)); ra[:, 0] = np.arange(ng)
    ba = np.random.choice(B, (ng, B)) + 1; ba[:, 0] = 0
    bs = object.__new__(engine.Bootstrap2D)               # only the replicate rows and their shape are needed here
    bs.yc, bs.ld, bs.B, bs.ng, bs.n_q = engine.dev(bc.copy()), B + 1, B, ng, ng
    good = np.ones((1, ng), dtype=bool)
    M, tt = design.residual_parts(cov, trt, Nc, good[0])
    coef, st = bs.contract_resampled(np.arange(1), tt[:1], good, np.zeros(1, np.int32), M[None], Nc, rep=ra[None].astype(np.int16),
                                     bcol=ba[None].astype(np.int32))
    np.testing.assert_allclose(st[0, 0], r["coef_exact"][0], rtol=1e-9)
    np.testing.assert_allclose(st[0, 1], r["se_exact"][0], rtol=1e-7)
    rows = engine.host(coef)
    for tag, approx in (("exact", False), ("approx", True)):
        p = asl.asl_from_stats(st, approx, lambda idx: rows[idx], num_cpus=1)
        np.testing.assert_allclose(p[0], r[f"asl_{tag}"][0], rtol=1e-5)


def test_ht_2d_resample_rep_api(api_small):
    """ht_2d_moments(resample_rep=True): assignments are drawn on the device, so coefficients equal the plain run's exactly and
    the standard errors / p-values are those of a different (hierarchical) null -- same order of magnitude, finite."""
    g = api_small
    memento, adata = _run_to_moments(g)
    names = np.asarray(adata.var.index)
    pairs = list(zip(names[g["pair_idx1"]].tolist(), names[g["pair_idx2"]].tolist()))
    memento.compute_2d_moments(adata, pairs)
    cov, trt = _design(memento, adata, g)
    res = {}
    for rr in (False, True):
        np.random.seed(11)
        memento.ht_2d_moments(adata, covariate=cov, treatment=trt, num_boot=300, num_cpus=1, verbose=0, resampling="bootstrap",
                              approx=True, resample_rep=rr)
        ht = adata.uns["memento"]["2d_ht"]
        res[rr] = (ht["corr_coef"].copy(), ht["corr_se"].copy(), ht["corr_asl"].copy())
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=1e-9, equal_nan=True)
    fin = np.isfinite(res[False][1])
    assert np.isfinite(res[True][1][fin]).all() and np.isfinite(res[True][2][fin]).all()
    ratio = res[True][1][fin] / res[False][1][fin]
    assert 0.3 < np.median(ratio) < 3.0


def test_ht_1d_fast_fill_statistically_equivalent(api_small):
    """strict=False: identical multinomial replay, but invalid replicates are refilled on the device with a
    counter-based RNG -> observed coefficients identical, SEs/p-values agree within Monte-Carlo error and
    exactly for genes without invalid replicates."""
    g = api_small
    memento, adata = _run_to_moments(g)
    cov, trt = _design(memento, adata, g)
    np.random.seed(int(g["ht_seed"]))
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=False)
    ht = adata.uns["memento"]["1d_ht"]
    np.testing.assert_allclose(ht["mean_coef"], g["ht_mean_coef"], rtol=1e-8, equal_nan=True)
    np.testing.assert_allclose(ht["var_coef"], g["ht_var_coef"], rtol=1e-8, equal_nan=True)
    ok = np.isfinite(g["ht_mean_se"])
    assert np.median(np.abs(ht["mean_se"][ok] / g["ht_mean_se"][ok] - 1)) < 0.1
    assert np.median(np.abs(ht["var_se"][ok] / g["ht_var_se"][ok] - 1)) < 0.15
    # gene-chunked execution (bounded replicate buffers) consumes the same stream in the same order
    whole = {k: ht[k].copy() for k in ("mean_coef", "mean_se", "var_coef", "mean_asl")}
    np.random.seed(int(g["ht_seed"]))
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=False, max_rows=40)
    ht = adata.uns["memento"]["1d_ht"]
    np.testing.assert_allclose(ht["mean_coef"], whole["mean_coef"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(ht["var_coef"], whole["var_coef"], rtol=1e-12, equal_nan=True)
    same = ht["mean_se"] == whole["mean_se"]
    assert same.mean() > 0.5            # genes without refilled replicates are bit-identical; the rest differ only by refill draws
    np.testing.assert_allclose(ht["mean_se"], whole["mean_se"], rtol=0.2, equal_nan=True)


def test_ht_1d_resample_rep_4_groups_degenerate_columns_are_a_deliberate_deviation(api_small):
    """resample_rep=True with only 2 x 2 groups: 12.5 % of the resampled columns draw groups of ONE treatment value, where
    the slope is 0/0.  The reference reports NaN or O(1) round-off noise for them (a ratio of two rounding residues, one
    of them out of LAPACK's least-squares residuals -- not reproducible bit for bit); the kernel always reports NaN.
    DELIBERATE DEVIATION: observed coefficients match the real reference exactly, everything matches
    the oracle run with its non-reference ``drop_degenerate`` switch (same stream), and the reference's SEs -- which include
    that noise -- agree within a few percent.  The noise-free case (16 groups) is pinned tightly in
    test_ht_1d_resample_rep_16_groups_matches_reference."""
    from conftest import golden_inputs
    from oracle import memento_oracle as orc

    g = api_small
    memento, adata = _run_to_moments(g)
    cov, trt = _design(memento, adata, g)
    np.random.seed(int(g["ht_seed"]) + 2)
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=False, resample_rep=True, strict=True)
    ht = {k: v.copy() for k, v in adata.uns["memento"]["1d_ht"].items() if k.endswith(("coef", "se", "asl"))}
    np.testing.assert_allclose(ht["mean_coef"], g["htrr_mean_coef"], rtol=1e-8, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(ht["var_coef"], g["htrr_var_coef"], rtol=1e-8, atol=1e-12, equal_nan=True)
    ok = np.isfinite(g["htrr_mean_se"])
    assert np.percentile(np.abs(ht["mean_se"][ok] / g["htrr_mean_se"][ok] - 1), 90) < 0.12
    X, gid, ng, q = golden_inputs(g)
    mom = dict(mean=g["mean"], res_var=g["res_var"], mv_fit=g["mv_regressor"])
    np.random.seed(int(g["ht_seed"]) + 2)
    want = orc.ht_1d(X[:, g["overall_gene_filter"]], gid, ng, g["approx_sf"], mom, g["covariate"], g["treatment"],
                     int(g["num_boot"]), g["group_q"], resampling="bootstrap", approx=False, resample_rep=True, drop_degenerate=True)
    for k, w in zip(["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"], want):
        np.testing.assert_allclose(ht[k], w, rtol=1e-5 if k.endswith("asl") else 1e-8, atol=1e-12, equal_nan=True, err_msg=k)
    # device-drawn assignments: same observed coefficients, SEs within Monte-Carlo error
    np.random.seed(int(g["ht_seed"]) + 2)
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=True, resample_rep=True)
    ht2 = adata.uns["memento"]["1d_ht"]
    np.testing.assert_allclose(ht2["mean_coef"], g["htrr_mean_coef"], rtol=1e-8, equal_nan=True)
    assert np.median(np.abs(ht2["mean_se"][ok] / g["htrr_mean_se"][ok] - 1)) < 0.15


def test_ht_1d_fast_rng_statistically_equivalent(api_small):
    """rng='fast': own RNG streams, replicate-parallel kernel.  Observed coefficients are RNG-independent (exact);
    standard errors agree with the reference's within Monte-Carlo error; p-values are strongly concordant."""
    g = api_small
    memento, adata = _run_to_moments(g)
    cov, trt = _design(memento, adata, g)
    np.random.seed(int(g["ht_seed"]))
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=2000, num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=True, rng="fast", fill_seed=7)
    ht = adata.uns["memento"]["1d_ht"]
    np.testing.assert_allclose(ht["mean_coef"], g["ht_mean_coef"], rtol=1e-8, equal_nan=True)
    np.testing.assert_allclose(ht["var_coef"], g["ht_var_coef"], rtol=1e-8, equal_nan=True)
    ok = np.isfinite(g["ht_mean_se"])
    rel = ht["mean_se"][ok] / g["ht_mean_se"][ok] - 1       # reference used B=300: ~4-5% MC error on an SE
    assert abs(np.median(rel)) < 0.03 and np.percentile(np.abs(rel), 90) < 0.15
    rel = ht["var_se"][ok] / g["ht_var_se"][ok] - 1
    assert abs(np.median(rel)) < 0.05 and np.percentile(np.abs(rel), 90) < 0.25


def test_2d_moments_ht_and_corr_matrix_match_reference(api_small):
    """compute_2d_moments / get_corr_matrix / ht_2d_moments against the real reference's outputs (the pair list
    holds a self pair and a duplicated unordered pair)."""
    g = api_small
    memento, adata = _run_to_moments(g)
    names = np.asarray(adata.var.index)
    pairs = list(zip(names[g["pair_idx1"]].tolist(), names[g["pair_idx2"]].tolist()))
    memento.compute_2d_moments(adata, pairs)
    m = adata.uns["memento"]
    for i, k in enumerate(m["groups"]):
        np.testing.assert_allclose(m["2d_moments"][k]["cov"], g["cov2d"][i], rtol=1e-8, atol=1e-14)
        np.testing.assert_allclose(m["2d_moments"][k]["corr"], g["corr2d"][i], rtol=1e-8, equal_nan=True)
    cov, trt = _design(memento, adata, g)
    np.random.seed(int(g["ht_seed"]) + 1)
    memento.ht_2d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=False)
    ht = m["2d_ht"]
    np.testing.assert_allclose(ht["corr_coef"], g["ht2_corr_coef"], rtol=1e-8, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(ht["corr_se"], g["ht2_corr_se"], rtol=1e-8, equal_nan=True)
    np.testing.assert_allclose(ht["corr_asl"], g["ht2_corr_asl"], rtol=1e-5, equal_nan=True)
    np.random.seed(int(g["ht_seed"]) + 1)                       # pair-chunked execution: identical results
    memento.ht_2d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=False, max_rows=12)
    np.testing.assert_allclose(m["2d_ht"]["corr_coef"], g["ht2_corr_coef"], rtol=1e-8, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(m["2d_ht"]["corr_asl"], g["ht2_corr_asl"], rtol=1e-5, equal_nan=True)
    ht = m["2d_ht"]
    df = memento.get_2d_ht_result(adata)
    assert list(df.columns) == ["gene_1", "gene_2", "corr_coef", "corr_se", "corr_pval"]
    g2 = memento.get_2d_moments(adata, groupby="cond")
    np.testing.assert_allclose(g2[list(g["groupby2d_cols"])].values, g["groupby2d"], rtol=1e-8, equal_nan=True)
    cm = memento.get_corr_matrix(adata, m["groups"][0])
    np.testing.assert_allclose(cm, g["corr_matrix_g0"], rtol=1e-8, atol=1e-12, equal_nan=True)


def test_ht_1d_rich_design_and_treatment_for_gene(api_approx):
    """Numeric covariate + two treatment columns, then per-gene treatment subsets (main.py:368-373, :389, :402):
    strict replay against the real reference, result order gene-major x treatment."""
    g = api_approx
    memento, adata = _run_to_moments(g)
    gdf = memento.get_groups(adata)
    cov = pd.DataFrame(g["cov2"], index=gdf.index, columns=["intercept", "rep"])
    trt = pd.DataFrame(g["trt2"], index=gdf.index, columns=["cond", "dose"])
    np.random.seed(int(g["ht_seed"]) + 4)
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=True, strict=True)
    ht = adata.uns["memento"]["1d_ht"]
    assert len(ht["mean_coef"]) == 2 * len(g["gene_list"])
    for k in ["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"]:
        np.testing.assert_allclose(ht[k], g["ht2t_" + k], rtol=1e-5 if k.endswith("asl") else 1e-8, atol=1e-12, equal_nan=True, err_msg=k)
    names = list(adata.var.index)
    tfg = {n: (["cond"] if i % 3 == 0 else (["dose"] if i % 3 == 1 else ["cond", "dose"])) for i, n in enumerate(names)}
    np.random.seed(int(g["ht_seed"]) + 5)
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, treatment_for_gene=tfg, num_boot=int(g["num_boot"]), num_cpus=1,
                          verbose=0, resampling="bootstrap", approx=True, strict=True)
    ht = adata.uns["memento"]["1d_ht"]
    for k in ["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"]:
        np.testing.assert_allclose(ht[k], g["httfg_" + k], rtol=1e-5 if k.endswith("asl") else 1e-8, atol=1e-12, equal_nan=True, err_msg=k)
    df = memento.get_1d_ht_result(adata)
    assert len(df) == len(g["httfg_mean_coef"]) and list(df["tx"][:4]) == ["cond", "dose", "cond", "dose"]


def test_ht_1d_vs_control_equals_two_group_regression(api_small):
    """Batched guide-vs-control contrasts: each contrast equals the reference's two-group regression
    (_regress_1d with covariate = intercept, treatment = guide indicator) applied to the same replicate rows."""
    from oracle import memento_oracle as orc
    from scrna_parameter_estimation_amd import engine

    g = api_small
    memento, adata = _run_to_moments(g)
    m = adata.uns["memento"]
    ng = len(m["groups"])
    np.random.seed(3)
    df = memento.ht_1d_vs_control(adata, control=m["groups"][1], num_boot=250, num_cpus=1, approx=True, max_rows=80)
    assert len(df) == len(g["gene_list"]) * (ng - 1) and list(df.columns[:2]) == ["gene", "group"]
    # recompute on the last gene chunk straight from the resident replicate rows
    np.random.seed(3)
    df1 = memento.ht_1d_vs_control(adata, control=1, num_boot=250, num_cpus=1, approx=True)       # one chunk
    np.testing.assert_allclose(df1["de_coef"].values, df["de_coef"].values, rtol=1e-12, equal_nan=True)
    # (one chunk: all genes are still resident)
    st = m["_hip"]
    np.random.seed(3)
    bs = engine.Bootstrap1D(st.blocks, st.gene_idx, st.maxx, st.sf_bin, st.sf_table, np.array([m["group_q"][k] for k in m["groups"]]), 250)
    tm = np.stack([m["1d_moments"][k][0] for k in m["groups"]]).T.reshape(-1)
    tv = np.stack([m["1d_moments"][k][2] for k in m["groups"]]).T.reshape(-1)
    skip = ~(np.isfinite(np.log(tm)) & np.isfinite(np.log(tv)))
    bs.alloc_outputs(np.log(tm), np.log(tv))
    u = np.random.random(2 * int((~skip).sum()))
    r1, r0 = np.zeros(bs.n_pairs), np.zeros(bs.n_pairs)
    r1[~skip], r0[~skip] = u[0::2], u[1::2]
    bs.run(skip, r1, r0, m["mv_regressor"]["all"], fill_mode=0, fill_seed=0)
    ym, yv = engine.host(bs.ym), engine.host(bs.yv)
    Nc = np.array([m["group_cells"][k].shape[0] for k in m["groups"]], dtype=float)
    others = [j for j in range(ng) if j != 1]
    for gi in range(0, len(g["gene_list"]), 7):
        for oi, j in enumerate(others):
            rows = [gi * ng + 1, gi * ng + j]
            ref = orc.regress_1d(np.ones((2, 1)), np.array([[0.0], [1.0]]), ym[rows], yv[rows], Nc[[1, j]], resampling="bootstrap", approx=True)
            r = df1.iloc[gi * (ng - 1) + oi]
            np.testing.assert_allclose([r.de_coef, r.de_se, r.de_pval, r.dv_coef, r.dv_se, r.dv_pval],
                                       [ref[0][0], ref[1][0], ref[2][0], ref[3][0], ref[4][0], ref[5][0]], rtol=1e-7, atol=1e-12)


def test_inplace_false_and_prepare_to_save(api_small):
    """inplace=False returns a copy and leaves the input untouched; prepare_to_save drops everything that cannot be
    written to disk (device handles, per-group regressors) -- reference: main.py:39-40, :673-683."""
    g = api_small
    from scrna_parameter_estimation_amd import memento

    adata = _adata_from_golden(g)
    memento.setup_memento(adata, q_column="q")
    memento.create_groups(adata, label_columns=["cond", "rep"])
    before = adata.shape
    out = memento.compute_1d_moments(adata, min_perc_group=0.7, inplace=False)
    assert adata.shape == before and "1d_moments" not in adata.uns["memento"]
    assert out.shape[1] == len(g["gene_list"]) and "1d_moments" in out.uns["memento"]
    np.testing.assert_allclose(out.uns["memento"]["1d_moments"][g["groups"][0]][0], g["mean"][0], rtol=1e-11)
    memento.prepare_to_save(out)
    m = out.uns["memento"]
    assert "_hip" not in m and all(k not in m["mv_regressor"] for k in list(g["groups"]) + ["all"])
    import pickle

    pickle.dumps({k: v for k, v in m.items() if k not in ("1d_ht",)})          # plain python / numpy only


def test_missing_library_fails_loudly(monkeypatch):
    from scrna_parameter_estimation_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmemento_hip.so")
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.load()


def test_corr_matrix_with_nonpositive_variances_matches_reference(corrmat_negvar):
    """get_corr_matrix on groups holding genes with variance estimates <= 0, against the real reference
    (estimator.py:259-268): same NaN mask, same finite values (1e-8), and no write into uns['memento']['1d_moments']."""
    g = corrmat_negvar
    memento, adata = _run_to_moments(g)
    m = adata.uns["memento"]
    assert m["gene_list"] == list(g["gene_list"])
    for k, grp in enumerate(m["groups"]):
        before = m["1d_moments"][grp][1].copy()
        assert (before <= 0).any()
        cm = memento.get_corr_matrix(adata, grp)
        np.testing.assert_array_equal(before, m["1d_moments"][grp][1])
        want = g[f"corr_matrix_{k}"]
        np.testing.assert_array_equal(np.isnan(cm), np.isnan(want))
        np.testing.assert_allclose(cm, want, rtol=1e-8, atol=1e-12, equal_nan=True)


def _rr16_design(memento, adata, g):
    gdf = memento.get_groups(adata)
    cov = pd.DataFrame(g["covariate"], index=gdf.index, columns=["intercept", "rep"])
    trt = pd.DataFrame(g["treatment"], index=gdf.index, columns=["cond"])
    return cov, trt


@pytest.mark.parametrize("tag,approx", [("exact", False), ("approx", True)])
def test_ht_1d_resample_rep_16_groups_matches_reference(api_rr16, tag, approx):
    """resample_rep=True, 2 x 8 groups, intercept + numeric covariate: no resampled column is degenerate, so the strict
    replay must reproduce the REAL reference: coefficients and standard errors to 1e-8, p-values to 1e-5."""
    g = api_rr16
    memento, adata = _run_to_moments(g)
    assert list(adata.var.index) == list(g["gene_list"])
    cov, trt = _rr16_design(memento, adata, g)
    np.random.seed(int(g[f"seed_{tag}"]))
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=approx, resample_rep=True, strict=True)
    ht = adata.uns["memento"]["1d_ht"]
    for k in ["mean_coef", "mean_se", "var_coef", "var_se"]:
        np.testing.assert_allclose(ht[k], g[f"htrr_{tag}_{k}"], rtol=1e-8, atol=1e-12, err_msg=k)
    for k in ["mean_asl", "var_asl"]:
        np.testing.assert_allclose(ht[k], g[f"htrr_{tag}_{k}"], rtol=1e-5, atol=1e-12, err_msg=k)


def test_ht_2d_resample_rep_16_groups_matches_reference(api_rr16):
    """ht_2d_moments(resample_rep=True, strict=True): the two np.random.choice draws of _regress_2d
    (hypothesis_test.py:395-398) are replayed from the global stream in pair order -- real-reference fixture, 1e-8 / 1e-5."""
    g = api_rr16
    memento, adata = _run_to_moments(g)
    names = np.asarray(adata.var.index)
    pairs = list(zip(names[g["pair_idx1"]].tolist(), names[g["pair_idx2"]].tolist()))
    memento.compute_2d_moments(adata, pairs)
    m = adata.uns["memento"]
    for i, k in enumerate(m["groups"]):
        np.testing.assert_allclose(m["2d_moments"][k]["corr"], g["true_corr"][i], rtol=1e-8, equal_nan=True)
    cov, trt = _rr16_design(memento, adata, g)
    for max_rows in (None, 64):                  # whole pair list at once / 4 pairs per chunk: same stream order
        np.random.seed(int(g["seed_2d"]))
        memento.ht_2d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0,
                              resampling="bootstrap", approx=False, resample_rep=True, strict=True, max_rows=max_rows)
        ht = m["2d_ht"]
        np.testing.assert_allclose(ht["corr_coef"], g["ht2rr_corr_coef"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(ht["corr_se"], g["ht2rr_corr_se"], rtol=1e-8)
        np.testing.assert_allclose(ht["corr_asl"], g["ht2rr_corr_asl"], rtol=1e-5)


def test_ht_1d_resample_rep_one_group_per_donor():
    """The reference's real use of resample_rep (analysis/lupus/run_memento.py:31-52): ONE GROUP PER DONOR (120 donors here,
    far beyond the 64 groups round 1 stopped at), genotype-like numeric treatments chosen per gene (treatment_for_gene),
    donor covariates -- strict replay against the oracle on the same stream."""
    from oracle import memento_oracle as orc
    from scrna_parameter_estimation_amd import AnnDataLite, memento
    from scrna_parameter_estimation_amd.synth import synth_counts

    n_donors, per, G, B = 120, 70, 40, 150
    rng = np.random.default_rng(5)
    X = synth_counts(n_donors * per, G, 0.25, seed=77, dtype=np.float32)
    donor = rng.permutation(np.repeat(np.arange(n_donors), per))
    obs = pd.DataFrame({"ind": [f"d{d:03d}" for d in donor], "q": np.full(len(donor), 0.1)}, index=[f"c{i}" for i in range(len(donor))])
    adata = AnnDataLite(X, obs, pd.DataFrame(index=[f"g{i}" for i in range(G)]))
    Xref = X.astype(np.float64).copy()
    memento.setup_memento(adata, q_column="q", filter_mean_thresh=0.01)
    memento.create_groups(adata, label_columns=["ind"])
    memento.compute_1d_moments(adata, min_perc_group=0.9)
    m = adata.uns["memento"]
    groups = m["groups"]
    ng = len(groups)
    assert ng == n_donors
    names = list(adata.var.index)
    assert len(names) >= 5
    geno = pd.DataFrame(rng.integers(0, 3, size=(ng, 3)).astype(float), index=groups, columns=["snp0", "snp1", "snp2"])
    covd = pd.DataFrame({"intercept": np.ones(ng), "age": rng.normal(size=ng), "sex": rng.integers(0, 2, size=ng).astype(float)}, index=groups)
    tfg = {n: (["snp0", "snp2"] if i % 2 else ["snp1"]) for i, n in enumerate(names)}
    np.random.seed(9)
    memento.ht_1d_moments(adata, covariate=covd, treatment=geno, treatment_for_gene=tfg, num_boot=B, num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=True, resample_rep=True, strict=True)
    ht = {k: v.copy() for k, v in m["1d_ht"].items() if k.endswith(("coef", "se", "asl"))}
    # oracle: same inputs, same global stream, gene by gene with the gene's own treatment columns
    gid = m["_hip"].group_id
    sf = adata.obs["memento_size_factor"].values
    gq = np.array([m["group_q"][k] for k in groups])
    keep = np.isin([f"g{i}" for i in range(G)], names)
    mom = dict(mean=np.stack([m["1d_moments"][k][0] for k in groups]), res_var=np.stack([m["1d_moments"][k][2] for k in groups]),
               mv_fit=m["mv_regressor"][groups[0]])
    Xk = sp.csc_matrix(Xref[:, keep])
    sel = [np.flatnonzero(gid == j) for j in range(ng)]
    Nc = np.array([len(s_) for s_ in sel], dtype=float)
    asf = [m["all_approx_size_factor"][s_] for s_ in sel]
    np.random.seed(9)
    want = [[] for _ in range(6)]
    for gi, n in enumerate(names):
        col = np.asarray(Xk[:, gi].todense()).ravel()
        res = orc.ht_1d_gene(mom["mean"][:, gi], mom["res_var"][:, gi], [col[s_] for s_ in sel], asf, covd.values, geno[tfg[n]].values,
                             Nc, B, mom["mv_fit"], gq, resampling="bootstrap", approx=True, resample_rep=True)
        for o, r in zip(want, res):
            o.append(np.atleast_1d(r) * np.ones(len(tfg[n])))
    for k, w in zip(["mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl"], want):
        w = np.concatenate(w)
        assert np.isfinite(w).all()
        np.testing.assert_allclose(ht[k], w, rtol=1e-5 if k.endswith("asl") else 1e-8, atol=1e-12, err_msg=k)
    # device-drawn assignments (the default, strict=False): same observed coefficients, SEs within Monte-Carlo error
    np.random.seed(9)
    memento.ht_1d_moments(adata, covariate=covd, treatment=geno, treatment_for_gene=tfg, num_boot=B, num_cpus=1, verbose=0,
                          resampling="bootstrap", approx=True, resample_rep=True)
    np.testing.assert_allclose(m["1d_ht"]["mean_coef"], ht["mean_coef"], rtol=1e-8)
    assert np.median(np.abs(m["1d_ht"]["mean_se"] / ht["mean_se"] - 1)) < 0.2


def test_resample_rep_drops_non_finite_replicate_columns():
    """hypothesis_test.py:249-254: replicate columns with a non-finite entry (in the mean OR the variability rows of any group)
    are dropped BEFORE the hierarchical resampling, which then draws among the survivors.  mm_valid_cols + mm_residualize +
    mm_cross_resampled fed with the reference's np.random.choice draws against the oracle's _regress_1d restatement."""
    from oracle import memento_oracle as orc
    from scrna_parameter_estimation_amd import engine
    from scrna_parameter_estimation_amd.memento import design

    rng = np.random.default_rng(3)
    ng, B = 10, 120
    cov = np.column_stack([np.ones(ng), rng.normal(size=ng)])
    trt = (np.arange(ng) % 2).astype(float).reshape(-1, 1)
    Nc = rng.integers(200, 900, size=ng).astype(float)
    bm = 0.1 * rng.normal(size=(ng, B + 1)) + 0.3 * trt
    bv = 0.2 * rng.normal(size=(ng, B + 1))
    bm[3, 17] = np.nan
    bv[0, 40] = np.inf
    bv[7, 41] = np.nan
    bm[2, B] = -np.inf
    np.random.seed(123)
    want = orc.regress_1d(cov, trt, bm, bv, Nc, resampling="bootstrap", approx=True, resample_rep=True)
    nb = B - 4
    np.random.seed(123)
    ra = np.random.choice(ng, size=(ng, nb)); ra[:, 0] = np.arange(ng)
    ba = np.random.choice(nb, (ng, nb)) + 1; ba[:, 0] = 0
    rep = np.zeros((1, ng, B), dtype=np.int16); rep[0, :, :nb] = ra
    bcol = np.zeros((1, ng, B), dtype=np.int32); bcol[0, :, :nb] = ba
    bs = object.__new__(engine.Bootstrap1D)
    bs.ym, bs.yv, bs.ld, bs.B, bs.ng, bs.n_tested = engine.dev(bm), engine.dev(bv), B + 1, B, ng, 1
    good = np.ones((1, ng), dtype=bool)
    col_map, n_valid = bs.valid_cols(good)
    assert int(n_valid[0]) == B + 1 - 4
    cm = engine.host(col_map)[0, : n_valid[0]]
    np.testing.assert_array_equal(cm, np.setdiff1d(np.arange(B + 1), [17, 40, 41, B]))
    M, tt = design.residual_parts(cov, trt, Nc, good[0])
    for which, (c_w, se_w) in enumerate(((want[0], want[1]), (want[3], want[4]))):
        coef, st = bs.contract_resampled(np.arange(1), tt[:1], good, which, np.zeros(1, np.int32), M[None], Nc, rep=rep, bcol=bcol,
                                         col_map=col_map, n_valid=n_valid)
        np.testing.assert_allclose(st[0, 0], c_w[0], rtol=1e-9)
        np.testing.assert_allclose(st[0, 1], se_w[0], rtol=1e-8)
        assert int(st[0, 2]) == nb - 1


def test_c1_pbmc3k_shape_matches_reference(api_c1):
    """BASELINE.json configs[0]: 2.7k cells x 1.8k genes, 2 groups, 100 bootstraps -- the full API in strict replay against
    the real reference's output (size factors 1e-12, gene mask bit-exact, coefficients / SE 1e-8, p-values 1e-5)."""
    from conftest import c1_inputs
    from scrna_parameter_estimation_amd import AnnDataLite, memento

    g = api_c1
    X, gid, ng, q = c1_inputs(g)
    obs = pd.DataFrame({"cond": g["in_cond"].astype(np.int64), "q": np.full(X.shape[0], q)}, index=[f"c{i}" for i in range(X.shape[0])])
    adata = AnnDataLite(X.astype(np.float32), obs, pd.DataFrame(index=[f"g{i}" for i in range(X.shape[1])]))
    memento.setup_memento(adata, q_column="q")
    np.testing.assert_allclose(adata.obs["memento_size_factor"].values, g["size_factor"], rtol=1e-12)
    memento.create_groups(adata, label_columns=["cond"])
    memento.compute_1d_moments(adata, min_perc_group=0.7)
    m = adata.uns["memento"]
    assert m["groups"] == list(g["groups"])
    np.testing.assert_array_equal(m["overall_gene_filter"], g["overall_gene_filter"])
    for i, k in enumerate(m["groups"]):
        np.testing.assert_allclose(m["1d_moments"][k][0], g["mean"][i], rtol=1e-11)
        np.testing.assert_allclose(m["1d_moments"][k][1], g["var"][i], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(m["1d_moments"][k][2], g["res_var"][i], rtol=1e-8, atol=1e-13, equal_nan=True)
    gdf = memento.get_groups(adata)
    cov = pd.DataFrame(g["covariate"], index=gdf.index, columns=["intercept"])
    trt = pd.DataFrame(g["treatment"], index=gdf.index, columns=["cond"])
    np.random.seed(71)
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=100, num_cpus=1, verbose=0, resampling="bootstrap", approx=False,
                          strict=True)
    ht = m["1d_ht"]
    assert len(ht["mean_coef"]) == int(g["overall_gene_filter"].sum()) == 819
    for k in ["mean_coef", "mean_se", "var_coef", "var_se"]:
        np.testing.assert_allclose(ht[k], g["ht_" + k], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg=k)
    for k in ["mean_asl", "var_asl"]:
        np.testing.assert_allclose(ht[k], g["ht_" + k], rtol=1e-5, atol=1e-12, equal_nan=True, err_msg=k)
