"""Edge cases on the GPU path: empty genes / cells, a gene expressed in one group only, tiny groups, cells outside
every group, single-bin pairs, and the host ordering fallback for pairs with more bins than the LDS sort holds."""

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from scrna_parameter_estimation_amd import engine

    engine._lib.load(require_gpu=True)
    return engine


def _edge_matrix():
    rng = np.random.default_rng(42)
    n, g = 700, 40
    X = rng.poisson(0.6, size=(n, g)).astype(np.float32)
    X[:, 3] = 0                      # gene never expressed
    gid = rng.integers(0, 3, size=n).astype(np.int32)
    gid[:6] = 3                      # a 6-cell group
    gid[6:20] = -1                   # cells in no group
    X[gid == 1, 5] = 0               # gene 5 silent in group 1
    X[gid != 0, 7] = 0               # gene 7 expressed in group 0 only
    X[30:40, :] = 0                  # empty cells
    X[50, 9] = 300                   # one large count
    return sp.csr_matrix(X), gid, 4


def test_edge_moments_and_bins(eng):
    from oracle import memento_oracle as orc

    X, gid, ng = _edge_matrix()
    rng = np.random.default_rng(1)
    sf = rng.lognormal(0, 0.4, size=X.shape[0])
    blocks = eng.CountBlocks(eng.DeviceCSR(X), gid, ng)
    S, sumx, maxx = blocks.moments(1.0 / sf)
    X64 = X.astype(np.float64).tocsc()
    for k in range(ng):
        sel = np.flatnonzero(gid == k)
        w = 1.0 / sf[sel]
        np.testing.assert_allclose(S[0, k], X64[sel].T.dot(w), rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(S[1, k], X64[sel].power(2).T.dot(w ** 2), rtol=1e-12, atol=1e-300)
        np.testing.assert_array_equal(sumx[k], np.asarray(X64[sel].sum(axis=0)).ravel().astype(np.uint64))
        np.testing.assert_array_equal(maxx[k], np.asarray(X64[sel].max(axis=0).todense()).ravel().astype(np.uint32))
    assert maxx[:, 3].max() == 0 and sumx[1, 5] == 0 and sumx[1:, 7].sum() == 0 and maxx.max() == 300
    # bins: 3 size-factor bins, every gene tested
    edges = np.quantile(sf, [1 / 3, 2 / 3])
    sf_bin = np.digitize(sf, edges).astype(np.uint8)
    sf_table = np.array([sf[sf_bin == b].mean() for b in range(3)])
    bs = eng.Bootstrap1D(blocks, np.arange(X.shape[1]), maxx, sf_bin, sf_table, np.full(ng, 0.1), 32)
    Xd = X.toarray()
    for gene in (3, 5, 7, 9, 0):
        for k in range(ng):
            sel = np.flatnonzero(gid == k)
            _, _, expr, mult = orc.unique_bins_1d(Xd[sel, gene].astype(np.float64), sf_table[sf_bin[sel]], 0.3, 0.7)
            bi, xi, mu = bs.bins_of_pair(gene * ng + k)
            assert sorted(zip(xi.tolist(), mu.tolist())) == sorted(zip(expr.astype(int).tolist(), mult.tolist()))
            assert mu.sum() == len(sel) and bs.K[gene * ng + k] == len(mult)


def test_replay_weights_edge_and_host_order_fallback(eng, monkeypatch):
    from oracle import memento_oracle as orc

    X, gid, ng = _edge_matrix()
    rng = np.random.default_rng(2)
    sf = rng.lognormal(0, 0.4, size=X.shape[0])
    blocks = eng.CountBlocks(eng.DeviceCSR(X), gid, ng)
    S, sumx, maxx = blocks.moments(1.0 / sf)
    sf_bin = np.digitize(sf, np.quantile(sf, [0.25, 0.5, 0.75])).astype(np.uint8)
    sf_table = np.array([sf[sf_bin == b].mean() for b in range(4)])
    B = 24
    Xd = X.toarray().astype(np.float64)
    results = []
    for caps in ((1024, 8192), (4, 6)):     # second pass: almost every pair goes through the big / host ordering paths
        monkeypatch.setattr(eng, "ORDER_SMALL_CAP", caps[0])
        monkeypatch.setattr(eng, "ORDER_BIG_CAP", caps[1])
        bs = eng.Bootstrap1D(blocks, np.arange(X.shape[1]), maxx, sf_bin, sf_table, np.full(ng, 0.1), B)
        r = np.random.default_rng(5).random((2, bs.n_pairs))
        zeros = np.zeros(bs.n_pairs)
        bs.alloc_outputs(zeros, zeros)
        bs.run(np.zeros(bs.n_pairs, bool), r[0], r[1], [0.0, 1.0, 0.0], fill_mode=1, dump_weights=True)
        rm = eng.host(bs.raw_mean)
        for p in range(bs.n_pairs):
            gene, k = divmod(p, ng)
            sel = np.flatnonzero(gid == k)
            inv_sf, inv_sf_sq, expr, mult = orc.unique_bins_1d(Xd[sel, gene], sf_table[sf_bin[sel]], r[0][p], r[1][p])
            if len(expr) <= 1:
                assert np.isnan(rm[p, 1:]).all()          # bootstrap.py:97-98
                continue
            w = orc.multinomial_weights(len(sel), mult, B)
            np.testing.assert_array_equal(bs.weights_of(p), w, err_msg=f"pair {p} caps {caps}")
        results.append(rm)
    np.testing.assert_array_equal(results[0], results[1])  # device-ordered and host-ordered operands give identical replicates


def test_explicitly_stored_zeros_are_accepted(eng):
    """A CSR with explicitly stored zeros (common after subsetting or arithmetic on adata.X; the reference accepts any scipy CSR,
    main.py:44) gives the same count blocks and moments as its zero-free twin."""
    X, gid, ng = _edge_matrix()
    Z = X.tolil()
    Z[0, 3] = 1.0
    Z[5, 11] = 1.0
    Z = Z.tocsr()
    Z.data[(Z.indices == 3) | ((Z.indices == 11) & (np.arange(Z.nnz) < Z.indptr[6]) & (np.arange(Z.nnz) >= Z.indptr[5]))] = 0.0
    keep = X[5, 11]
    Z[5, 11] = keep if keep else 0.0
    assert (Z.data == 0).any() and (Z != X).nnz == 0
    sf = np.random.default_rng(1).lognormal(0, 0.4, size=X.shape[0])
    a = eng.CountBlocks(eng.DeviceCSR(X), gid, ng).moments(1.0 / sf)
    b = eng.CountBlocks(eng.DeviceCSR(Z), gid, ng).moments(1.0 / sf)
    for u, v in zip(a, b):
        np.testing.assert_allclose(u, v, rtol=1e-13, atol=0)


def test_2d_host_ordering_fallback_equals_device_order(api_small, monkeypatch):
    """A (pair, group) with more unique bins than the in-LDS sort holds is ordered on the host (engine.Bootstrap2D._order_on_host,
    the same arithmetic as k_bins_order2d): forcing EVERY pair through that path gives bit-identical replicate correlations."""
    import pandas as pd

    from scrna_parameter_estimation_amd import AnnDataLite, engine, memento

    g = api_small
    X = sp.csr_matrix((g["in_data"].astype(np.float32), g["in_indices"], g["in_indptr"]), shape=tuple(g["in_shape"]))
    obs = pd.DataFrame({"cond": g["in_cond"], "rep": g["in_rep"], "q": g["in_q"]}, index=[f"c{i}" for i in range(X.shape[0])])
    adata = AnnDataLite(X, obs, pd.DataFrame(index=g["in_gene_names"].tolist()))
    memento.setup_memento(adata, q_column="q")
    memento.create_groups(adata, label_columns=["cond", "rep"])
    memento.compute_1d_moments(adata, min_perc_group=0.7)
    names = np.asarray(adata.var.index)
    pairs = [(names[i], names[i + 1]) for i in range(0, 12, 2)]
    memento.compute_2d_moments(adata, pairs)
    gdf = memento.get_groups(adata)
    cov = pd.DataFrame(g["covariate"], index=gdf.index, columns=["intercept"])
    trt = pd.DataFrame(g["treatment"], index=gdf.index, columns=["cond"])
    rows = []
    for cap in (engine.ORDER_BIG_CAP_2D, 0):
        monkeypatch.setattr(engine, "ORDER_BIG_CAP_2D", cap)
        if cap == 0:
            monkeypatch.setattr(engine, "ORDER_SMALL_CAP", 0)
        np.random.seed(4)
        memento.ht_2d_moments(adata, covariate=cov, treatment=trt, num_boot=64, num_cpus=1, verbose=0, resampling="bootstrap", approx=True)
        rows.append(engine.host(adata.uns["memento"]["_hip"].last_bootstrap2d.yc).copy())
    np.testing.assert_array_equal(rows[0], rows[1])


def test_mean_filter_tie_follows_scipy_rounding(eng):
    """sum / n == filter_mean_thresh exactly (189 counts in 2,700 cells = 0.07): the reference's scipy mean is
    sum_c fl(x_c / n) in a specific order and lands a few ulp off the exact quotient; memento.main._plain_means reproduces it
    (sequential for the CSR of all cells, first + pairwise(rest) for a group's CSC copy) -- compared here with scipy itself."""
    from scrna_parameter_estimation_amd.memento.main import _plain_means

    rng = np.random.default_rng(8)
    n, G = 2700, 24
    X = np.zeros((n, G))
    for gcol in range(G):                      # every gene: exactly 189 counts, spread differently (1s, 2s, 3s) over the cells
        left = 189
        while left > 0:
            v = min(left, int(rng.integers(1, 4)))
            c = int(rng.integers(0, n))
            if X[c, gcol] == 0:
                X[c, gcol] = v
                left -= v
    Xs = sp.csr_matrix(X)
    assert (np.asarray(Xs.sum(axis=0)).ravel() == 189).all()
    blocks = eng.CountBlocks(eng.DeviceCSR(Xs.astype(np.float32)), np.zeros(n, dtype=np.int32), 1)
    _, sumx, _ = blocks.moments(np.ones(n))
    want_csr = np.asarray(Xs.mean(axis=0)).ravel()
    want_csc = np.asarray(Xs.tocsc().mean(axis=0)).ravel()
    assert len(set(want_csr.tolist())) > 1 or len(set(want_csc.tolist())) > 1      # the rounding really depends on the gene
    np.testing.assert_array_equal(_plain_means(blocks, sumx, [n], 0.07, 'csr')[0], want_csr)
    np.testing.assert_array_equal(_plain_means(blocks, sumx, [n], 0.07, 'csc')[0], want_csc)
    np.testing.assert_array_equal(_plain_means(blocks, sumx, [n], 0.05, 'csc')[0], np.full(G, 189 / 2700))    # no tie: exact quotient


def test_unsorted_rows_take_the_unpartitioned_ingest(eng):
    """The range-partitioned ingest needs ascending column indices inside a row; a device CSR whose rows are not sorted is
    detected (mm_sell_split_count) and goes through the unpartitioned kernels -- same count blocks, same moments."""
    import torch

    X, gid, ng = _edge_matrix()
    sf = np.random.default_rng(1).lognormal(0, 0.4, size=X.shape[0])
    a = eng.CountBlocks(eng.DeviceCSR(X), gid, ng)
    assert a.ranged
    idx, dat, ptr = X.indices.copy(), X.data.copy(), X.indptr
    rng = np.random.default_rng(3)
    for r in range(X.shape[0]):                       # shuffle the entries of every row
        p = rng.permutation(ptr[r + 1] - ptr[r]) + ptr[r]
        idx[ptr[r]:ptr[r + 1]], dat[ptr[r]:ptr[r + 1]] = idx[p], dat[p]
    csr = eng.DeviceCSR.from_device(torch.from_numpy(ptr.astype(np.int64)).cuda(), torch.from_numpy(idx.astype(np.int32)).cuda(),
                                    torch.from_numpy(dat.astype(np.float32)).cuda(), X.shape)
    b = eng.CountBlocks(csr, gid, ng)
    assert not b.ranged
    for u, v in zip(a.moments(1.0 / sf), b.moments(1.0 / sf)):
        np.testing.assert_allclose(u, v, rtol=1e-13, atol=0)


def test_invalid_counts_are_rejected(eng):
    X, gid, ng = _edge_matrix()
    for bad in (0.5, -1.0, 600000.0):
        Y = X.copy()
        Y.data[7] = bad
        with pytest.raises(ValueError):
            eng.CountBlocks(eng.DeviceCSR(Y) if bad != 600000.0 else eng.DeviceCSR(Y), gid, ng)


def test_ingest_is_deterministic_and_cell_ordered(eng):
    """The quad-staged scatter gives every gene's entries in ascending cell order, so two ingests of the same CSR produce
    bit-identical count blocks (round 1's per-entry atomics did not) and K1's fp64 sums are reproducible across ingests."""
    import torch

    X, gid, ng = _edge_matrix()
    sf = np.random.default_rng(1).lognormal(0, 0.4, size=X.shape[0])
    a = eng.CountBlocks(eng.DeviceCSR(X), gid, ng)
    b = eng.CountBlocks(eng.DeviceCSR(X), gid, ng)
    assert a.ranged and torch.equal(a.ent, b.ent)
    Sa, Sb = a.moments(1.0 / sf), b.moments(1.0 / sf)
    assert all(np.array_equal(u, v) for u, v in zip(Sa, Sb))
    # inside every (block, gene): cell_local strictly ascending over the non-padding entries
    ent = eng.host(a.ent, np.uint32).reshape(-1, 64, 4)
    sp_, sw, base, perm = eng.host(a.slice_ptr), eng.host(a.slice_w), eng.host(a.blk_base), eng.host(a.perm)
    for blk in range(a.n_blocks):
        for t in range(a.n_slices):
            rows = ent[base[blk] + sp_[blk, t]: base[blk] + sp_[blk, t] + sw[blk, t]]           # [rows][lane][4]
            for ln in range(64):
                if perm[blk, t * 64 + ln] < 0:
                    continue
                e = rows[:, ln, :].reshape(-1)
                e = e[e != 0]
                assert (np.diff((e & 8191).astype(np.int64)) > 0).all()


def test_ingest_with_many_genes_and_very_sparse_rows(eng):
    """36k genes, ~12 non-zeros per row: the number of gene ranges is set by the LDS budget of a range, not by the row length."""
    rng = np.random.default_rng(5)
    n, G = 3000, 36000
    rows = np.repeat(np.arange(n), 12)
    cols = rng.integers(0, G, size=n * 12)
    X = sp.csr_matrix((np.ones(n * 12, dtype=np.float32), (rows, cols)), shape=(n, G))
    X.sum_duplicates()
    gid = rng.integers(0, 2, size=n).astype(np.int32)
    blocks = eng.CountBlocks(eng.DeviceCSR(X), gid, 2)
    assert blocks.ranged
    S, sumx, maxx = blocks.moments(np.ones(n))
    for k in range(2):
        want = np.asarray(X[gid == k].sum(axis=0)).ravel()
        np.testing.assert_array_equal(sumx[k], want.astype(np.uint64))
        np.testing.assert_array_equal(S[0, k], want)
