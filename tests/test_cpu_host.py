"""CPU-only tests (-m "not gpu"): the C-ABI library loads and exports everything include/memento_hip.h
declares (no compute calls), the host-side pieces of the product (design folding, p-values, block
planning, lane packing inputs) agree with the oracle / golden fixtures, and the multi-rank exchange works
over gloo with world_size 2."""

import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from scrna_parameter_estimation_amd import _lib, build

    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.LIB_PATH


def test_cabi_exports_every_declared_symbol(lib_path):
    hdr = open(os.path.join(ROOT, "include", "memento_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/memento_hip.h but not exported"
    # the ctypes binding table covers exactly the declared entry points
    from scrna_parameter_estimation_amd import _lib

    assert sorted(_lib.EXPORTS) == declared
    lib.mm_version.restype = ctypes.c_int
    assert lib.mm_version() == 100


def test_product_path_needs_gpu(lib_path):
    """No CPU fallback: with the library present but no HIP device the product path raises."""
    import torch

    from scrna_parameter_estimation_amd import _lib

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.MementoHipError):
        _lib.load(require_gpu=True)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "scrna_parameter_estimation_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), f"{f} mentions the oracle"


def test_design_weight_rows_match_oracle_regression():
    from oracle import memento_oracle as orc
    from scrna_parameter_estimation_amd.memento import design

    rng = np.random.default_rng(3)
    for trial in range(20):
        n, T, B = int(rng.integers(3, 12)), int(rng.integers(1, 4)), 30
        cov = np.column_stack([np.ones(n)] + [rng.normal(size=n) for _ in range(int(rng.integers(0, 2)))])
        trt = np.column_stack([(rng.random(n) < 0.5).astype(float) if k == 0 else rng.normal(size=n) for k in range(T)])
        if trt[:, 0].std() == 0:
            trt[0, 0] = 1 - trt[0, 0]
        Nc = rng.integers(50, 5000, size=n).astype(float)
        y = rng.normal(size=(n, B))
        good = rng.random(n) < 0.8
        good[:3] = True
        W = design.weight_rows(cov, trt, Nc, good)
        ref = orc.cross_coef(orc._weighted_residualize(trt[good], cov[good], Nc[good]),
                             orc._weighted_residualize(y[good], cov[good], Nc[good]), Nc[good])
        np.testing.assert_allclose(W @ y, ref, rtol=1e-9, atol=1e-12)
        assert (W[:, ~good] == 0).all()
    W1 = design.weight_rows(np.ones((5, 1)), np.ones((5, 1)), np.arange(1.0, 6.0), np.ones(5, bool))
    np.testing.assert_allclose(W1[0], np.arange(1.0, 6.0) / 15.0)


def test_asl_from_stats_matches_golden(regress_asl):
    """The host p-value logic fed with the statistics the contraction kernel would produce."""
    from scrna_parameter_estimation_amd.memento import asl

    r = regress_asl
    rows, stats, want, approx = [], [], [], []
    for tag, ap in [("count", False), ("tail", False), ("negtail", False), ("approx", True)]:
        row = r["asl_in_" + tag]
        null = row[1:] - row[0]
        a = abs(row[0])
        st = [row[0], row[1:].std(), len(null), float((null > a).sum() + (null < -a).sum()), null.mean(), 0.0, row.min(), row.max()]
        rows.append(row), stats.append(st), want.append(float(r["asl_out_" + tag])), approx.append(ap)
    rows = np.stack(rows)
    got = asl.asl_from_stats(np.array(stats[:3]), False, lambda idx: rows[idx], num_cpus=1)
    np.testing.assert_allclose(got, want[:3], rtol=1e-9)
    got = asl.asl_from_stats(np.array(stats[3:]), True, lambda idx: rows[3:][idx], num_cpus=1)
    np.testing.assert_allclose(got, want[3:], rtol=1e-9)


def test_asl_from_stats_uncentred_null_matches_oracle():
    """resampling != 'bootstrap': the null is coef[1:] itself (hypothesis_test.py:69-70); counting, normal and tail-fit
    branches against the oracle's compute_asl (itself pinned by the api_perm fixture from the real reference)."""
    sys.path.insert(0, ROOT)
    from oracle import memento_oracle as orc
    from scrna_parameter_estimation_amd.memento import asl

    rng = np.random.default_rng(4)
    rows = []
    for stat, loc in ((0.3, 0.0), (2.9, 0.0), (-3.1, 0.1), (0.0, 1.0), (1.2, -0.5)):   # many / few extreme replicates
        rows.append(np.concatenate([[stat], rng.normal(loc, 1.0, size=1500)]))
    rows = np.stack(rows)
    rows[2, 7] = np.nan                                                                # a dropped replicate column
    st = []
    for row in rows:
        v = row[1:][~np.isnan(row[1:])]
        null, a = v - row[0], abs(row[0])
        st.append([row[0], v.std(), len(v), float((null > a).sum() + (null < -a).sum()), null.mean(), 0.0,
                   float((v > a).sum() + (v < -a).sum()), np.nanmax(row) - np.nanmin(row)])
    st = np.array(st)
    for approx in (False, True):
        got = asl.asl_from_stats(st, approx, lambda idx: rows[idx], num_cpus=1, resampling="permutation")
        want = [orc.compute_asl(r[~np.isnan(r)], "permutation", approx) for r in rows]
        np.testing.assert_allclose(got, want, rtol=1e-9)
        got_b = asl.asl_from_stats(st, approx, lambda idx: rows[idx], num_cpus=1, resampling="bootstrap")
        want_b = [orc.compute_asl(r[~np.isnan(r)], "bootstrap", approx) for r in rows]
        np.testing.assert_allclose(got_b, want_b, rtol=1e-9)


def test_plan_blocks():
    from scrna_parameter_estimation_amd.engine import plan_blocks

    rng = np.random.default_rng(0)
    gid = rng.integers(-1, 4, size=30000)
    gid[gid == 2] = 0  # an empty group
    order, cell0, bgrp, gblk0, ncell = plan_blocks(gid, 4)
    assert sorted(order.tolist()) == np.flatnonzero(gid >= 0).tolist()
    assert (np.diff(gid[order]) >= 0).all()                     # grouped, stable
    assert np.diff(cell0).max() <= 8192 and np.diff(cell0).min() > 0
    assert ncell[2] == 0 and gblk0[2] == gblk0[3]
    for b in range(len(bgrp)):
        assert (gid[order[cell0[b]:cell0[b + 1]]] == bgrp[b]).all()
    assert cell0[-1] == len(order)


def test_fdrcorrect_matches_bh():
    from scrna_parameter_estimation_amd.memento.util import _fdrcorrect

    p = np.array([0.01, 0.04, np.nan, 0.03, 0.5, 0.002])
    np.testing.assert_allclose(_fdrcorrect(p), [0.025, 0.05, 1.0, 0.05, 0.5, 0.01])


GLOO_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
dist.init_process_group(backend="gloo")
from scrna_parameter_estimation_amd.dist import Comm, shard_genes
from scrna_parameter_estimation_amd.memento.main import _mv_fit
c = Comm()
rank, world = c.rank, c.world
rng = np.random.default_rng(7)
G, N = 101, 500
mean, var = rng.lognormal(size=G), rng.lognormal(size=G)
rows = rng.poisson(3.0, size=(N, G)).astype(float)
lo, hi = shard_genes(G, rank, world)
tot = c.allreduce_sum(rows[:, lo:hi].sum(axis=1))           # per-cell totals over gene shards
assert np.array_equal(tot, rows.sum(axis=1))
gm = c.allgather_concat(mean[lo:hi]); gv = c.allgather_concat(var[lo:hi])
assert np.array_equal(gm, mean) and np.array_equal(gv, var)
assert np.allclose(_mv_fit(gm, gv), _mv_fit(mean, var), rtol=0, atol=0)   # pooled fit identical on every rank
e = c.allgather_concat(np.zeros(0) if rank == 0 else np.ones(3))         # ragged / empty shard
assert e.tolist() == [1.0, 1.0, 1.0]
# result gather of the gene-sharded 1D test: kept genes differ per shard, two result vectors, a gene with two tests
from scrna_parameter_estimation_amd.dist import gather_1d_ht, gather_pair_results, shard_pairs
names_all = [f"g{i}" for i in range(G)]
kept = rng.random(G) < 0.6
ntest = np.where(np.arange(G) %% 5 == 0, 2, 1)                            # treatment_for_gene-like: some genes carry two tests
coef_all = np.concatenate([np.full(ntest[i], mean[i]) for i in range(G) if kept[i]])
mine = [i for i in range(lo, hi) if kept[i]]
out = {"mean_coef": np.concatenate([np.full(ntest[i], mean[i]) for i in mine]) if mine else np.zeros(0),
       "mean_asl": np.concatenate([np.full(ntest[i], var[i]) for i in mine]) if mine else np.zeros(0)}
gn, full = gather_1d_ht(c, [names_all[i] for i in mine], out)
assert gn == [names_all[i] for i in range(G) if kept[i]]
assert np.array_equal(full["mean_coef"], coef_all) and len(full["mean_asl"]) == len(coef_all)
# cost-balanced (non-contiguous) shards: every gene owned once, loads level; the gather puts results back in the unsharded order
from scrna_parameter_estimation_amd.dist import gene_cost, shard_genes_balanced, shard_stream_uniforms
cost = gene_cost(mean)
sh = [shard_genes_balanced(cost, r_, world) for r_ in range(world)]
assert sorted(np.concatenate(sh).tolist()) == list(range(G)) and all((np.diff(x) > 0).all() for x in sh)
loads = [cost[x].sum() for x in sh]
assert max(loads) - min(loads) <= cost.max()
bal = [i for i in sh[rank] if kept[i]]
outb = {"mean_coef": np.concatenate([np.full(ntest[i], mean[i]) for i in bal]) if bal else np.zeros(0),
        "mean_asl": np.concatenate([np.full(ntest[i], var[i]) for i in bal]) if bal else np.zeros(0)}
gnb, fullb = gather_1d_ht(c, [names_all[i] for i in bal], outb, gene_pos=np.array(bal, dtype=np.int64), n_tests=ntest[bal])
assert gnb == [names_all[i] for i in range(G) if kept[i]] and np.array_equal(fullb["mean_coef"], coef_all)
# the hash uniforms of a rank's chains come from the ONE global stream at the chains' unsharded positions (identical seeds)
ng_ = 3
live_all = rng.random((G, ng_)) < 0.8
kept_idx = np.flatnonzero(kept)
np.random.seed(11)
u_ref = np.random.random(2 * int(live_all[kept_idx].sum()))                 # the unsharded run: gene-major over the kept genes
state_ref = np.random.get_state()[1].copy()
ref1, ref0 = np.zeros((G, ng_)), np.zeros((G, ng_))
gg, jj = np.nonzero(live_all[kept_idx])                                      # row-major = gene-major, group by group
ref1[kept_idx[gg], jj], ref0[kept_idx[gg], jj] = u_ref[0::2], u_ref[1::2]
np.random.seed(11)
r1_, r0_ = shard_stream_uniforms(c, np.array(bal, dtype=np.int64), live_all[bal])
assert np.array_equal(r1_.reshape(-1, ng_), ref1[bal]) and np.array_equal(r0_.reshape(-1, ng_), ref0[bal])
assert np.array_equal(np.random.get_state()[1], state_ref)                 # every rank leaves the stream where the unsharded run does
# 2D: pair blocks from shard_pairs, results back in the caller's order on every rank
pairs = [(names_all[int(a)], names_all[int(b)]) for a, b in rng.integers(0, G, size=(37, 2))]
blk, pos = shard_pairs(pairs, rank, world)
assert [pairs[i] for i in pos] == blk
val = np.array([float(int(a[1:]) * 1000 + int(b[1:])) for a, b in blk])
full2 = gather_pair_results(c, pos, {"corr_coef": val}, len(pairs))
assert np.array_equal(full2["corr_coef"], np.array([float(int(a[1:]) * 1000 + int(b[1:])) for a, b in pairs]))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_gene_sharded_exchange_gloo_world2(tmp_path):
    script = tmp_path / "gloo_check.py"
    script.write_text(GLOO_SCRIPT % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_api_error_conventions_before_touching_the_device():
    """Same exception types as the reference for bad input: AssertionError for q >= 1 and non-CSR X
    (main.py:42-44), TypeError when `resampling` is missing (hypothesis_test.py:57) -- all raised before any HIP call."""
    import pandas as pd
    import scipy.sparse as sp

    from scrna_parameter_estimation_amd import AnnDataLite, memento

    X = sp.random(50, 8, density=0.3, format="csr", dtype=np.float32)
    X.data[:] = 1
    obs = pd.DataFrame({"q": np.full(50, 1.2), "g": np.arange(50) % 2})
    ad = AnnDataLite(X, obs)
    with pytest.raises(AssertionError):
        memento.setup_memento(ad, q_column="q")
    ad.obs["q"] = 0.1
    ad.X = X.tocsc()
    with pytest.raises(AssertionError):
        memento.setup_memento(ad, q_column="q")
    ad.uns["memento"] = {}
    cov = pd.DataFrame({"i": [1.0, 1.0]})
    with pytest.raises(TypeError):
        memento.ht_1d_moments(ad, covariate=cov, treatment=cov)
    with pytest.raises(TypeError):
        memento.ht_2d_moments(ad, covariate=cov, treatment=cov)
    with pytest.raises(AssertionError):
        memento.compute_1d_moments(AnnDataLite(X, obs.copy()))       # setup_memento was not run (main.py:181)
    with pytest.raises(TypeError):
        AnnDataLite(X.toarray())


def test_public_api_names_match_reference():
    """The 13 names the reference re-exports (memento/__init__.py:1) exist with the reference's leading parameters."""
    import inspect

    from scrna_parameter_estimation_amd import memento

    want = {
        "setup_memento": ["adata", "q_column", "inplace", "filter_mean_thresh", "trim_percent", "shrinkage", "num_bins", "estimator_type"],
        "create_groups": ["adata", "label_columns", "label_delimiter", "inplace"],
        "compute_1d_moments": ["adata", "inplace", "min_perc_group", "filter_genes", "gene_list"],
        "compute_2d_moments": ["adata", "gene_pairs", "inplace"],
        "ht_1d_moments": ["adata", "covariate", "treatment", "treatment_for_gene", "inplace", "num_boot", "verbose", "num_cpus"],
        "ht_2d_moments": ["adata", "covariate", "treatment", "treatment_for_gene", "inplace", "num_boot", "verbose", "num_cpus"],
        "get_1d_moments": ["adata", "groupby"], "get_2d_moments": ["adata", "groupby"], "get_1d_ht_result": ["adata"],
        "get_2d_ht_result": ["adata"], "prepare_to_save": ["adata", "keep"], "get_corr_matrix": ["adata", "group"], "get_groups": ["adata"],
    }
    for name, params in want.items():
        got = list(inspect.signature(getattr(memento, name)).parameters)
        assert got[: len(params)] == params, (name, got)
    assert inspect.signature(memento.ht_1d_moments).parameters["num_boot"].default == 10000
    assert inspect.signature(memento.setup_memento).parameters["filter_mean_thresh"].default == 0.07


def test_tail_fit_pool_does_not_rerun_unguarded_main(tmp_path):
    """The tail-fit workers must not re-execute a user script that has no __main__ guard (spawn would, by default)."""
    import subprocess
    import sys

    script = tmp_path / "unguarded.py"
    script.write_text(
        "import sys, os\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "print('MAIN EXECUTED', flush=True)\n"
        "import numpy as np\n"
        "from scrna_parameter_estimation_amd.memento import asl\n"
        "rng = np.random.default_rng(0)\n"
        "B = 600\n"
        "st = np.zeros((4, 8)); rows = rng.normal(size=(4, B + 1)); rows[:, 0] = 0.5\n"
        "st[:, 0] = 0.5; st[:, 1] = 1.0; st[:, 2] = B; st[:, 3] = 3\n"
        "p = asl.asl_from_stats(st, False, lambda idx: rows[idx], num_cpus=2)\n"
        "assert np.isfinite(p).all()\n"
        "print('DONE', flush=True)\n")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.count("MAIN EXECUTED") == 1 and "DONE" in out.stdout


def test_pack_and_pair_tiles_are_a_valid_schedule():
    """Lane packing + dispatch order: every chain gets exactly one (tile, lane) slot, tiles hold chains of similar length,
    the first half of the grid holds the longest tiles (longest first) and the second half pairs them with the shortest."""
    from scrna_parameter_estimation_amd import engine

    rng = np.random.default_rng(0)
    K = np.sort(rng.integers(2, 400, size=50_000))[::-1]
    slot, nt = engine.pack_lanes(K, engine.PACK_WAVES)
    assert nt <= 2048 and len(np.unique(slot)) == len(K) and slot.max() < nt * 64
    slot2 = engine.pair_tiles(slot, nt, K)
    assert len(np.unique(slot2)) == len(K) and slot2.max() < nt * 64
    # same tiles, only renumbered: chains that shared a tile still do, lanes unchanged
    assert (slot2 % 64 == slot % 64).all()
    same_before = slot[:-1] // 64 == slot[1:] // 64
    same_after = slot2[:-1] // 64 == slot2[1:] // 64
    assert (same_before == same_after).all()
    tile = slot2 // 64
    lanes = np.bincount(tile, minlength=nt)
    kmax = np.zeros(nt)
    np.maximum.at(kmax, tile, K)
    est = kmax * engine.PACK_COST[lanes - 1]
    h = nt // 2 + nt % 2
    assert (np.diff(est[:h]) <= 1e-9).all()            # first half: longest first
    assert (np.diff(est[h:2 * h]) >= -1e-9).all()      # second half: ascending, so tile t meets a short partner at t + h
    assert est[:h].min() >= est[h:].max() - 1e-9
    # many-chain regime (more tiles than resident wave slots): long chains ride in narrow waves, short ones 64 wide
    Kb = np.sort(np.clip(rng.lognormal(np.log(260), 0.75, 200_000), 2, 3500).astype(int))[::-1]
    sb, nb = engine.pack_lanes(Kb, engine.PACK_WAVES)
    assert nb > 2048 and len(np.unique(sb)) == len(Kb)
    tb = sb // 64
    lb = np.bincount(tb, minlength=nb)
    kb = np.zeros(nb)
    np.maximum.at(kb, tb, Kb)
    cost = kb * engine.PACK_COST[lb - 1]
    assert lb[0] < 8 and lb[-1] == 64 or lb[-2] == 64                  # heaviest chain nearly alone, lightest in full waves
    assert cost.max() <= max(Kb[0] * engine.PACK_COST[0], engine.PACK_TAIL * cost.sum() / engine.PACK_RATE) * 1.001
    assert len(np.unique(engine.pair_tiles(sb, nb, Kb))) == len(Kb)
    # a heavy head of long chains that 2,000 tiles cannot give lanes of their own (BASELINE configs[2]'s shape): narrower tiles in
    # up to three waves per SIMD (plus the late starters of PACK_OVERSUB) shorten the longest tile, and the packer takes them; every chain still has exactly one slot
    Kc = np.sort(np.concatenate([rng.integers(250, 351, size=600), np.clip(rng.lognormal(np.log(70), 0.45, 43_000), 2, 249).astype(int)]))[::-1]
    sc, nc = engine.pack_lanes(Kc, engine.PACK_WAVES)
    assert engine.PACK_LAST["chosen"].startswith("resident, 3 waves") and 2048 < nc <= 3 * engine.PAIR_SLOTS + engine.PACK_OVERSUB
    assert len(np.unique(sc)) == len(Kc) and len(np.unique(engine.pair_tiles(sc, nc, Kc))) == len(Kc)
    lc = np.bincount(sc // 64, minlength=nc)
    kc = np.zeros(nc)
    np.maximum.at(kc, sc // 64, Kc)
    assert (kc * engine.PACK_COST[lc - 1]).max() <= 0.97 * engine.PACK_LAST["longest_resident"]
    # tiles beyond the 3 x 1,024 resident slots start late (as the first workgroups retire): after pair_tiles they are the shortest of all
    sp = engine.pair_tiles(sc, nc, Kc)
    tp = sp // 64
    lp = np.bincount(tp, minlength=nc)
    kp = np.zeros(nc)
    np.maximum.at(kp, tp, Kc)
    estp = kp * engine.PACK_COST[lp - 1]
    if nc > 3 * engine.PAIR_SLOTS:
        assert estp[3 * engine.PAIR_SLOTS:].max() <= estp[:3 * engine.PAIR_SLOTS].min() + 1e-9
    # ... and it keeps 2,000 tiles when the longest chain already sits alone in its tile (the uniform K above)
    engine.pack_lanes(K, engine.PACK_WAVES)
    assert engine.PACK_LAST["chosen"] == "resident"
    # degenerate inputs
    assert engine.pack_lanes(np.zeros(0), 2000) [1] == 0
    s1, n1 = engine.pack_lanes(np.array([7]), 2000)
    assert n1 == 1 and engine.pair_tiles(s1, n1, np.array([7])).tolist() == [0]
    # fast mode: dense 64-wide tiles
    sd, nd = engine.pack_lanes(K[:1000], 2000, dense=True)
    assert nd == 16 and (np.bincount(sd // 64)[:15] == 64).all()


def test_lean_genextreme_fit_reproduces_scipys_fit_bit_for_bit():
    """memento/asl.py fits the tails with scipy's optimizer on a lean restatement of genextreme's penalized likelihood: every value
    of the objective and every fitted parameter must be the double scipy's own code returns (the p-values of the tail-fit branch,
    hypothesis_test.py:94-141, are pinned to 1e-5 through fixtures; this pins the mechanism)."""
    import warnings

    import scipy.stats as stats

    from scrna_parameter_estimation_amd.memento import asl

    if asl._FAST_FIT["state"] == "off":
        pytest.skip("scipy's private names moved: the lean objective is not in use")
    rng = np.random.default_rng(0)
    g = stats.genextreme
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for trial in range(1500):                                     # the objective, incl. parameters outside the data's support
            n = int(rng.choice([60, 90, 150, 300]))
            data = np.sort(rng.normal(0, 1, 1000))[:n] if trial % 2 else np.sort(rng.gumbel(0, 1, 1000))[-n:]
            c = float(rng.choice([0.5, -0.5, 0.0, 1.0, rng.normal(0, 0.7), rng.normal(0, 3)]))
            th = np.array([c, float(np.mean(data) + rng.normal(0, data.std() * 2)), float(abs(rng.normal(0, 1)) * data.std() + 1e-3)])
            a, b = g._penalized_nnlf(th, data), asl._fast_nnlf(th, data)
            assert a == b or (np.isnan(a) and np.isnan(b)), (th, a, b)
        assert asl._fast_nnlf(np.array([0.1, 0.0, -1.0]), data) == np.inf and asl._fast_nnlf(np.array([np.nan, 0.0, 1.0]), data) == np.inf
        for trial in range(60):                                       # whole fits: same parameters or the same refusal
            null = rng.normal(0, 1, 1000) if trial % 3 else rng.standard_t(4, 1000)
            srt = np.sort(null)
            k = int(rng.choice([300, 270, 150, 60]))
            tail = srt[:k] if trial % 2 else srt[-k:]
            try:
                want = tuple(g.fit(tail))
            except Exception:
                want = None
            try:
                got = tuple(asl._fast_gev_fit(tail))
            except Exception:
                got = None
            assert got == want
        asl._FAST_FIT["state"] = "unchecked"                          # the switch: first call checks, later calls take the lean path
        first = asl.gev_fit(tail)
        assert asl._FAST_FIT["state"] == "on" and tuple(asl.gev_fit(tail)) == tuple(first) == want


def test_shard_pairs_partitions_the_pair_list():
    from scrna_parameter_estimation_amd.dist import shard_pairs

    rng = np.random.default_rng(1)
    genes = [f"g{i}" for i in range(40)]
    pairs = [(genes[a], genes[b]) for a, b in rng.integers(0, 40, size=(500, 2))]
    for world in (1, 2, 8):
        seen = np.zeros(len(pairs), dtype=int)
        sizes = []
        for r in range(world):
            mine, pos = shard_pairs(pairs, r, world)
            assert [pairs[i] for i in pos] == mine
            seen[pos] += 1
            sizes.append(len(mine))
        assert (seen == 1).all() and max(sizes) - min(sizes) <= 1
