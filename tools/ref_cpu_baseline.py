"""Time the REAL reference (memento 0.0.9 under /root/reference) on its own CPU path, on gene subsamples of the BASELINE.json
shapes -- build container only (the reference cannot travel to the GPU box).  Writes baselines/ref_cpu_<config>.json, which
bench.py prints beside its own cpu_baseline as a stated, other-hardware figure.

  python tools/ref_cpu_baseline.py C1 C2 C3

What is timed: compute_1d_moments + ht_1d_moments(num_boot=B, approx=False, resampling='bootstrap') at num_cpus=1 and
num_cpus=8, on <= 200 kept genes of a matrix with the config's cells / groups / expression profile (per-gene work is independent:
memento/main.py:379-397, so gene-tests/s on the subsample is the per-gene rate of the full shape).  setup_memento runs on the
subsample too (untimed); the size factors are then set to the generator's per-cell depth (a small gene subsample cannot
estimate them)."""
import json
import os
import platform
import sys
import tempfile
import time
import warnings

import numpy as np
import pandas as pd
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")

SHAPES = {   # cells, genes of the full shape, density, conditions, replicates, bootstraps, genes timed
    "C1": dict(cells=2_700, genes=1_800, density=0.10, n_cond=2, n_rep=1, num_boot=100, n_timed=200),
    "C2": dict(cells=100_000, genes=20_000, density=0.05, n_cond=2, n_rep=4, num_boot=1_000, n_timed=64),
    "C3": dict(cells=1_000_000, genes=20_000, density=0.03, n_cond=2, n_rep=10, num_boot=10_000, n_timed=16),
}


def _stub_pkgs():
    d = tempfile.mkdtemp(prefix="memento_stubs_")
    for name, body in {"patsy": "def dmatrix(*a, **k):\n    raise NotImplementedError\n", "scanpy": "", "statsmodels": "",
                       "statsmodels/api": "", "statsmodels/stats": "",
                       "statsmodels/stats/multitest": "def fdrcorrection(*a, **k):\n    raise NotImplementedError\n"}.items():
        p = os.path.join(d, name)
        os.makedirs(p, exist_ok=True)
        open(os.path.join(p, "__init__.py"), "w").write(body)
    return d


def make_subsample(cfg, seed=20250117):
    """Kept-gene columns of the config's synthetic matrix (same generator family as bench.synth_device_csr: lognormal gene
    means calibrated to the density, lognormal depth, gamma-Poisson counts), float64 CSR, plus obs."""
    from scrna_parameter_estimation_amd.synth import _expected_density
    from scrna_parameter_estimation_amd.anndata_lite import AnnDataLite

    N, G, dens = cfg["cells"], cfg["genes"], cfg["density"]
    rng = np.random.default_rng(seed)
    mu = rng.lognormal(-2.2, 1.2, size=G)
    depth = np.random.default_rng(seed).lognormal(0.0, 0.35, size=N)
    nodes = np.quantile(depth, (np.arange(64) + 0.5) / 64)
    w = np.full(64, 1.0 / 64)
    lo, hi = 1e-4, 1e4
    for _ in range(60):
        mid = np.sqrt(lo * hi)
        if _expected_density(mu, mid, nodes, w) < dens:
            lo = mid
        else:
            hi = mid
    mu = mu * np.sqrt(lo * hi)
    expected = mu * depth.mean()                       # gamma(2, .5) has mean 1
    cand = np.flatnonzero(expected > 0.09)             # comfortably above the reference's mean filter (0.07)
    pick = cand[np.linspace(0, len(cand) - 1, min(cfg["n_timed"], len(cand))).astype(int)]
    cols = []
    for g in pick:
        lam = depth * mu[g] * rng.gamma(2.0, 0.5, size=N)
        cols.append(sp.csc_matrix(rng.poisson(lam).astype(np.float64).reshape(-1, 1)))
    X = sp.hstack(cols, format="csr")
    n_groups = cfg["n_cond"] * cfg["n_rep"]
    grp = np.random.default_rng(seed).integers(0, n_groups, size=N)
    obs = pd.DataFrame({"cond": grp // cfg["n_rep"], "rep": grp % cfg["n_rep"], "q": np.full(N, 0.07)}, index=[f"c{i}" for i in range(N)])
    obs["depth_sf"] = depth / depth.mean()
    return AnnDataLite(X, obs, pd.DataFrame(index=[f"g{int(g)}" for g in pick])), len(cand)


def main(names):
    sys.path.insert(0, _stub_pkgs())
    sys.path.insert(0, "/root/reference")
    import memento
    import scipy
    import sklearn

    os.makedirs(os.path.join(ROOT, "baselines"), exist_ok=True)
    for name in names:
        cfg = SHAPES[name]
        base, n_cand = make_subsample(cfg)
        res = {"config": name, "shape": {k: cfg[k] for k in ("cells", "genes", "density", "n_cond", "n_rep", "num_boot")},
               "what": "REAL reference (memento 0.0.9): compute_1d_moments + ht_1d_moments(approx=False, resampling='bootstrap') wall time",
               "hardware": f"build container, {os.cpu_count()} cores ({platform.processor() or platform.machine()}); NOT the GPU box's host",
               "versions": {"numpy": np.__version__, "scipy": scipy.__version__, "sklearn": sklearn.__version__, "python": platform.python_version()},
               "genes_passing_filter_in_full_shape_estimate": int(n_cand), "runs": []}
        for ncpu in (1, 8):
            adata = base.copy()
            memento.setup_memento(adata, q_column="q")
            # a <= 200-gene subsample cannot estimate size factors (the trimmed gene set leaves most cells without counts):
            # use the generator's per-cell depth, normalised to mean 1, as the full matrix's estimate would be
            adata.obs["memento_size_factor"] = adata.obs["depth_sf"].values
            memento.create_groups(adata, label_columns=["cond", "rep"])
            gdf = memento.get_groups(adata)
            cov = pd.DataFrame({"intercept": np.ones(len(gdf))}, index=gdf.index)
            trt = pd.DataFrame({"cond": (gdf["cond"].astype(int) == cfg["n_cond"] - 1).astype(float)}, index=gdf.index)
            np.random.seed(1)
            t0 = time.time()
            memento.compute_1d_moments(adata, min_perc_group=0.7)
            t1 = time.time()
            memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=cfg["num_boot"], num_cpus=ncpu, verbose=0,
                                  resampling="bootstrap", approx=False)
            t2 = time.time()
            n_tests = len(adata.uns["memento"]["1d_ht"]["mean_asl"])
            res["runs"].append({"num_cpus": ncpu, "genes_tested": int(n_tests), "moments_s": round(t1 - t0, 3), "ht_s": round(t2 - t1, 3),
                                "gene_tests_per_s": round(n_tests / (t2 - t0), 4),
                                "gene_tests_per_s_per_core": round(n_tests / (t2 - t0) / ncpu, 4)})
            print(name, res["runs"][-1], flush=True)
        json.dump(res, open(os.path.join(ROOT, "baselines", f"ref_cpu_{name}.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:] or ["C1", "C2", "C3"])
