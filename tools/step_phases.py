"""Where one bench step spends its wall time outside the replay kernel: wraps the phases of compute_1d_moments / ht_1d_moments with
torch.cuda.synchronize() + wall clocks (diagnostic: the synchronisation removes the overlap the real step has).
usage: python tools/step_phases.py [config=C3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch, scipy.sparse as sp
import bench
from scrna_parameter_estimation_amd import AnnDataLite, engine, memento
from scrna_parameter_estimation_amd.memento import asl as _asl

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = bench.CONFIGS[name]
N, G, B = cfg["cells"], cfg["genes"], cfg["num_boot"]
ng = cfg["n_cond"] * cfg["n_rep"]
csr = bench.synth_device_csr(cfg, 20250117, torch)
grp = np.random.default_rng(20250117).integers(0, ng, size=N)
obs = pd.DataFrame({"cond": grp // cfg["n_rep"], "rep": grp % cfg["n_rep"], "q": np.full(N, 0.07)})
adata = AnnDataLite(sp.csr_matrix((N, G), dtype=np.float32), obs, pd.DataFrame(index=[f"g{i}" for i in range(G)]))
memento.setup_memento(adata, q_column="q", device_csr=csr)
memento.create_groups(adata, label_columns=["cond", "rep"])
gdf = memento.get_groups(adata)
cov = pd.DataFrame({"intercept": np.ones(len(gdf))}, index=gdf.index)
trt = pd.DataFrame({"cond": (gdf["cond"].astype(int) == cfg["n_cond"] - 1).astype(float)}, index=gdf.index)
T = {}


def timed(label, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.time()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[label] = T.get(label, 0.0) + time.time() - t0
        return r
    return w


engine.Bootstrap1D.__init__ = timed("Bootstrap1D.__init__ (hist, bins count)", engine.Bootstrap1D.__init__)
engine.Bootstrap1D.run = timed("Bootstrap1D.run (pack, order, replay, fill)", engine.Bootstrap1D.run)
engine.Bootstrap1D.contract = timed("Bootstrap1D.contract", engine.Bootstrap1D.contract)
engine.Bootstrap1D.alloc_outputs = timed("alloc_outputs", engine.Bootstrap1D.alloc_outputs)
_orig_asl = _asl.asl_from_stats
memento.main._asl.asl_from_stats = timed("asl_from_stats (enqueue)", _orig_asl)
for rd in range(2):
    T.clear()
    st = adata.uns["memento"]["_hip"]
    st.gene_idx = np.arange(G); st.var_names = None
    for k in ("size_factor", "approx_size_factor", "all_approx_size_factor"):
        adata.uns["memento"].pop(k, None)
    np.random.seed(5 + rd)
    torch.cuda.synchronize(); t0 = time.time()
    memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
    torch.cuda.synchronize(); t1 = time.time()
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=B, num_cpus=16, verbose=0, resampling="bootstrap", approx=False)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"round {rd}: compute_1d_moments {t1 - t0:.3f} s, ht_1d_moments {t2 - t1:.3f} s; inside ht: " +
          ", ".join(f"{k} {v:.3f}" for k, v in T.items()) + f"; unaccounted {t2 - t1 - sum(T.values()):.3f}", flush=True)
