"""cProfile of one C3 bench step (host side): where the ~0.15 s outside the replay kernel goes.  usage: python tools/step_cprofile.py [config]"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch, scipy.sparse as sp
import bench
from scrna_parameter_estimation_amd import AnnDataLite, engine, memento

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
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
st = adata.uns["memento"]["_hip"]
full_idx = st.gene_idx.copy()


def step(seed):
    st.gene_idx = full_idx.copy(); st.var_names = None
    for k in ("size_factor", "approx_size_factor", "all_approx_size_factor"):
        adata.uns["memento"].pop(k, None)
    for g in adata.uns["memento"]["groups"]:
        adata.uns["memento"]["group_cells"][g].shape = (adata.uns["memento"]["group_cells"][g].shape[0], G)
    np.random.seed(seed)
    memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
    memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=B, num_cpus=16, verbose=0, resampling="bootstrap", approx=False)


step(1); step(2)
pr = cProfile.Profile(); pr.enable(); step(3); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:6000])
