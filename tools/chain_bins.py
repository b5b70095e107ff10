"""Multiplicities of the bins of the longest / typical chains of a config (what the samplers see).  usage: python tools/chain_bins.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch, scipy.sparse as sp
import bench
from scrna_parameter_estimation_amd import AnnDataLite, engine, memento

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = dict(bench.CONFIGS[name])
N, G, B = cfg["cells"], cfg["genes"], 10
ng = cfg["n_cond"] * cfg["n_rep"]
csr = bench.synth_device_csr(cfg, 20250117, torch)
grp = np.random.default_rng(20250117).integers(0, ng, size=N)
obs = pd.DataFrame({"cond": grp // cfg["n_rep"], "rep": grp % cfg["n_rep"], "q": np.full(N, 0.07)})
adata = AnnDataLite(sp.csr_matrix((N, G), dtype=np.float32), obs, pd.DataFrame(index=[f"g{i}" for i in range(G)]))
memento.setup_memento(adata, q_column="q", device_csr=csr)
memento.create_groups(adata, label_columns=["cond", "rep"])
memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
m = adata.uns["memento"]; st = m["_hip"]
gq = np.array([m["group_q"][g] for g in m["groups"]])
bs = engine.Bootstrap1D(st.blocks, st.gene_idx, st.maxx, st.sf_bin, st.sf_table, gq, B)
order = np.argsort(-bs.K, kind="stable")
edges = [1, 2, 3, 4, 6, 11, 31, 101, 1001, 10 ** 9]
for label, sel in (("64 longest chains", order[:64]), ("chains 1000-1064 by length", order[1000:1064]), ("64 median chains", order[len(order) // 2: len(order) // 2 + 64])):
    mus = np.concatenate([bs.bins_of_pair(int(p))[2] for p in sel]).astype(np.int64)
    h = np.histogram(mus, bins=edges)[0] / len(mus)
    print(f"{label}: K {bs.K[sel].max()}..{bs.K[sel].min()}; share of bins by multiplicity: " +
          ", ".join(f"[{a},{b}) {x:.3f}" for a, b, x in zip(edges[:-1], edges[1:], h)) + f"; expected share of draws with X=0: {np.exp(-mus.astype(float)).mean():.3f}", flush=True)
