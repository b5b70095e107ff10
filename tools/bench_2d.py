"""2D sanity / timing run (shape of BASELINE configs[3], scaled): compute_2d_moments + ht_2d_moments for
n_left x n_right gene pairs.  usage: python tools/bench_2d.py [cells genes n_left n_right num_boot]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch, scipy.sparse as sp
import bench
from scrna_parameter_estimation_amd import AnnDataLite, memento

torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count()))
cells, genes, nl, nr, B = [int(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else (100_000, 4000, 60, 60, 500)
cfg = dict(cells=cells, genes=genes, density=0.08)
csr = bench.synth_device_csr(cfg, 3, torch)
rng = np.random.default_rng(1)
grp = rng.integers(0, 2, size=cells)
obs = pd.DataFrame({"cond": grp, "q": np.full(cells, 0.07)})
adata = AnnDataLite(sp.csr_matrix((cells, genes), dtype=np.float32), obs, pd.DataFrame(index=[f"g{i}" for i in range(genes)]))
memento.setup_memento(adata, q_column="q", device_csr=csr)
memento.create_groups(adata, label_columns=["cond"])
memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
names = memento.main._var_names(adata)
print("genes kept", len(names))
left, right = names[:nl], names[nl:nl + nr]
pairs = [(a, b) for a in left for b in right]
# multi-GPU: one process per GPU (RANK / WORLD_SIZE), every rank tests its own block of pairs -- no collective
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
if world > 1:
    from scrna_parameter_estimation_amd.dist import shard_pairs
    pairs, _ = shard_pairs(pairs, rank, world)
gdf = memento.get_groups(adata)
cov = pd.DataFrame({"intercept": np.ones(len(gdf))}, index=gdf.index)
trt = pd.DataFrame({"cond": gdf["cond"].astype(float)}, index=gdf.index)
torch.cuda.synchronize(); t0 = time.time()
memento.compute_2d_moments(adata, pairs)
torch.cuda.synchronize(); t1 = time.time()
np.random.seed(0)
memento.ht_2d_moments(adata, covariate=cov, treatment=trt, num_boot=B, num_cpus=8, verbose=0, resampling="bootstrap", approx=True)
torch.cuda.synchronize(); t2 = time.time()
ht = adata.uns["memento"]["2d_ht"]
bs = adata.uns["memento"]["_hip"].last_bootstrap2d
print(f"pairs={len(pairs)} compute_2d={t1-t0:.3f}s ht_2d={t2-t1:.3f}s -> {len(pairs)/(t2-t0):.1f} pair-tests/s; K mean {bs.K.mean():.0f} max {bs.K.max()}; finite p {np.isfinite(ht['corr_asl']).mean():.3f}")
from scrna_parameter_estimation_amd import engine
st = adata.uns["memento"]["_hip"]
# algorithmic bytes of the pair kernels: every pair reads its partner column's entries (4 B each) in every block
name_col = {n: i for i, n in enumerate(names)}
nnz_gene = st.blocks.blk_cnt.astype(np.int64).sum(axis=0)[st.gene_idx]
right_bytes = 4 * int(sum(nnz_gene[name_col[b]] for _, b in pairs))
rows = int(getattr(bs, "wave_steps_per_replicate", 0))
print(f"pair kernels: partner-column entries read per pass {right_bytes / 1e9:.2f} GB; last replay chunk: {bs.n_tiles} tiles, "
      f"{bs.draws_per_replicate} useful draws and {rows} wave-steps per replicate -> lane occupancy {bs.draws_per_replicate / max(1, rows * 64):.3f}")
print("packing:", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in engine.PACK_LAST.items()})
