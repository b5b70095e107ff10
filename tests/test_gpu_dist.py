"""Gene-sharded execution (one process per rank) gives the unsharded -- i.e. the real reference's --
results: two ranks (gloo rendezvous, both on cuda:0) each hold all cells x half of the genes of the api_small fixture.
Both ways in: the caller pre-slices a contiguous gene range of X on the host, or (``shard=True``) hands the FULL matrix to every
rank and a COST-BALANCED gene set (not a range) is cut out on the device (mm_csr_colsum + mm_csr_mapsplit).  Every rank is
seeded alike and takes its chains' hash uniforms from the one global np.random stream at their unsharded positions, so the
gathered standard errors and p-values are those of the 1-rank run -- and the real reference's up to the first gene whose
invalid replicates the reference re-fills from that stream (the timed mode re-fills on the device).  A second test drives the
same exchanges over the nccl (= RCCL) backend on cuda tensors.  ht_1d_moments ends with the gather: every rank holds the full result vectors in
the unsharded run's gene order (the scatter-back of memento/main.py:399-412).  2D: pair blocks from dist.shard_pairs, results
reassembled in the caller's pair order."""

import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, pandas as pd, scipy.sparse as sp
import torch, torch.distributed as dist
dist.init_process_group(backend="gloo")
from scrna_parameter_estimation_amd import AnnDataLite, memento
from scrna_parameter_estimation_amd.dist import Comm, shard_genes
comm = Comm(device="cpu")
g = dict(np.load(os.path.join(%(root)r, "tests", "golden", "api_small.npz"), allow_pickle=False))
X = sp.csr_matrix((g["in_data"].astype(np.float32), g["in_indices"], g["in_indptr"]), shape=tuple(g["in_shape"]))
lo, hi = shard_genes(X.shape[1], comm.rank, comm.world)
obs = pd.DataFrame({"cond": g["in_cond"], "rep": g["in_rep"], "q": g["in_q"]}, index=[f"c{i}" for i in range(X.shape[0])])
if %(device_split)r:
    adata = AnnDataLite(X, obs, pd.DataFrame(index=g["in_gene_names"].tolist()))       # the FULL matrix on every rank
    memento.setup_memento(adata, q_column="q", comm=comm, shard=True)                   # device-side column split
else:
    Xs = sp.csr_matrix(X[:, lo:hi])
    adata = AnnDataLite(Xs, obs, pd.DataFrame(index=g["in_gene_names"].tolist()[lo:hi]))
    memento.setup_memento(adata, q_column="q", comm=comm)
memento.create_groups(adata, label_columns=["cond", "rep"])
memento.compute_1d_moments(adata, min_perc_group=0.7)
m = adata.uns["memento"]
gdf = memento.get_groups(adata)
cov = pd.DataFrame(g["covariate"], index=gdf.index, columns=["intercept"])
trt = pd.DataFrame(g["treatment"], index=gdf.index, columns=["cond"])
np.random.seed(int(g["ht_seed"]))                      # the SAME seed on every rank
# contiguous pre-sliced shards: strict replay (the ranks take turns, the stream state handed on) = the real reference, fills and all;
# cost-balanced device split: the timed mode (every rank takes its chains' uniforms out of the one global stream)
memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0, resampling="bootstrap",
                      approx=bool(g["approx"]), strict=not %(device_split)r)
groups = m["groups"]
np.savez(os.path.join(%(out)r, f"rank{comm.rank}.npz"), size_factor=adata.obs["memento_size_factor"].values,
         gene_list=np.array(m["gene_list"]), mean=np.stack([m["1d_moments"][k][0] for k in groups]),
         res_var=np.stack([m["1d_moments"][k][2] for k in groups]), mv=np.asarray(m["mv_regressor"]["all"]),
         mean_coef=m["1d_ht"]["mean_coef"], var_coef=m["1d_ht"]["var_coef"], ht_names=np.array(m["1d_ht"]["gene_names"]),
         mean_se=m["1d_ht"]["mean_se"], mean_asl=m["1d_ht"]["mean_asl"], var_se=m["1d_ht"]["var_se"], var_asl=m["1d_ht"]["var_asl"],
         df_genes=np.array(memento.get_1d_ht_result(adata)["gene"].tolist()))
# 2D on pair blocks (every rank needs all genes' columns: a second, unsharded state on the same device)
from scrna_parameter_estimation_amd.dist import shard_pairs, gather_pair_results
ad2 = AnnDataLite(X, obs.copy(), pd.DataFrame(index=g["in_gene_names"].tolist()))
memento.setup_memento(ad2, q_column="q")
memento.create_groups(ad2, label_columns=["cond", "rep"])
memento.compute_1d_moments(ad2, min_perc_group=0.7)
names = np.asarray(ad2.var.index)
pairs = list(zip(names[g["pair_idx1"]].tolist(), names[g["pair_idx2"]].tolist()))
mine, pos = shard_pairs(pairs, comm.rank, comm.world)
memento.compute_2d_moments(ad2, mine)
gdf2 = memento.get_groups(ad2)
cov2 = pd.DataFrame(g["covariate"], index=gdf2.index, columns=["intercept"])
trt2 = pd.DataFrame(g["treatment"], index=gdf2.index, columns=["cond"])
np.random.seed(7 + comm.rank)
memento.ht_2d_moments(ad2, covariate=cov2, treatment=trt2, num_boot=100, num_cpus=1, verbose=0, resampling="bootstrap", approx=True)
m2 = ad2.uns["memento"]
vals = {"corr_coef": m2["2d_ht"]["corr_coef"]}
for gi, grp in enumerate(m2["groups"]):
    vals[f"corr_{gi}"] = m2["2d_moments"][grp]["corr"]
full = gather_pair_results(comm, pos, vals, len(pairs))
np.savez(os.path.join(%(out)r, f"pairs_rank{comm.rank}.npz"), **full)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("device_split", [False, True])
def test_two_gene_shards_equal_the_unsharded_reference(api_small, tmp_path, device_split):
    g = api_small
    script = tmp_path / "shard.py"
    script.write_text(SCRIPT % {"root": ROOT, "out": str(tmp_path), "device_split": device_split})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    parts = [dict(np.load(tmp_path / f"rank{k}.npz")) for k in range(2)]
    for p in parts:                                              # every rank ends with the GLOBAL size factors and the pooled fit
        np.testing.assert_allclose(p["size_factor"], g["size_factor"], rtol=1e-12)
        np.testing.assert_allclose(p["mv"], g["mv_regressor"], rtol=1e-8)
    # the two shards partition the kept genes (contiguous halves when pre-sliced; a cost-balanced, interleaved split on the device)
    both = list(parts[0]["gene_list"]) + list(parts[1]["gene_list"])
    assert sorted(both) == sorted(g["gene_list"])
    if not device_split:
        assert both == list(g["gene_list"])
    else:
        assert both != list(g["gene_list"]), "cost-balanced shards are gene sets, not ranges"
    at = {n: i for i, n in enumerate(both)}
    back = [at[n] for n in g["gene_list"]]                                              # unsharded order
    np.testing.assert_allclose(np.concatenate([p["mean"] for p in parts], axis=1)[:, back], g["mean"], rtol=1e-11)
    np.testing.assert_allclose(np.concatenate([p["res_var"] for p in parts], axis=1)[:, back], g["res_var"], rtol=1e-7, equal_nan=True)
    # the gather: EVERY rank holds the full result vectors, in the unsharded gene order; observed coefficients do not depend on
    # the bootstrap draws, so they are the real reference's
    for p in parts:
        assert list(p["ht_names"]) == list(g["gene_list"]) and list(p["df_genes"]) == list(g["gene_list"])
        np.testing.assert_allclose(p["mean_coef"], g["ht_mean_coef"], rtol=1e-8, atol=1e-12, equal_nan=True)
        np.testing.assert_allclose(p["var_coef"], g["ht_var_coef"], rtol=1e-7, atol=1e-12, equal_nan=True)
    if not device_split:
        # strict replay over two ranks == the REAL reference: standard errors 1e-8, p-values 1e-5 (fixture), every gene
        for p in parts:
            for k in ("mean_se", "var_se"):
                np.testing.assert_allclose(p[k], g["ht_" + k], rtol=1e-8, atol=1e-12, equal_nan=True, err_msg=k)
            for k in ("mean_asl", "var_asl"):
                np.testing.assert_allclose(p[k], g["ht_" + k], rtol=1e-5, atol=1e-12, equal_nan=True, err_msg=k)
    else:
        # timed mode over two cost-balanced shards == the same call on ONE rank (this process, same seed): N-rank results are
        # 1-rank results.  The mean statistics agree to round-off; the variability statistics see the pooled mean-variance fit,
        # which the ranks compute from the gathered moments in another order (np.polyfit: ~1e-9 relative).  Invalid replicates
        # are re-filled on the device from streams keyed by (gene, group), so sharding does not change them either.
        from test_gpu_api import _design, _run_to_moments
        memento, adata = _run_to_moments(g)
        cov, trt = _design(memento, adata, g)
        np.random.seed(int(g["ht_seed"]))
        memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=int(g["num_boot"]), num_cpus=1, verbose=0, resampling="bootstrap",
                              approx=bool(g["approx"]))
        one = adata.uns["memento"]["1d_ht"]
        assert adata.uns["memento"]["_hip"].refill_stats["genes_refilled"] > 0         # the refill path is part of what is compared
        for p in parts:
            for k in ("mean_coef", "mean_se", "mean_asl"):
                np.testing.assert_allclose(p[k], one[k], rtol=1e-12, atol=1e-14, equal_nan=True, err_msg=k)
            for k in ("var_coef", "var_se", "var_asl"):
                np.testing.assert_allclose(p[k], one[k], rtol=1e-6, atol=1e-9, equal_nan=True, err_msg=k)
    # 2D: pair blocks reassembled in the caller's order == the unsharded fixture (moments exactly; observed coefficients too)
    pr = [dict(np.load(tmp_path / f"pairs_rank{k}.npz")) for k in range(2)]
    for p in pr:
        for gi in range(len(g["groups"])):
            np.testing.assert_allclose(p[f"corr_{gi}"], g["corr2d"][gi], rtol=1e-8, equal_nan=True)
        # ht_2d skips self pairs and fills duplicates within ONE call's list (main.py:473-482); a duplicate split over two ranks
        # is tested on both, with the same observed coefficient
        ok = np.isfinite(g["ht2_corr_coef"])
        np.testing.assert_allclose(p["corr_coef"][ok], g["ht2_corr_coef"][ok], rtol=1e-8, atol=1e-12)


NCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, pandas as pd, scipy.sparse as sp
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
from scrna_parameter_estimation_amd import AnnDataLite, memento
from scrna_parameter_estimation_amd.dist import Comm
comm = Comm()
assert comm.device == "cuda" and dist.get_backend() == "nccl"
a = comm.allreduce_sum(np.arange(5.0))                       # RCCL all-reduce / all-gather on cuda tensors
b = comm.allgather_concat(np.arange(3.0) + comm.rank)
assert a.tolist() == (np.arange(5.0) * comm.world).tolist() and len(b) == 3 * comm.world
g = dict(np.load(os.path.join(%(root)r, "tests", "golden", "api_small.npz"), allow_pickle=False))
X = sp.csr_matrix((g["in_data"].astype(np.float32), g["in_indices"], g["in_indptr"]), shape=tuple(g["in_shape"]))
obs = pd.DataFrame({"cond": g["in_cond"], "rep": g["in_rep"], "q": g["in_q"]}, index=[f"c{i}" for i in range(X.shape[0])])
adata = AnnDataLite(X, obs, pd.DataFrame(index=g["in_gene_names"].tolist()))
memento.setup_memento(adata, q_column="q", comm=comm, shard=True)          # size factors through all-reduce, pooled fit through all-gather
memento.create_groups(adata, label_columns=["cond", "rep"])
memento.compute_1d_moments(adata, min_perc_group=0.7)
m = adata.uns["memento"]
np.savez(os.path.join(%(out)r, "nccl.npz"), size_factor=adata.obs["memento_size_factor"].values, mv=np.asarray(m["mv_regressor"]["all"]),
         gene_list=np.array(m["gene_list"]), mean=np.stack([m["1d_moments"][k][0] for k in m["groups"]]))
dist.barrier()
dist.destroy_process_group()
"""


def test_exchanges_run_over_nccl_on_cuda_tensors(api_small, tmp_path):
    """torch.distributed backend "nccl" (= RCCL on ROCm), one rank: Comm's all-reduce and all-gather run on cuda tensors inside
    setup_memento / compute_1d_moments (the backend of the multi-GPU bench); results equal the real reference's."""
    g = api_small
    script = tmp_path / "nccl.py"
    script.write_text(NCCL_SCRIPT % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = dict(np.load(tmp_path / "nccl.npz"))
    np.testing.assert_allclose(out["size_factor"], g["size_factor"], rtol=1e-12)
    np.testing.assert_allclose(out["mv"], g["mv_regressor"], rtol=1e-8)
    assert list(out["gene_list"]) == list(g["gene_list"])
    np.testing.assert_allclose(out["mean"], g["mean"], rtol=1e-11)
