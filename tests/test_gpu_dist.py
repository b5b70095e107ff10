"""Gene-sharded execution (one process per rank) gives the unsharded -- i.e. the real reference's --
results: two ranks (gloo rendezvous, both on cuda:0) each hold all cells x half of the genes of the api_small fixture."""

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
Xs = sp.csr_matrix(X[:, lo:hi])
obs = pd.DataFrame({"cond": g["in_cond"], "rep": g["in_rep"], "q": g["in_q"]}, index=[f"c{i}" for i in range(X.shape[0])])
adata = AnnDataLite(Xs, obs, pd.DataFrame(index=g["in_gene_names"].tolist()[lo:hi]))
memento.setup_memento(adata, q_column="q", comm=comm)
memento.create_groups(adata, label_columns=["cond", "rep"])
memento.compute_1d_moments(adata, min_perc_group=0.7)
m = adata.uns["memento"]
gdf = memento.get_groups(adata)
cov = pd.DataFrame(g["covariate"], index=gdf.index, columns=["intercept"])
trt = pd.DataFrame(g["treatment"], index=gdf.index, columns=["cond"])
np.random.seed(1 + comm.rank)
memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=100, num_cpus=1, verbose=0, resampling="bootstrap", approx=True)
groups = m["groups"]
np.savez(os.path.join(%(out)r, f"rank{comm.rank}.npz"), size_factor=adata.obs["memento_size_factor"].values,
         gene_list=np.array(m["gene_list"]), mean=np.stack([m["1d_moments"][k][0] for k in groups]),
         res_var=np.stack([m["1d_moments"][k][2] for k in groups]), mv=np.asarray(m["mv_regressor"]["all"]),
         mean_coef=m["1d_ht"]["mean_coef"], var_coef=m["1d_ht"]["var_coef"])
dist.barrier()
dist.destroy_process_group()
"""


def test_two_gene_shards_equal_the_unsharded_reference(api_small, tmp_path):
    g = api_small
    script = tmp_path / "shard.py"
    script.write_text(SCRIPT % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    parts = [dict(np.load(tmp_path / f"rank{k}.npz")) for k in range(2)]
    for p in parts:                                              # every rank ends with the GLOBAL size factors and the pooled fit
        np.testing.assert_allclose(p["size_factor"], g["size_factor"], rtol=1e-12)
        np.testing.assert_allclose(p["mv"], g["mv_regressor"], rtol=1e-8)
    assert list(parts[0]["gene_list"]) + list(parts[1]["gene_list"]) == list(g["gene_list"])
    np.testing.assert_allclose(np.concatenate([p["mean"] for p in parts], axis=1), g["mean"], rtol=1e-11)
    np.testing.assert_allclose(np.concatenate([p["res_var"] for p in parts], axis=1), g["res_var"], rtol=1e-7, equal_nan=True)
    # observed coefficients do not depend on the bootstrap draws: identical to the reference's
    np.testing.assert_allclose(np.concatenate([p["mean_coef"] for p in parts]), g["ht_mean_coef"], rtol=1e-8, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(np.concatenate([p["var_coef"] for p in parts]), g["ht_var_coef"], rtol=1e-7, atol=1e-12, equal_nan=True)
