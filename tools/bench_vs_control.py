"""Perturb-seq shape (BASELINE configs[4]): n_guides guide groups x 1 shared control (20 % of the cells), every kept gene
tested guide-vs-control in ONE call (memento.ht_1d_vs_control), next to the per-guide loop the reference's analyses use
(subset -> ht_1d_moments with 2 groups; here timed on a few guides through the same HIP path and extrapolated).
usage: python tools/bench_vs_control.py [cells genes n_guides num_boot approx(0/1)]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch, scipy.sparse as sp
import bench
from scrna_parameter_estimation_amd import AnnDataLite, memento


def main():
    cells, genes, n_guides, B, approx = [int(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else (200_000, 15_000, 500, 5_000, 0)
    cfg = dict(cells=cells, genes=genes, density=0.05)
    # multi-GPU: one process per GPU (torch.distributed.run); every rank holds all cells x its own gene shard of this shape
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count()))
    comm = None
    if world > 1:
        import torch.distributed as dist
        from scrna_parameter_estimation_amd.dist import Comm
        backend = os.environ.get("MM_DIST_BACKEND", "nccl")
        dist.init_process_group(backend=backend)
        comm = Comm(device="cuda" if backend == "nccl" else "cpu")
    csr = bench.synth_device_csr(cfg, 20250117 + 5 + 1000 * rank, torch)
    rng = np.random.default_rng(20250117 + 5)
    is_ctrl = rng.random(cells) < 0.2
    guide = np.where(is_ctrl, 0, 1 + rng.integers(0, n_guides, size=cells))
    obs = pd.DataFrame({"guide": guide, "q": np.full(cells, 0.07)})
    adata = AnnDataLite(sp.csr_matrix((cells, genes), dtype=np.float32), obs, pd.DataFrame(index=[f"r{rank}g{i}" for i in range(genes)]))
    t0 = time.time()
    memento.setup_memento(adata, q_column="q", device_csr=csr, comm=comm)
    memento.create_groups(adata, label_columns=["guide"])
    torch.cuda.synchronize(); t1 = time.time()
    memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
    torch.cuda.synchronize(); t2 = time.time()
    m = adata.uns["memento"]
    ctrl = [g for g in m["groups"] if g.split("^")[-1] == "0"][0]
    print(f"setup+groups {t1-t0:.2f}s compute_1d_moments {t2-t1:.2f}s genes kept {len(m['_hip'].gene_idx)} groups {len(m['groups'])} control {ctrl}", flush=True)
    np.random.seed(0)
    t3 = time.time()
    df = memento.ht_1d_vs_control(adata, control=ctrl, num_boot=B, num_cpus=16, approx=bool(approx))
    torch.cuda.synchronize(); t4 = time.time()
    n = len(df)
    print(f"ht_1d_vs_control: {n} (gene, guide) tests in {t4-t3:.2f}s -> {n/(t4-t3):.0f} tests/s ({n/(t4-t2+ (t2-t1)):.0f} incl. moments); "
          f"finite de_pval {np.isfinite(df.de_pval).mean():.3f} dv_pval {np.isfinite(df.dv_pval).mean():.3f}", flush=True)


if __name__ == "__main__":
    main()
