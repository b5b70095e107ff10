"""bench.py -- gene-tests/s of the memento 1D hot path (compute_1d_moments + ht_1d_moments) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 launched by torch.distributed.run, one rank
per GPU).  One STEP = one pass of the hot path over the synthetic matrix, with the CSR and the
group-ordered count blocks already resident in HBM:
    compute_1d_moments  (K1 moments kernel + filters + pooled mean-variance fit)
  + ht_1d_moments       (K5 histograms, bin ordering, numpy-replay bootstrap, fill/log, contraction,
                         p-values incl. the host-side extreme-value tail fits)
Multi-GPU: genes are sharded, no data-path collective (the size factors are one all-reduce of an N-vector, the pooled
fit one small all-gather, the results one gather at the end of the step).  ``--scaling strong`` (default): the SAME
matrix -- BASELINE.json configs[2]: 1M cells x 20k genes -- with rank r holding all cells x genes [r G/N, (r+1) G/N);
``--scaling weak``: every rank holds a full-size gene shard (N x 20k genes in total).  value = gene-tests of all ranks /
max-over-ranks time.

Extra objects on the JSON line:
  "roofline"      the K1 kernel on the bytes it really moves (the packed 4 B/entry count blocks): bytes / HIP-event time vs 8 TB/s
  "roofline_csr"  SURVEY.md section 8(d)'s unit of work -- the CSR's 8 B/nnz + per-cell and output vectors -- over the time of
                  CSR -> moments for one grouping: ingest (K0: count + layout + scatter, paid once per grouping) + K1
  "bootstrap"     binomial draws/s of the replay kernel, refilled-chain fraction
  "cpu_baseline"  the CPU oracle on a bounded sample (rank 0, N=1 only); "reference_cpu": the REAL reference timed in the build
                  container (baselines/ref_cpu_*.json, other hardware, stated as such)
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: cells, genes, density, conditions, replicates, bootstraps   (BASELINE.json configs[1], [2])
    "C2": dict(cells=100_000, genes=20_000, density=0.05, n_cond=2, n_rep=4, num_boot=1_000),
    "C3": dict(cells=1_000_000, genes=20_000, density=0.03, n_cond=2, n_rep=10, num_boot=10_000),
    "tiny": dict(cells=6_000, genes=400, density=0.08, n_cond=2, n_rep=2, num_boot=200),
    # scaled-down shapes of configs[3] / [4] for sanity runs (not headline): many groups; see tools/bench_2d.py for 2D
    # shapes of the reference's own published timings, for context in DESIGN.md
    "tutorial": dict(cells=5_341, genes=7_000, density=0.10, n_cond=2, n_rep=1, num_boot=5_000),      # ifn_mono_ht.ipynb: 53.2 s / 1877 genes
    "runtime1M": dict(cells=1_000_000, genes=12_000, density=0.06, n_cond=2, n_rep=1, num_boot=1_000),  # runtime/plots.ipynb: 0.1113 s/gene
    "G36k": dict(cells=150_000, genes=36_601, density=0.02, n_cond=3, n_rep=1, num_boot=300),   # > 32768 genes: tiled ingest counters
    "C5s": dict(cells=60_000, genes=3_000, density=0.05, n_cond=101, n_rep=1, num_boot=500),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
TRAFFIC_FILE = "r03_k1_traffic_C3.json"   # this round's PMC record of K1's HBM traffic (tools/k1_traffic.py)


def calibrated_profile(cfg, seed):
    """Per-gene base means mu (all G genes) rescaled so that the expected nnz fraction hits cfg['density'], and the cell depths
    (SURVEY.md section 8d).  Host only; every rank computes the same arrays."""
    from scrna_parameter_estimation_amd.synth import _expected_density

    N, G, dens = cfg["cells"], cfg["genes"], cfg["density"]
    rng = np.random.default_rng(seed)
    mu = rng.lognormal(-2.2, 1.2, size=G)                                  # weak scaling: differs per rank (own gene shard)
    depth = np.random.default_rng(20250117).lognormal(0.0, 0.35, size=N)   # cells: the SAME cells on every rank
    nodes = np.quantile(depth, (np.arange(64) + 0.5) / 64)
    w = np.full(64, 1.0 / 64)
    lo, hi = 1e-4, 1e4
    for _ in range(60):
        mid = np.sqrt(lo * hi)
        if _expected_density(mu, mid, nodes, w) < dens:
            lo = mid
        else:
            hi = mid
    return mu * np.sqrt(lo * hi), depth


def synth_device_csr(cfg, seed, torch, genes=None):
    """Gamma-Poisson counts generated ON THE GPU (SURVEY.md section 8d shape); returns engine.DeviceCSR.
    ``genes``: only those genes of the matrix (ascending indices; strong scaling: every rank calibrates the SAME 20k-gene
    expression profile and generates the columns of its own shard)."""
    from scrna_parameter_estimation_amd import engine

    N, G = cfg["cells"], cfg["genes"]
    mu, depth = calibrated_profile(cfg, seed)
    if genes is not None:
        genes = np.asarray(genes, dtype=np.int64)
        mu = mu[genes]
        G = len(mu)
        seed = seed + 7919 * (int(genes[0]) + 1 if len(genes) else 1)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    d_mu = torch.from_numpy(mu.astype(np.float32)).cuda()
    d_depth = torch.from_numpy(depth.astype(np.float32)).cuda()
    idx_parts, val_parts, cnt_parts = [], [], []
    rows = 8192
    for r0 in range(0, N, rows):
        r1 = min(N, r0 + rows)
        u = torch.rand((2, r1 - r0, G), generator=gen, device="cuda")
        gam = -0.5 * torch.log(u[0] * u[1] + 1e-38)            # Gamma(2, scale 0.5)
        lam = d_depth[r0:r1, None] * d_mu[None, :] * gam
        x = torch.poisson(lam, generator=gen)
        x = torch.clamp(x, max=500000.0)
        nz = x != 0
        cnt_parts.append(nz.sum(dim=1))
        cols = nz.nonzero()[:, 1].to(torch.int32)
        idx_parts.append(cols)
        val_parts.append(x[nz])
        del u, gam, lam, x, nz
    counts = torch.cat(cnt_parts)
    indptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    indptr[1:] = torch.cumsum(counts, 0)
    return engine.DeviceCSR.from_device(indptr, torch.cat(idx_parts), torch.cat(val_parts).to(torch.float32), (N, G))


def balanced_shard(cfg, rank, world, seed=20250117):
    """Gene indices of rank ``rank``'s cost-balanced shard of the synthetic matrix (every rank computes the same partition)."""
    from scrna_parameter_estimation_amd.dist import gene_cost, shard_genes_balanced

    mu, _ = calibrated_profile(cfg, seed)
    return shard_genes_balanced(gene_cost(mu * float(np.exp(0.35 ** 2 / 2))), rank, world)     # E[depth] = exp(sigma^2 / 2), E[gamma] = 1


def sample_columns(csr, genes, torch):
    """Dense host columns [N][len(genes)] of a device CSR (for the CPU baseline sample)."""
    genes_t = torch.as_tensor(np.asarray(genes), dtype=torch.int32, device="cuda")
    hit = torch.isin(csr.indices, genes_t)
    pos = hit.nonzero()[:, 0]
    rows = torch.searchsorted(csr.indptr, pos, right=True) - 1
    cols = csr.indices[pos].long()
    lut = torch.full((csr.shape[1],), -1, dtype=torch.int64, device="cuda")
    lut[genes_t.long()] = torch.arange(len(genes), device="cuda")
    out = np.zeros((csr.shape[0], len(genes)))
    out[rows.cpu().numpy(), lut[cols].cpu().numpy()] = csr.data[pos].double().cpu().numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("MM_BENCH_CONFIG", "C3"), choices=list(CONFIGS))
    ap.add_argument("--num-cpus", type=int, default=0, help="host processes for the tail fits (default: min(16, cores / ranks))")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-baseline-cores", type=int, default=max(1, min(16, os.cpu_count() or 1)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--predict-shards", type=int, default=0, metavar="N",
                    help="single process, no collectives: time the step of each of the N cost-balanced gene shards of the config on this "
                         "GPU, one after the other, and print the per-shard times and the predicted N-GPU makespan (a PREDICTION, "
                         "not a multi-GPU measurement)")
    ap.add_argument("--extra", action="store_true",
                    help="also time BASELINE.json configs[3] (one GPU's share: 250 x 2000 gene pairs, 500k cells, 1k bootstraps) and "
                         "configs[4] (Perturb-seq: 200k x 15k, 500 guides vs control, 5k bootstraps) and add pair_tests_per_s / "
                         "vs_control_tests_per_s to the JSON line (N = 1 only; about two more minutes)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: the same 20k-gene matrix split over the ranks (configs[2]); weak: a full-size gene shard per rank")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.num_cpus <= 0:
        args.num_cpus = max(1, min(16, (os.cpu_count() or 1) // world))
    dev_idx = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_idx)
    comm = None
    if world > 1:
        import torch.distributed as dist

        # nccl (= RCCL over xGMI) in production; MM_DIST_BACKEND=gloo rehearses the multi-rank path on one GPU
        backend = os.environ.get("MM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_idx))
        else:
            dist.init_process_group(backend=backend)
        from scrna_parameter_estimation_amd.dist import Comm

        comm = Comm(device="cuda" if backend == "nccl" else "cpu")

    from scrna_parameter_estimation_amd import AnnDataLite, engine, memento
    from scrna_parameter_estimation_amd import _lib
    import scipy.sparse as sp

    _lib.load(require_gpu=True)
    N, G, B = cfg["cells"], cfg["genes"], cfg["num_boot"]
    n_groups = cfg["n_cond"] * cfg["n_rep"]
    if args.predict_shards:
        return predict_shards(args, cfg, torch)

    # ---- untimed preparation: data in HBM, size factors, groups + count blocks -------------------
    t0 = time.time()
    G_total = G
    mine = None
    if args.scaling == "strong" and world > 1:
        # the same matrix on any world size: this rank generates the columns of ITS cost-balanced gene shard (dist.py: LPT over
        # the predicted chain cost of every gene, here from the calibrated expression profile every rank can compute)
        mine = balanced_shard(cfg, rank, world)
        csr = synth_device_csr(cfg, 20250117, torch, genes=mine)
        G = len(mine)
        var = pd.DataFrame(index=[f"g{i}" for i in mine])
    else:
        csr = synth_device_csr(cfg, 20250117 + 1000 * rank, torch)  # weak: every rank its own full-size gene shard (same cells)
        var = pd.DataFrame(index=[f"r{rank}g{i}" for i in range(G)])
    rng = np.random.default_rng(20250117)                            # identical cell metadata on all ranks
    grp = rng.integers(0, n_groups, size=N)
    obs = pd.DataFrame({"cond": grp // cfg["n_rep"], "rep": grp % cfg["n_rep"], "q": np.full(N, 0.07)})
    Xstub = sp.csr_matrix((N, G), dtype=np.float32)                   # shape only: the counts live in HBM
    adata = AnnDataLite(Xstub, obs, var)
    memento.setup_memento(adata, q_column="q", device_csr=csr, comm=comm, shard=mine if mine is not None else False)
    memento.create_groups(adata, label_columns=["cond", "rep"])
    gdf = memento.get_groups(adata)
    cov = pd.DataFrame({"intercept": np.ones(len(gdf))}, index=gdf.index)
    trt = pd.DataFrame({"cond": (gdf["cond"].astype(int) == cfg["n_cond"] - 1).astype(float)}, index=gdf.index)
    torch.cuda.synchronize()
    prep_s = time.time() - t0
    state = adata.uns["memento"]["_hip"]
    full_idx = state.gene_idx.copy()

    def step(seed):
        # compute_1d_moments filters genes in place; restore the full gene set so every step does the same work
        state.gene_idx = full_idx.copy()
        state.var_names = None
        for k in ("size_factor", "approx_size_factor", "all_approx_size_factor"):
            adata.uns["memento"].pop(k, None)
        for g in adata.uns["memento"]["groups"]:
            adata.uns["memento"]["group_cells"][g].shape = (adata.uns["memento"]["group_cells"][g].shape[0], G)
        np.random.seed(seed)
        memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
        memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=B, num_cpus=args.num_cpus, verbose=0,
                              resampling="bootstrap", approx=False)
        # multi-GPU: ht_1d_moments ends with the gather, so "1d_ht" holds the tests of ALL ranks; count this rank's own genes
        return len(state.gene_idx) * trt.shape[1]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        step(100 + w)
    barrier()
    t0 = time.time()
    n_tests = 0
    for k in range(args.steps):
        n_tests += step(1000 + k)
    barrier()
    elapsed = time.time() - t0
    if world > 1:
        cdev = comm.device
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        nt = torch.tensor([float(n_tests)], dtype=torch.float64, device=cdev)
        torch.distributed.all_reduce(nt, op=torch.distributed.ReduceOp.SUM)
        elapsed, n_tests = float(tt.item()), float(nt.item())

    # ---- roofline of the dominant HBM kernel (K1 moments) and bootstrap draw rate, rank 0 --------
    roof = roof_csr = boot = cpu = None
    if rank == 0:
        blocks = state.blocks
        stream = engine._stream()
        d_inv = engine.dev((1.0 / adata.obs["memento_size_factor"].values)[blocks.cell_order])
        import ctypes

        timer = ctypes.c_void_p()
        _lib.call("mm_timer_create", ctypes.byref(timer))
        # The launch time depends on the clock / power state: the first ~40 launches after an idle or compute-bound phase run
        # 5-20 % slower (tools/k1_warmup.py: 0.53, 0.45, 0.442, 0.442, ... ms per group of 20).  The roofline figure is the
        # SUSTAINED rate: 100 untimed launches (45 ms), then three timed passes of 20 back-to-back launches, median pass.
        for _ in range(100):
            blocks.launch_moments(d_inv)
        reps = 20
        ms = ctypes.c_float()
        passes = []
        for _ in range(3):
            _lib.call("mm_timer_begin", timer, stream)
            for _ in range(reps):
                blocks.launch_moments(d_inv)
            _lib.call("mm_timer_end", timer, stream)
            _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms))
            passes.append(ms.value / reps)
        k1_ms = sorted(passes)[1]
        nbytes = blocks.moments_bytes()
        achieved = nbytes / (k1_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if args.config == "C3" and world == 1 and os.path.exists(tfile):
            # PMC counters cannot be read from inside this process: RECORDED figure of this round, collected with
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 x2 read correction) on the same shape and
            # kernel by tools/k1_traffic.py -- see profiles/README.md.  null when this round has no such file.
            traffic = json.load(open(tfile))["hbm_bytes_per_launch"]
            traffic_src = "recorded: profiles/" + TRAFFIC_FILE
        roof = {"kernel": "k_moments1d_sell", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_launch": int(nbytes), "bytes_are": "this layout's (4 B per stored entry); see roofline_csr for SURVEY 8(d)'s CSR bytes",
                "ms_per_launch": round(k1_ms, 4), "ms_per_launch_passes": [round(x, 4) for x in passes], "nnz": int(blocks.nnz_sel),
                "note": "sustained rate of THIS process (100 untimed launches, median of three 20-launch passes); the plateau differs "
                        "by up to ~12 % between processes/boxes with identical code (profiles/README.md)"}
        # SURVEY 8(d)'s unit of work: moments of one grouping straight from the CSR = ingest (K0) + K1, against the CSR's bytes
        k0 = {}
        tmp_blocks = engine.CountBlocks(state.csr, state.group_id, n_groups, timing=k0)
        del tmp_blocks
        k0_ms = sum(k0.values())
        nnz = int(state.csr.nnz)
        csr_bytes = nnz * 8 + (N + 1) * 4 + N * 8 + n_groups * G * 5 * 8
        t_ms = k0_ms + k1_ms
        roof_csr = {"kernels": "K0 ingest, once per grouping (k_sell_split + k_sell_count_ranges + k_sell_layout + k_sell_scatter_tiles) + k_moments1d_sell", "bound": "hbm",
                    "achieved": round(csr_bytes / (t_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(csr_bytes / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "bytes": int(csr_bytes),
                    "ms": round(t_ms, 3), "ms_parts": dict({k: round(v, 3) for k, v in k0.items()}, mm_moments1d_sell=round(k1_ms, 4)),
                    "note": "every later moments pass of the same grouping (setup x2, compute_1d_moments, each bench step) costs K1 only"}
        bs = state.last_bootstrap
        # time one replay launch by itself (HIP events on the launch stream)
        skip = ~((bs.K >= 2))
        r = np.random.default_rng(0).random((2, bs.n_pairs))
        _lib.call("mm_timer_begin", timer, stream)
        bs.run(skip, r[0], r[1], adata.uns["memento"]["mv_regressor"]["all"])
        _lib.call("mm_timer_end", timer, stream)
        _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms))
        draws = bs.draws_per_replicate * B
        chain_steps = int(np.maximum(bs.K[bs.chain_pairs] - 1, 0).sum()) if bs.n_chain else 0      # chains that run one per wave
        wave_steps = int(bs.tile_ptr[-1]) + chain_steps
        boot = {"kernel": "k_boot1d_replay (lane-per-chain tiles; chains alone in their tile run wave-uniform: chain_body) + order, fill",
                "binomial_draws": int(draws), "ms": round(ms.value, 2),
                "draws_per_s": round(draws / (ms.value * 1e-3), 1), "pairs": int((~skip).sum()), "waves": int(bs.n_tiles),
                "chain_waves": int(bs.n_chain), "wave_steps_per_replicate": wave_steps,
                "lane_occupancy": round(bs.draws_per_replicate / max(1, wave_steps * 64), 4),
                "lane_occupancy_is": "useful draws / (wave-steps x 64); a chain wave counts one useful lane per step",
                "K_max": int(bs.K.max()), "K_mean": round(float(bs.K[~skip].mean()), 1), "rng": "numpy-PCG64-replay"}
        rs = getattr(state, "refill_stats", None)
        if rs and rs["chains"]:
            # timed mode = strict=False: invalid replicates refilled on the device; every chain WITHOUT a refill is bit-identical
            # to the strict (reference-stream) path, which is the mode pinned to the reference's p-values
            boot["refilled_chain_frac"] = round(rs["chains_refilled"] / rs["chains"], 6)
            boot["refilled_gene_frac"] = round(rs["genes_refilled"] / max(1, rs["genes"]), 6)
        # fast mode (own RNG streams, replicate-parallel): reported beside the headline, never as `value`
        _lib.call("mm_timer_begin", timer, stream)
        bs.run(skip, r[0], r[1], adata.uns["memento"]["mv_regressor"]["all"], fast=True)
        _lib.call("mm_timer_end", timer, stream)
        _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms))
        boot["fast_mode_ms"] = round(ms.value, 2)
        boot["fast_mode_draws_per_s"] = round(draws / (ms.value * 1e-3), 1)
        _lib.call("mm_timer_destroy", timer)

        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, cfg, adata, csr, state, cov, trt, torch, last_seed=1000 + args.steps - 1)

    e2e = None
    if rank == 0 and world == 1:
        # SURVEY 8(d)'s metric to the letter: the step PLUS the upload of the CSR from pinned host memory and create_groups (K0
        # ingest) once per step -- value stays "inputs resident in HBM", value_e2e is printed beside it
        import ctypes

        c = state.csr
        host_parts = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t) for t in (c.indptr, c.indices, c.data)]
        torch.cuda.synchronize()
        t0 = time.time()
        up = [h.to("cuda", non_blocking=True) for h in host_parts]
        torch.cuda.synchronize()
        h2d_s = time.time() - t0
        del up, host_parts
        k0 = {}
        t0 = time.time()
        tmp = engine.CountBlocks(state.csr, state.group_id, n_groups, timing=k0)
        torch.cuda.synchronize()
        ingest_s = time.time() - t0
        del tmp
        step_s = elapsed / args.steps
        e2e = {"value_e2e": round((n_tests / args.steps) / (step_s + h2d_s + ingest_s), 2), "h2d_ms": round(1000 * h2d_s, 1),
               "ingest_ms": round(1000 * ingest_s, 1), "ingest_kernels_ms": round(sum(k0.values()), 2),
               "csr_bytes": int(c.nbytes), "h2d_GBps": round(c.nbytes / h2d_s / 1e9, 1),
               "what": "gene-tests / (step + pinned-host -> HBM upload of the CSR + create_groups' ingest), per step; results reach the host inside the step"}
    if rank == 0:
        value = n_tests / elapsed
        metric = ("gene-tests/sec (1D moments + 10k bootstraps), 1M\u00d720k sparse, 1/2/4/8 GPU" if args.config == "C3"
                  else "gene-tests/sec (1D moments + bootstrap hypothesis test)")
        line = {
            "metric": metric, "value": round(value, 2), "unit": "gene-tests/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 2),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {N} cells x {G_total if args.scaling == 'strong' else G} genes"
                                   f"{' in total' if args.scaling == 'strong' else '/GPU'}, {cfg['density']:.0%} nnz, {n_groups} groups, "
                                   f"{B} bootstraps, 1D moments + ht (resampling=bootstrap, approx=False)",
                       "rng": "numpy PCG64 multinomial replay; device refill of invalid replicates", "parallelism": f"genes x{world} (cost-balanced shards)" if world > 1 else "1 GPU",
                       "host_tail_fit_procs": args.num_cpus, "prep_s": round(prep_s, 2)},
            "roofline": roof, "roofline_csr": roof_csr, "bootstrap": boot, "cpu_baseline": cpu, "reference_cpu": reference_cpu(args.config),
            "e2e": e2e,
        }
        if e2e:
            line["value_e2e"] = e2e["value_e2e"]
        if args.extra and world == 1:
            state.last_bootstrap = None
            del csr, adata, state
            torch.cuda.empty_cache()
            line.update(extra_configs(torch))
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def extra_configs(torch):
    """--extra: the other two GPU configurations of BASELINE.json on one GPU, as tools/bench_2d.py / tools/bench_vs_control.py run
    them (tests/test_gpu_configs.py holds their parity checks): configs[3]'s per-GPU share of the 2000 x 2000 pair grid and the
    whole of configs[4]."""
    import scipy.sparse as sp

    from scrna_parameter_estimation_amd import AnnDataLite, memento

    out = {}
    # configs[3]: 250 x 2000 = 500,000 pairs (one of eight GPUs' share), 500k cells, 2 groups, 1,000 bootstraps
    cells, genes, B = 500_000, 8_000, 1_000
    csr = synth_device_csr(dict(cells=cells, genes=genes, density=0.08), 3, torch)
    grp = np.random.default_rng(1).integers(0, 2, size=cells)
    adata = AnnDataLite(sp.csr_matrix((cells, genes), dtype=np.float32), pd.DataFrame({"cond": grp, "q": np.full(cells, 0.07)}),
                        pd.DataFrame(index=[f"g{i}" for i in range(genes)]))
    memento.setup_memento(adata, q_column="q", device_csr=csr)
    memento.create_groups(adata, label_columns=["cond"])
    memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
    names = memento.main._var_names(adata)
    pairs = [(a, b) for a in names[:250] for b in names[250:2250]]
    gdf = memento.get_groups(adata)
    cov = pd.DataFrame({"intercept": np.ones(len(gdf))}, index=gdf.index)
    trt = pd.DataFrame({"cond": gdf["cond"].astype(float)}, index=gdf.index)
    torch.cuda.synchronize()
    t0 = time.time()
    memento.compute_2d_moments(adata, pairs)
    np.random.seed(12)
    memento.ht_2d_moments(adata, covariate=cov, treatment=trt, num_boot=B, num_cpus=8, verbose=0, resampling="bootstrap", approx=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    out["pair_tests_per_s"] = round(len(pairs) / dt, 1)
    out["pair_tests_config"] = f"configs[3], one GPU's share: {len(pairs)} pairs, {cells} cells, 2 groups, {B} bootstraps, {dt:.1f} s"
    adata.uns["memento"]["_hip"].last_bootstrap2d = None
    del adata, csr
    torch.cuda.empty_cache()
    # configs[4]: 200k cells x 15k genes, 500 guides + control (20 % of the cells), 5,000 bootstraps, one batched call
    cells, genes, n_guides, B = 200_000, 15_000, 500, 5_000
    csr = synth_device_csr(dict(cells=cells, genes=genes, density=0.05), 20250117 + 5, torch)
    rng = np.random.default_rng(20250117 + 5)
    is_ctrl = rng.random(cells) < 0.2
    guide = np.where(is_ctrl, 0, 1 + rng.integers(0, n_guides, size=cells))
    adata = AnnDataLite(sp.csr_matrix((cells, genes), dtype=np.float32), pd.DataFrame({"guide": guide, "q": np.full(cells, 0.07)}),
                        pd.DataFrame(index=[f"g{i}" for i in range(genes)]))
    memento.setup_memento(adata, q_column="q", device_csr=csr)
    memento.create_groups(adata, label_columns=["guide"])
    memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
    ctrl = [g for g in adata.uns["memento"]["groups"] if g.split("^")[-1] == "0"][0]
    np.random.seed(0)
    torch.cuda.synchronize()
    t0 = time.time()
    df = memento.ht_1d_vs_control(adata, control=ctrl, num_boot=B, num_cpus=16, approx=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    out["vs_control_tests_per_s"] = round(len(df) / dt, 1)
    out["vs_control_config"] = f"configs[4]: {len(df)} (gene, guide) tests, {cells} cells x {genes} genes, {n_guides} guides + control, {B} bootstraps, {dt:.1f} s"
    return out


def predict_shards(args, cfg, torch):
    """--predict-shards N: the step of each of the N cost-balanced gene shards, timed alone on this GPU (single process, no
    collectives; the shard runs with the size factors of the whole matrix, its own pooled mean-variance fit).  The N-GPU strong-
    scaling run does this concurrently, so its step takes about the longest shard's time plus the small exchanges: a PREDICTION
    of the makespan, printed as such -- not a multi-GPU measurement."""
    import scipy.sparse as sp

    from scrna_parameter_estimation_amd import AnnDataLite, memento

    n = args.predict_shards
    N, G, B = cfg["cells"], cfg["genes"], cfg["num_boot"]
    n_groups = cfg["n_cond"] * cfg["n_rep"]
    csr = synth_device_csr(cfg, 20250117, torch)
    grp = np.random.default_rng(20250117).integers(0, n_groups, size=N)
    obs = pd.DataFrame({"cond": grp // cfg["n_rep"], "rep": grp % cfg["n_rep"], "q": np.full(N, 0.07)})
    full = AnnDataLite(sp.csr_matrix((N, G), dtype=np.float32), obs.copy(), pd.DataFrame(index=[f"g{i}" for i in range(G)]))
    memento.setup_memento(full, q_column="q", device_csr=csr)
    sf = full.obs["memento_size_factor"].values.copy()
    shards = []
    for r in range(n):
        mine = balanced_shard(cfg, r, n)
        sub = csr.colselect(mine)
        adata = AnnDataLite(sp.csr_matrix((N, len(mine)), dtype=np.float32), obs.copy(), pd.DataFrame(index=[f"g{i}" for i in mine]))
        memento.setup_memento(adata, q_column="q", device_csr=sub, size_factor=sf)
        memento.create_groups(adata, label_columns=["cond", "rep"])
        gdf = memento.get_groups(adata)
        cov = pd.DataFrame({"intercept": np.ones(len(gdf))}, index=gdf.index)
        trt = pd.DataFrame({"cond": (gdf["cond"].astype(int) == cfg["n_cond"] - 1).astype(float)}, index=gdf.index)
        state = adata.uns["memento"]["_hip"]
        full_idx = state.gene_idx.copy()

        def step(seed):
            state.gene_idx = full_idx.copy()
            state.var_names = None
            for k in ("size_factor", "approx_size_factor", "all_approx_size_factor"):
                adata.uns["memento"].pop(k, None)
            for g in adata.uns["memento"]["groups"]:
                adata.uns["memento"]["group_cells"][g].shape = (adata.uns["memento"]["group_cells"][g].shape[0], len(mine))
            np.random.seed(seed)
            memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
            memento.ht_1d_moments(adata, covariate=cov, treatment=trt, num_boot=B, num_cpus=args.num_cpus, verbose=0,
                                  resampling="bootstrap", approx=False)
            return len(state.gene_idx) * trt.shape[1]

        for w in range(args.warmup):
            step(100 + w)
        torch.cuda.synchronize()
        t0 = time.time()
        tests = 0
        for k in range(args.steps):
            tests += step(1000 + k)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.steps
        bs = state.last_bootstrap
        shards.append({"rank": r, "genes": int(len(mine)), "tests_per_step": int(tests // args.steps), "ms_per_step": round(1000 * dt, 1),
                       "chains": int((bs.K >= 2).sum()), "K_max": int(bs.K.max()), "chain_waves": int(bs.n_chain), "tile_waves": int(bs.n_tiles),
                       "draws_per_replicate": int(bs.draws_per_replicate)})
        print(json.dumps({"shard": shards[-1]}), file=sys.stderr, flush=True)
        state.last_bootstrap = None
        del adata, sub, state, bs
        torch.cuda.empty_cache()
    makespan = max(s_["ms_per_step"] for s_ in shards)
    total = sum(s_["tests_per_step"] for s_ in shards)
    print(json.dumps({"metric": "PREDICTED strong-scaling makespan from per-shard step times on ONE GPU (not a multi-GPU measurement)",
                      "config": args.config, "n_shards": n, "sharding": "dist.shard_genes_balanced (LPT over predicted chain cost)",
                      "shards": shards, "predicted_ms_per_step": makespan, "predicted_value": round(total / (makespan * 1e-3), 2),
                      "unit": "gene-tests/s", "steps": args.steps, "warmup": args.warmup,
                      "note": "each shard alone on the GPU, whole-matrix size factors, no collectives; the N-rank run adds one "
                              "all-gather of the pooled-fit inputs and the result gather per step"}), flush=True)


def reference_cpu(config):
    """The REAL reference's own CPU timing on this shape, recorded in the build container by tools/ref_cpu_baseline.py (the
    reference cannot travel to the GPU box): other hardware, stated as such, printed beside cpu_baseline."""
    f = os.path.join(ROOT, "baselines", f"ref_cpu_{config}.json")
    if not os.path.exists(f):
        return None
    try:
        return json.load(open(f))
    except Exception:
        return None


_CPU = {}


def _cpu_init(shared):
    """Pool initializer: static inputs of the CPU baseline (group membership, size factors, design) arrive once."""
    sys.path.insert(0, ROOT)
    _CPU.update(shared)


def _cpu_gene(job):
    """One gene of the CPU baseline: its share of the moments pass + the oracle's _ht_1d (bootstrap, regression, ASL).
    ``skip`` = uniforms the global np.random stream has handed out before this gene's first chain in the GPU run being
    compared with (2 per live chain, gene-major): the oracle then orders its bins with the very hash uniforms the GPU used, so
    its coefficients / standard errors / p-values are comparable value for value (returned beside the time)."""
    from oracle import memento_oracle as orc

    col32, tm_g, trv_g, seed, skip = job
    c = _CPU
    col = col32.astype(np.float64)
    np.random.seed(seed)
    if skip:
        np.random.random(int(skip))
    t0 = time.time()
    for k in range(len(c["sel"])):          # moments part of the step for this gene (estimator.py:177-183)
        x = col[c["sel"][k]]
        w = 1.0 / c["sf"][c["sel"][k]]
        _ = ((x * w).sum(), (x * x * w * w).sum(), (x * w * w).sum())
    res = orc.ht_1d_gene(tm_g, trv_g, [col[s] for s in c["sel"]], c["asf"], c["cov"], c["trt"], c["Nc"], c["num_boot"], c["fit"], c["gq"],
                         resampling="bootstrap", approx=False)
    return time.time() - t0, [np.atleast_1d(np.asarray(r, dtype=np.float64)) for r in res]


def stream_skips(true_mean, true_rv):
    """Per kept gene: how many uniforms of the global np.random stream ht_1d_moments (strict=False) has consumed before the
    gene's first chain -- two per live (gene, group) pair, gene-major (memento/bootstrap.py:62, :65; hypothesis_test.py:167-171
    decides which pairs are live).  ``true_mean`` / ``true_rv``: [n_groups][genes]."""
    with np.errstate(invalid="ignore"):
        live = ~(np.isnan(true_mean) | np.isnan(true_rv) | (true_mean == 0) | (true_rv < 0))
    per_gene = live.sum(axis=0)
    return 2 * (np.cumsum(per_gene) - per_gene)


def cpu_baseline(args, cfg, adata, csr, state, cov, trt, torch, last_seed):
    """The CPU oracle (numpy restatement pinned to the reference) on a bounded sample of genes of the same matrix:
    ``--cpu-baseline-cores`` worker processes (default: the host cores of one GPU's share, at most 16), one gene per task,
    for about ``--cpu-baseline-seconds`` of wall time."""
    m = adata.uns["memento"]
    groups = m["groups"]
    ng = len(groups)
    kept = state.gene_idx
    cores = max(1, args.cpu_baseline_cores)
    n_try = min(len(kept), max(8 * cores, int(1.5e9 / (8 * adata.shape[0]))))      # dense sample columns: <= ~1.5 GB of host memory
    pick = kept[:: max(1, len(kept) // n_try)][:n_try]
    cols = sample_columns(csr, pick, torch).astype(np.float32)
    gid = state.group_id
    sel = [np.flatnonzero(gid == k) for k in range(ng)]
    slot = {g: i for i, g in enumerate(kept)}
    tm = np.stack([m["1d_moments"][g][0] for g in groups])
    trv = np.stack([m["1d_moments"][g][2] for g in groups])
    shared = dict(sel=sel, sf=adata.obs["memento_size_factor"].values, asf=[m["all_approx_size_factor"][s] for s in sel],
                  gq=np.array([m["group_q"][g] for g in groups]), Nc=np.array([len(s) for s in sel], dtype=float), cov=cov.values,
                  trt=trt.values, num_boot=cfg["num_boot"], fit=m["mv_regressor"]["all"])
    # every sampled gene is run on the uniforms the GPU's LAST timed step gave it (seed and stream position), so the oracle's
    # outputs can be compared with the GPU's value for value
    skips = stream_skips(tm, trv)
    jobs = [(np.ascontiguousarray(cols[:, j]), tm[:, slot[g]], trv[:, slot[g]], last_seed, int(skips[slot[g]])) for j, g in enumerate(pick)]
    done, busy = 0, 0.0
    results = {}
    if cores == 1:
        _cpu_init(shared)
        t0 = time.time()
        for j, job in enumerate(jobs):
            dt_, results[j] = _cpu_gene(job)
            busy += dt_
            done += 1
            if time.time() - t0 > args.cpu_baseline_seconds:
                break
        dt = time.time() - t0
    else:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor, wait, FIRST_COMPLETED

        with ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn"), initializer=_cpu_init, initargs=(shared,)) as ex:
            list(ex.map(int, range(4 * cores)))                 # start all workers (imports, shared data) outside the timed region
            t0 = time.time()
            it = iter(enumerate(jobs))
            pending = {}
            for _ in range(min(cores, len(jobs))):
                j, job = next(it)
                pending[ex.submit(_cpu_gene, job)] = j
            while pending:
                fin, _ = wait(list(pending), return_when=FIRST_COMPLETED)
                for f in fin:
                    j = pending.pop(f)
                    dt_, results[j] = f.result()
                    busy += dt_
                    done += 1
                    if time.time() - t0 < args.cpu_baseline_seconds:      # keep every worker busy until the budget is spent
                        nxt = next(it, None)
                        if nxt is not None:
                            pending[ex.submit(_cpu_gene, nxt[1])] = nxt[0]
            dt = time.time() - t0
    # the oracle's outputs for the sampled genes against the GPU's last timed step (same matrix, same hash uniforms; genes
    # with a device-refilled replicate are left out: there the timed mode is statistically, not numerically, the reference)
    ht = m["1d_ht"]
    refilled = state.refill_stats["gene_refilled"]
    nt = trt.shape[1]
    diffs = {k: 0.0 for k in ("mean_coef", "mean_se", "mean_asl", "var_coef", "var_se", "var_asl")}
    compared = 0
    for j, res in results.items():
        gi = slot[pick[j]]
        if refilled[gi]:
            continue
        compared += 1
        for k, r in zip(diffs, res):
            got, want = ht[k][gi * nt:(gi + 1) * nt], r * np.ones(nt)
            ok = np.isfinite(want) | np.isfinite(got)
            if ok.any():
                with np.errstate(invalid="ignore", divide="ignore"):
                    d = np.abs(got[ok] - want[ok]) / np.maximum(np.abs(want[ok]), 1e-300)
                diffs[k] = max(diffs[k], float(np.nan_to_num(d, nan=np.inf).max()))
    return {"value": round(done * trt.shape[1] / dt, 4), "unit": "gene-tests/s", "cores": cores, "kind": "port",
            "per_core": round(done * trt.shape[1] / max(busy, 1e-9), 4),
            "genes_compared_with_gpu": compared, "max_rel_p_diff": max(diffs["mean_asl"], diffs["var_asl"]),
            "max_rel_diff": {k: float(f"{v:.3g}") for k, v in diffs.items()},
            "sample": f"{done} genes x {ng} groups x {cfg['num_boot']} bootstraps of the same matrix, oracle/memento_oracle.py, "
                      f"{cores} worker processes, {dt:.1f} s wall"}


if __name__ == "__main__":
    main()
