"""K0 ingest timing at BASELINE configs[2] shape (HIP events per launch), three rounds.  usage: python tools/ingest_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from scrna_parameter_estimation_amd import engine

cells, genes, dens, groups = 1_000_000, 20_000, 0.03, 20
csr = bench.synth_device_csr(dict(cells=cells, genes=genes, density=dens), 20250117, torch)
gid = np.random.default_rng(20250117).integers(0, groups, size=cells).astype(np.int32)
for rd in range(3):
    t = {}
    blocks = engine.CountBlocks(csr, gid, groups, timing=t)
    print({k: round(v, 3) for k, v in t.items()}, "total", round(sum(t.values()), 3), "ms; ranged", blocks.ranged, flush=True)
    del blocks
