"""Per-phase cycle stamps of k_sell_scatter_tiles (diagnostic build -DINGEST_STAMPS; run through tools/ingest_stamps.sh)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from scrna_parameter_estimation_amd import engine, _lib

cells, genes, dens, groups = 1_000_000, 20_000, 0.03, 20
csr = bench.synth_device_csr(dict(cells=cells, genes=genes, density=dens), 20250117, torch)
gid = np.random.default_rng(20250117).integers(0, groups, size=cells).astype(np.int32)
lib = _lib.load()
f = lib.mm_debug_ingest_stamps
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
for rd in range(2):
    f(None, 1)
    t = {}
    blocks = engine.CountBlocks(csr, gid, groups, timing=t)
    f(out, 0)
    v = np.array(list(out), dtype=np.float64)
    tiles = v[10]          # wave-tiles
    names = ["bounds+T", "P1 loads+masks", "barrier1", "P2 rest", "P3 place", "barrier3", "P4 store", "barrier4", "P5+barrier", "loop"]
    print({k: round(x, 3) for k, x in t.items()})
    extra = {"issue next loads": int(v[11] / tiles), "P2 popcounts+scan": int(v[12] / tiles), "barrier2": int(v[13] / tiles)}
    print("wave-tiles", int(tiles), "cycles per wave-tile:", {n: int(v[i] / tiles) for i, n in enumerate(names)}, extra, "sum", int((v[:10].sum() + v[11:14].sum()) / tiles))
    del blocks
