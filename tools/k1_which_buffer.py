"""Which of K1's buffers carries the placement sensitivity?  One process, one set of count blocks; each buffer in turn is re-allocated
(fresh allocation, the old one kept alive) and refilled with the same bytes, K1 is timed after each move.
usage: python tools/k1_which_buffer.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from scrna_parameter_estimation_amd import engine, _lib

cells, genes, dens, groups = 1_000_000, 20_000, 0.03, 20
csr = bench.synth_device_csr(dict(cells=cells, genes=genes, density=dens), 20250117, torch)
gid = np.random.default_rng(20250117).integers(0, groups, size=cells).astype(np.int32)
timer = ctypes.c_void_p(); _lib.call("mm_timer_create", ctypes.byref(timer)); ms = ctypes.c_float()
blocks = engine.CountBlocks(csr, gid, groups)
hold = {"d_inv": engine.dev(np.random.default_rng(1).lognormal(0, .3, size=cells)[blocks.cell_order])}


def time_k1(n=40):
    stream = engine._stream()
    for _ in range(80):
        blocks.launch_moments(hold["d_inv"])
    out = []
    for _ in range(3):
        _lib.call("mm_timer_begin", timer, stream)
        for _ in range(n):
            blocks.launch_moments(hold["d_inv"])
        _lib.call("mm_timer_end", timer, stream)
        _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms))
        out.append(ms.value / n)
    torch.cuda.synchronize()
    return sorted(out)[1]


print("as built: %.4f ms" % time_k1(), flush=True)
graveyard = []
for name in ("ent", "_slab", "d_inv", "tables", "ent", "ent", "_slab", "ent"):
    if name == "tables":
        for t in ("blk_base", "slice_w", "slice_ptr", "item_ptr", "blk_item_base", "d_blk_cell0"):
            old = getattr(blocks, t); graveyard.append(old); setattr(blocks, t, old.clone())
    elif name == "d_inv":
        graveyard.append(hold["d_inv"]); hold["d_inv"] = hold["d_inv"].clone()
    else:
        old = getattr(blocks, name); graveyard.append(old)
        graveyard.append(torch.empty(int(np.random.default_rng(len(graveyard)).integers(1, 400)) << 20, dtype=torch.uint8, device="cuda"))
        setattr(blocks, name, old.clone())
    print("moved %-7s -> %.4f ms" % (name, time_k1()), flush=True)
