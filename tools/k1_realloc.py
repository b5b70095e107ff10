"""Does K1's plateau follow the ALLOCATION of its buffers?  One process: the count blocks are rebuilt several times (new entry /
slab / table allocations each time, earlier ones kept alive so the memory really is different) and K1 is timed on each.
usage: python tools/k1_realloc.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from scrna_parameter_estimation_amd import engine, _lib

cells, genes, dens, groups = 1_000_000, 20_000, 0.03, 20
csr = bench.synth_device_csr(dict(cells=cells, genes=genes, density=dens), 20250117, torch)
gid = np.random.default_rng(20250117).integers(0, groups, size=cells).astype(np.int32)
timer = ctypes.c_void_p(); _lib.call("mm_timer_create", ctypes.byref(timer)); ms = ctypes.c_float()
inv = np.random.default_rng(1).lognormal(0, .3, size=cells)


def time_k1(blocks, d_inv, n=40):
    stream = engine._stream()
    for _ in range(80):
        blocks.launch_moments(d_inv)
    out = []
    for _ in range(3):
        _lib.call("mm_timer_begin", timer, stream)
        for _ in range(n):
            blocks.launch_moments(d_inv)
        _lib.call("mm_timer_end", timer, stream)
        _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms))
        out.append(ms.value / n)
    torch.cuda.synchronize()
    return sorted(out)[1]


keep = []
for i in range(6):
    blocks = engine.CountBlocks(csr, gid, groups)
    d_inv = engine.dev(inv[blocks.cell_order])
    print("build %d: ent at 0x%x, slab at 0x%x: %.4f ms" % (i, blocks.ent.data_ptr(), blocks._slab.data_ptr(), time_k1(blocks, d_inv)), flush=True)
    keep.append((blocks, d_inv))
    if i == 2:
        keep.append(torch.empty(3_000_000_000, dtype=torch.uint8, device="cuda"))   # shift what comes next
b0, d0 = keep[0]
print("build 0 again: %.4f ms" % time_k1(b0, d0), flush=True)
