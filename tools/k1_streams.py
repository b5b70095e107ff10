"""Does K1's per-process plateau depend on the HIP stream (hardware queue) it is launched on?  usage: python tools/k1_streams.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from scrna_parameter_estimation_amd import engine, _lib

cells, genes, dens, groups = 1_000_000, 20_000, 0.03, 20
csr = bench.synth_device_csr(dict(cells=cells, genes=genes, density=dens), 20250117, torch)
gid = np.random.default_rng(20250117).integers(0, groups, size=cells).astype(np.int32)
blocks = engine.CountBlocks(csr, gid, groups)
d_inv = engine.dev(np.random.default_rng(1).lognormal(0, .3, size=cells)[blocks.cell_order])
timer = ctypes.c_void_p(); _lib.call("mm_timer_create", ctypes.byref(timer)); ms = ctypes.c_float()


def time_k1(n=40):
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


print("default stream: %.4f ms" % time_k1(), flush=True)
for i in range(6):
    s = torch.cuda.Stream(priority=-1 if i % 2 else 0)
    with torch.cuda.stream(s):
        print("stream %d (priority %d): %.4f ms" % (i, -1 if i % 2 else 0, time_k1()), flush=True)
print("default stream again: %.4f ms" % time_k1(), flush=True)
