"""Does K1's per-process plateau follow the PLACEMENT of its buffers?  One process, same count blocks: the slab (written) and
the entries (read) are re-placed at several offsets inside larger allocations and K1 is timed at each placement.
usage: python tools/k1_offsets.py"""
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
stream = engine._stream()
timer = ctypes.c_void_p(); _lib.call("mm_timer_create", ctypes.byref(timer)); ms = ctypes.c_float()


def time_k1(n=40):
    for _ in range(60):
        blocks.launch_moments(d_inv)
    out = []
    for _ in range(3):
        _lib.call("mm_timer_begin", timer, stream)
        for _ in range(n):
            blocks.launch_moments(d_inv)
        _lib.call("mm_timer_end", timer, stream)
        _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms))
        out.append(ms.value / n)
    return sorted(out)[1]


print("as allocated: %.4f ms" % time_k1(), flush=True)
slab0, ent0 = blocks._slab, blocks.ent
n_slab, n_ent = slab0.numel(), ent0.numel()
pad = 96 * 1024 * 1024 // 4
big_slab = torch.empty(n_slab + pad, dtype=slab0.dtype, device="cuda")
for off_kb in (0, 4, 64, 256, 1024, 2048, 3072, 8192, 32768, 65536):
    o = off_kb * 1024 // 4
    blocks._slab = big_slab[o:o + n_slab]
    print("slab at +%6d KiB: %.4f ms" % (off_kb, time_k1()), flush=True)
blocks._slab = slab0
big_ent = torch.zeros(n_ent + pad, dtype=ent0.dtype, device="cuda")
for off_kb in (0, 4, 256, 2048, 3072, 32768):
    o = off_kb * 1024 // 4
    big_ent[o:o + n_ent].copy_(ent0)
    blocks.ent = big_ent[o:o + n_ent]
    print("entries at +%6d KiB: %.4f ms" % (off_kb, time_k1()), flush=True)
# fresh allocations after churning the allocator
blocks.ent = ent0
junk = [torch.empty(int(x) * 1024 * 1024 // 4, dtype=torch.int32, device="cuda") for x in (100, 700, 33, 1500, 256)]
for i in range(4):
    blocks._slab = torch.empty(n_slab, dtype=slab0.dtype, device="cuda")
    print("fresh slab %d (ptr %% 2MiB = %d KiB): %.4f ms" % (i, blocks._slab.data_ptr() % (2 << 20) // 1024, time_k1()), flush=True)
    junk.append(blocks._slab); junk.append(torch.empty((37 + 61 * i) * 1024 * 256, dtype=torch.int32, device="cuda"))
