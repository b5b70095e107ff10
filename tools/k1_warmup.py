"""How does the K1 launch time evolve with sustained launching in one process (clock / power ramp)?  HIP events over groups of
20 launches, same buffers."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from scrna_parameter_estimation_amd import engine, _lib
cfg = dict(cells=1_000_000, genes=20_000, density=0.03)
csr = bench.synth_device_csr(cfg, 1, torch)
gid = np.random.default_rng(0).integers(0, 20, size=cfg["cells"]).astype(np.int32)
timer = ctypes.c_void_p(); _lib.call("mm_timer_create", ctypes.byref(timer)); s = engine._stream()
blocks = engine.CountBlocks(csr, gid, 20)
d_inv = engine.dev(np.random.default_rng(1).lognormal(0, .3, size=cfg["cells"])[blocks.cell_order])
torch.cuda.synchronize()
res = []
for p in range(60):
    _lib.call("mm_timer_begin", timer, s)
    for _ in range(20): blocks.launch_moments(d_inv)
    _lib.call("mm_timer_end", timer, s)
    ms = ctypes.c_float(); _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms)); res.append(ms.value / 20)
print("K1 ms per launch, consecutive groups of 20 launches:", [round(x, 4) for x in res], flush=True)
time.sleep(3.0)     # idle, then again
res = []
for p in range(10):
    _lib.call("mm_timer_begin", timer, s)
    for _ in range(20): blocks.launch_moments(d_inv)
    _lib.call("mm_timer_end", timer, s)
    ms = ctypes.c_float(); _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms)); res.append(ms.value / 20)
print("after 3 s idle:", [round(x, 4) for x in res], flush=True)
