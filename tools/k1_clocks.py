"""K1's per-process plateau vs the clocks the driver reports while it runs (rocm-smi, read-only).  usage: python tools/k1_clocks.py"""
import os, subprocess, sys, ctypes, threading, time
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
samples = []


def sample():
    for _ in range(3):
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
            keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "socclk", "Power", "junction", "memory)"))]
            samples.append(keep)
        except Exception as e:
            samples.append([repr(e)])
        time.sleep(0.3)


th = threading.Thread(target=sample); th.start()
t_end = time.time() + 3.0
per = []
while time.time() < t_end:
    _lib.call("mm_timer_begin", timer, stream)
    for _ in range(40):
        blocks.launch_moments(d_inv)
    _lib.call("mm_timer_end", timer, stream)
    _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms))
    per.append(ms.value / 40)
th.join()
print("K1 ms per launch over 3 s: first %.4f median %.4f last %.4f" % (per[0], float(np.median(per)), per[-1]))
for s in samples:
    print(" | ".join(s)[:700])
