"""Micro-benchmark of the K1 moments kernel (HIP events) on a synthetic matrix already in HBM.
usage: python tools/k1_bench.py [cells genes density groups]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from scrna_parameter_estimation_amd import engine, _lib

cells, genes, dens, groups = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1_000_000, 20_000, 0.03, 20)
cfg = dict(cells=cells, genes=genes, density=dens)
t0 = time.time(); csr = bench.synth_device_csr(cfg, 1, torch); torch.cuda.synchronize(); t_gen = time.time() - t0
gid = np.random.default_rng(0).integers(0, groups, size=cells).astype(np.int32)
t0 = time.time(); blocks = engine.CountBlocks(csr, gid, groups); torch.cuda.synchronize(); t_ing = time.time() - t0
d_inv = engine.dev(np.random.default_rng(1).lognormal(0, .3, size=cells)[blocks.cell_order])
timer = ctypes.c_void_p(); _lib.call("mm_timer_create", ctypes.byref(timer)); s = engine._stream()
for _ in range(100): blocks.launch_moments(d_inv)      # sustained rate: the first ~40 launches after idle are 5-20 % slower
reps = 20
_lib.call("mm_timer_begin", timer, s)
for _ in range(reps): blocks.launch_moments(d_inv)
_lib.call("mm_timer_end", timer, s)
ms = ctypes.c_float(); _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms)); ms = ms.value / reps
nb = blocks.moments_bytes()
print(f"defs={os.environ.get('MM_EXTRA_DEFS','')!r} nnz={csr.nnz:.3e} blocks={blocks.n_blocks} items={blocks.total_items} ent={blocks.ent_bytes/1e9:.3f}GB "
      f"pad={blocks.ent_bytes/4/max(1,blocks.nnz_sel):.3f} bytes={nb/1e9:.3f}GB  K1={ms:.4f} ms  {nb/ms/1e6:.1f} GB/s  ({nb/ms/1e6/8000:.1%} of 8TB/s)  gen={t_gen:.1f}s ingest={t_ing:.2f}s", flush=True)
# clocks of this box while the kernel runs (boxes of the pool differ by ~10 % on K1): sample rocm-smi from a helper process
import subprocess, threading
def _spin():
    for _ in range(4000): blocks.launch_moments(d_inv)
    torch.cuda.synchronize()
th = threading.Thread(target=_spin); th.start(); time.sleep(0.5)
try:
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=30).stdout
    print(" | ".join(l.split(":", 1)[1].strip() if ":" in l else l for l in out.splitlines() if any(k in l for k in ("fclk", "mclk", "sclk", "Power (W)"))), flush=True)
except Exception as e:
    print("rocm-smi unavailable:", e)
th.join()
