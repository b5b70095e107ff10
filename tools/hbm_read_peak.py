"""What does a plain streaming read reach on this box?  (context for K1's roofline fraction)
torch reductions / copies over a 2.5 GB buffer, HIP-event timed."""
import torch, time
n = 2_532_000_000 // 4
x = torch.ones(n, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
def timeit(f, reps=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = timeit(lambda: x.sum())
print(f"torch.sum fp32 {x.numel()*4/1e9:.2f} GB: {ms:.3f} ms -> {x.numel()*4/ms/1e6:.0f} GB/s read")
xi = x.view(torch.int32)
ms = timeit(lambda: xi.max())
print(f"torch.max int32: {ms:.3f} ms -> {x.numel()*4/ms/1e6:.0f} GB/s read")
ms = timeit(lambda: y.copy_(x))
print(f"copy: {ms:.3f} ms -> {2*x.numel()*4/ms/1e6:.0f} GB/s read+write")
ms = timeit(lambda: y.fill_(0.0))
print(f"fill: {ms:.3f} ms -> {x.numel()*4/ms/1e6:.0f} GB/s write")

# the K1 access pattern without its arithmetic (mm_debug_read_probe), HIP-event timed on the current stream
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_parameter_estimation_amd import _lib, engine
sink = torch.zeros((x.numel() * 4 // 65536) * 64 * 8 + 16, dtype=torch.int32, device="cuda")     # mode 3 writes 32 B per (chunk, lane)
nbytes = (x.numel() * 4 // 65536) * 65536
timer = ctypes.c_void_p(); _lib.call("mm_timer_create", ctypes.byref(timer)); st = engine._stream()
xr = torch.randint(0, 2 ** 31 - 1, (x.numel(),), dtype=torch.int32, device="cuda")      # random cell indices / counts for modes 1, 2
for mode, what in ((0, "loads only"), (1, "+ LDS gather"), (2, "+ LDS gather + fp64 math"), (3, "+ gather + math + result stores")):
    for wgs in (256, 2048):
        for _ in range(3):
            _lib.call("mm_debug_read_probe", engine.P(xr), nbytes, wgs, mode, engine.P(sink), st)
        _lib.call("mm_timer_begin", timer, st)
        for _ in range(20):
            _lib.call("mm_debug_read_probe", engine.P(xr), nbytes, wgs, mode, engine.P(sink), st)
        _lib.call("mm_timer_end", timer, st)
        t = ctypes.c_float(); _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(t))
        print(f"read probe (K1 pattern, {what}), {wgs} workgroups x 1024 threads: {t.value/20:.3f} ms -> {nbytes/(t.value/20)/1e6:.0f} GB/s")
