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
