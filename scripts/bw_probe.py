"""Achievable HBM write / copy bandwidth on this GPU with plain torch kernels (calibration for the roofline numbers)."""
import torch, time
n = 3_300_000_000 // 8
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n // 2, dtype=torch.float64, device="cuda")
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = t(lambda: x.fill_(1.0)); print(f"fill  {n*8/1e9:.2f} GB  {ms:.3f} ms  {n*8/ms/1e6:.0f} GB/s written")
ms = t(lambda: x.zero_()); print(f"zero  {n*8/1e9:.2f} GB  {ms:.3f} ms  {n*8/ms/1e6:.0f} GB/s written")
ms = t(lambda: y.copy_(x[: n // 2])); print(f"copy  {n*4/1e9:.2f} GB  {ms:.3f} ms  {n*4/ms/1e6:.0f} GB/s written (+ same read)")
ms = t(lambda: x.sum()); print(f"sum   {n*8/1e9:.2f} GB  {ms:.3f} ms  {n*8/ms/1e6:.0f} GB/s read")
