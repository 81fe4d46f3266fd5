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

# the library's own block-contiguous pattern (one workgroup per contiguous chunk, 16 B per lane)
import ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raptor_amd", "libraptor_hip.so"))
lib.raptor_probe_fill.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
for chunk in (8624, 8628, 8624 * 14, 1584, 4752, 924, 132):
    blocks = n // chunk
    ms = t(lambda: lib.raptor_probe_fill(x.data_ptr(), blocks, chunk, st))
    print(f"probe chunk {chunk*8/1024:.0f} KB x {blocks} blocks: {ms:.3f} ms  {blocks*chunk*8/ms/1e6:.0f} GB/s")

# the Digit expand kernel's own pattern (scattered pieces) against the same bytes in address order: 4096 z x 7 workgroups
zs = n // (1005 * 132)
for mode, name in ((-1, "expand pattern (3 pieces + 130 scattered rows per workgroup)"), (-2, "same bytes, address order"),
                   (-3, "3 pieces per workgroup, customized rows as one contiguous run per workgroup")):
    ms = t(lambda: lib.raptor_probe_fill(x.data_ptr(), zs * 7, mode, st))
    print(f"probe {name}: {ms:.3f} ms  {zs*1005*132*8/ms/1e6:.0f} GB/s")

# grid-stride order (dense moving window) with few or many workgroups
for blocks in (2048, 8192, 32768):
    ms = t(lambda: lib.raptor_probe_fill(x.data_ptr(), blocks, -4, st))
    print(f"probe grid-stride, {blocks} workgroups x {8192*8//1024} KB total each: {ms:.3f} ms  {blocks*8192*8/ms/1e6:.0f} GB/s")

# wider contiguous runs per lane (64 KB per workgroup)
for mode, name in ((-6, "32 B per lane"), (-5, "64 B per lane")):
    blocks = n // 8192
    ms = t(lambda: lib.raptor_probe_fill(x.data_ptr(), blocks, mode, st))
    print(f"probe {name}, 64 KB chunks: {ms:.3f} ms  {blocks*8192*8/ms/1e6:.0f} GB/s")
ms = t(lambda: lib.raptor_probe_fill(x.data_ptr(), n // 8192, 8192, st))
print(f"probe 16 B per lane, 64 KB chunks: {ms:.3f} ms  {(n//8192)*8192*8/ms/1e6:.0f} GB/s")
