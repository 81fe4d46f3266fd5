"""Write rate of the Digit expansion's byte pattern from a persistent grid (scripts/bw_probe.py has the one-item-per-workgroup forms)."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raptor_amd", "libraptor_hip.so"))
lib.raptor_probe_fill.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p]
n = 4096 * 1005 * 132
x = torch.empty(n, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
nbytes = 4096 * (1005 - 91) * 132 * 8
for mode, blocks, name in ((-3, 4096 * 7, "one item per workgroup (the kernel's form)"),) + tuple((-7, g, f"persistent, {g} workgroups") for g in (512, 768, 1024, 1536, 2048, 4096)):
    ms = t(lambda: lib.raptor_probe_fill(x.data_ptr(), blocks, mode, st))
    print(f"{name}: {ms:.3f} ms  {nbytes/ms/1e6:.0f} GB/s")
