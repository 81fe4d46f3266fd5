#!/usr/bin/env python3
"""Can the HBM-bound expansion share the GPU with the VALU-bound record kernels by CU partition?  Probe: write bandwidth of a
fill kernel on a stream restricted to a CU mask (hipExtStreamCreateWithCUMask), alone and beside an fp64-FMA kernel on the complement.
usage: python scripts/cu_mask_probe.py"""
import ctypes
import time

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise SystemExit(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(s.value)


torch.cuda.init()
x = torch.empty(1 << 28, dtype=torch.float64, device="cuda")  # 2 GiB
a = torch.rand(1 << 22, dtype=torch.float64, device="cuda")


def bw(stream, reps=10):
    with torch.cuda.stream(stream):
        x.fill_(1.0)
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            x.fill_(1.0)
        e1.record(stream)
        stream.synchronize()
    return reps * x.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e12


def fma_work(stream, iters):
    with torch.cuda.stream(stream):
        y = a
        for _ in range(iters):
            y = torch.addcmul(y, y, y, value=1e-9)
    return y


print("all CUs, default stream: %.2f TB/s" % bw(torch.cuda.current_stream()))
for name, bits in (("first 128", set(range(128))), ("first 64", set(range(64))), ("every 2nd", set(range(0, 256, 2))),
                   ("every 4th", set(range(0, 256, 4))), ("first 96", set(range(96))), ("every 3rd", set(range(0, 256, 3)))):
    s = masked_stream(bits)
    print(f"{name:10s} ({len(bits)} CUs): {bw(s):.2f} TB/s", flush=True)
