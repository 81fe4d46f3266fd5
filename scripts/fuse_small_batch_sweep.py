#!/usr/bin/env python3
"""Fused core kernel against the two-kernel form at small batches: host-synchronised time per eval_g + eval_jac_g call, device-resident
buffers, B = 1 .. 1024 (RAPTOR_NO_FUSE is read at every call, so both forms run in one process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import raptor_amd
from raptor_amd import _capi as A
import helpers as H
api = raptor_amd.api(); api.init(0)
keep = []
which = sys.argv[1] if len(sys.argv) > 1 else "digit"
if which == "digit":
    p = H.digit_problem(api, keep, N=14, customized=True, periodic=True)[0]
else:
    p = H.legged_problem(api, which, keep, N=14, customized=True, periodic=True)[0]
flags = A.EVAL_G | A.EVAL_JAC
for B in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
    Z = torch.tensor(0.1 * np.random.default_rng(0).uniform(-1, 1, (B, p.n)), device="cuda")
    G = torch.empty(B, p.m, dtype=torch.float64, device="cuda")
    J = torch.empty(B, p.m, p.n, dtype=torch.float64, device="cuda")
    out = []
    for form in ("fused", "two"):
        if form == "two":
            os.environ["RAPTOR_NO_FUSE"] = "1"
        else:
            os.environ.pop("RAPTOR_NO_FUSE", None)
        for _ in range(10):
            p.eval_device(Z.data_ptr(), B, flags, G.data_ptr(), J.data_ptr(), 0)
        torch.cuda.synchronize()
        reps = 200 if B <= 64 else 60
        t0 = time.perf_counter()
        for _ in range(reps):
            p.eval_device(Z.data_ptr(), B, flags, G.data_ptr(), J.data_ptr(), 0)
            torch.cuda.synchronize()
        out.append(1e6 * (time.perf_counter() - t0) / reps)
    os.environ.pop("RAPTOR_NO_FUSE", None)
    print(f"{which} B={B:5d}: fused {out[0]:8.1f} us   two-kernel {out[1]:8.1f} us")
