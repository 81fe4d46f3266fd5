#!/usr/bin/env python3
"""Single-problem Digit eval_g + eval_jac_g: 200 device-resident calls, host-synchronised (run under rocprofv3 --kernel-trace --stats
to see the kernel durations of a B = 1 call beside the wall time per call)."""
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
p, _, _ = H.digit_problem(api, keep, N=14, customized=True, periodic=True)
Z = torch.tensor(0.1 * np.random.default_rng(0).uniform(-1, 1, (1, p.n)), device="cuda")
G = torch.empty(1, p.m, dtype=torch.float64, device="cuda")
J = torch.empty(1, p.m, p.n, dtype=torch.float64, device="cuda")
flags = A.EVAL_G | A.EVAL_JAC
for _ in range(10):
    p.eval_device(Z.data_ptr(), 1, flags, G.data_ptr(), J.data_ptr(), 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    p.eval_device(Z.data_ptr(), 1, flags, G.data_ptr(), J.data_ptr(), 0)
    torch.cuda.synchronize()
print(f"digit B=1 device-resident: {1e6 * (time.perf_counter() - t0) / 200:.1f} us per call")
z1 = Z.cpu().numpy()
for _ in range(10): p.eval_host(z1, flags)
t0 = time.perf_counter()
for _ in range(200): p.eval_host(z1, flags)
print(f"digit B=1 host buffers: {1e6 * (time.perf_counter() - t0) / 200:.1f} us per call")
