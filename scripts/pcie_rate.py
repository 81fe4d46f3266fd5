#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (raptor_eval_batch_host) on the headline workload: z in host memory in,
g / pg_pz in host memory out, buffers reused.  The figure DESIGN.md 4 quotes next to the HBM-resident `value` of bench.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raptor_amd
from raptor_amd import _capi as A
import helpers as H
api = A.CApi(os.path.join(ROOT, "raptor_amd", "libraptor_hip.so"), "raptor_"); api.init(0)
model = raptor_amd.load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
keep = []
p = H.kinova_problem(api, model, keep, N=128, boxes=H.reach_boxes())
for B in (64, 512, 4096):
    Z = np.random.default_rng(1).uniform(-1, 1, (B, 7))
    G = np.empty((B, p.m)); J = np.empty((B, p.m, 7))
    fn = api.fn("eval_batch_host")
    args = (p.handle, A._ptr(Z), B, A.EVAL_G | A.EVAL_JAC, A._ptr(G), A._ptr(J))
    api.check(fn(*args))
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): api.check(fn(*args))
    dt = (time.perf_counter() - t0) / reps
    by = 8 * B * (7 + p.m * 8)
    print(f"B={B:5d}: {dt*1e3:8.2f} ms per call  {B*128/dt:.3e} sample-evals/s  {by/dt/1e9:.1f} GB/s over the host interface")
