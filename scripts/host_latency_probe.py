#!/usr/bin/env python3
"""Where the time of a single-problem raptor_eval_batch_host call goes (Kinova): flags and sample counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raptor_amd
from raptor_amd import _capi as A
import helpers as H
api = raptor_amd.api(); api.init(0)
model = raptor_amd.load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
for N in (16, 128):
    keep = []
    p = H.kinova_problem(api, model, keep, N=N, boxes=H.reach_boxes())
    z = np.random.default_rng(0).uniform(-1, 1, (1, 7))
    for flags, name in ((A.EVAL_G, "G"), (A.EVAL_JAC, "JAC"), (A.EVAL_G | A.EVAL_JAC, "G|JAC")):
        for _ in range(10): p.eval_host(z, flags)
        t0 = time.perf_counter()
        for _ in range(200): p.eval_host(z, flags)
        print(f"N={N} {name}: {1e6*(time.perf_counter()-t0)/200:.1f} us")
