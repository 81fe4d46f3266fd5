#!/usr/bin/env python3
"""A/B timing of Optimizer::eval_h (Kinova, 4096 x 128) over several builds of libraptor_hip.so in ONE process; the first
library's Hessians are the comparison for the others.  usage: python scripts/ab_hess.py lib_a.so lib_b.so ..."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import raptor_amd
from raptor_amd import _capi as A
import helpers as H

libs = sys.argv[1:]
model = raptor_amd.load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
B, N = int(os.environ.get("AB_B", 4096)), 128
keep, probs = [], []
for path in libs:
    api = A.CApi(path, "raptor_")
    api.init(0)
    p = H.kinova_problem(api, model, keep, N=N, boxes=H.reach_boxes())
    p.set_objective(np.zeros(p.n), N // 2)
    probs.append(p)
g = torch.Generator(device="cpu").manual_seed(1)
Z = (torch.rand(B, 7, generator=g, dtype=torch.float64) * 2 - 1).cuda()
lam = (torch.rand(B, probs[0].m, generator=g, dtype=torch.float64) * 2 - 1).cuda()
of = torch.ones(B, dtype=torch.float64, device="cuda")
Hs = [torch.empty(B, 28, dtype=torch.float64, device="cuda") for _ in libs]
st = torch.cuda.current_stream().cuda_stream
times = [[] for _ in libs]
for rnd in range(5):
    for k, p in enumerate(probs):
        p.eval_h_device(Z.data_ptr(), B, of.data_ptr(), lam.data_ptr(), Hs[k].data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            p.eval_h_device(Z.data_ptr(), B, of.data_ptr(), lam.data_ptr(), Hs[k].data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 5)
for k, (path, t) in enumerate(zip(libs, times)):
    err = H.rel_err(Hs[k].cpu().numpy(), Hs[0].cpu().numpy())
    print(f"{os.path.basename(path):28s} median {statistics.median(t):.4f} ms  min {min(t):.4f} ms  rel diff vs first {err:.2e}")
