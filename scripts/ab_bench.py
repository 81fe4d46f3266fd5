#!/usr/bin/env python3
"""A/B timing of several builds of libraptor_hip.so in ONE process, interleaved rounds (median / min per build).
usage: python scripts/ab_bench.py lib_a.so lib_b.so ...   (Kinova config-2 workload, eval_g + eval_jac_g)"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import raptor_amd
from raptor_amd import _capi as A
import helpers as H

libs = sys.argv[1:]
model = raptor_amd.load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
B, N = 4096, 128
keep, probs = [], []
for path in libs:
    api = A.CApi(path, "raptor_")
    api.init(0)
    probs.append(H.kinova_problem(api, model, keep, N=N, boxes=H.reach_boxes()))
g = torch.Generator(device="cpu").manual_seed(1)
Z = (torch.rand(B, 7, generator=g, dtype=torch.float64) * 2 - 1).cuda()
G = torch.empty(B, probs[0].m, dtype=torch.float64, device="cuda")
J = torch.empty(B, probs[0].m, 7, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
times = [[] for _ in libs]
for rnd in range(8):
    for k, p in enumerate(probs):
        for _ in range(2):
            p.eval_device(Z.data_ptr(), B, 3, G.data_ptr(), J.data_ptr(), st)
        torch.cuda.synchronize()
        p.profile_begin(10)
        for _ in range(10):
            p.eval_device(Z.data_ptr(), B, 3, G.data_ptr(), J.data_ptr(), st)
        torch.cuda.synchronize()
        n, a, b = p.profile_end()
        times[k].append(a + b)
for path, t in zip(libs, times):
    print(f"{os.path.basename(path):28s} median {statistics.median(t):.4f} ms  min {min(t):.4f} ms")
