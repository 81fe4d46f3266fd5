#!/usr/bin/env python3
"""Single-problem eval_g + eval_jac_g: direct launches against the same launches captured once in a HIP graph and replayed
(torch.cuda.CUDAGraph around raptor_eval_batch on the capturing stream).  usage: python scripts/graph_latency_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import raptor_amd
from raptor_amd import _capi as A
import helpers as H

hip = raptor_amd.api()
hip.init(0)
keep = []
probs = {"kinova": H.kinova_problem(hip, raptor_amd.load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf")), keep, N=128, boxes=H.reach_boxes())}
import bench
for name in ("g1", "digit"):
    probs[name] = bench.WORKLOADS[name]().build(hip, keep)
flags = A.EVAL_G | A.EVAL_JAC
for name, p in probs.items():
    rng = np.random.default_rng(1)
    Z = torch.from_numpy(0.1 * rng.uniform(-1, 1, (1, p.n))).cuda()
    G = torch.empty(1, p.m, dtype=torch.float64, device="cuda")
    J = torch.empty(1, p.m, p.n, dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            p.eval_device(Z.data_ptr(), 1, flags, G.data_ptr(), J.data_ptr(), s.cuda_stream)
    s.synchronize()
    Gd, Jd = G.clone(), J.clone()
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        p.eval_device(Z.data_ptr(), 1, flags, G.data_ptr(), J.data_ptr(), s.cuda_stream)
        s.synchronize()
    direct = 1e6 * (time.perf_counter() - t0) / reps
    g = torch.cuda.CUDAGraph()
    G.zero_(); J.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        p.eval_device(Z.data_ptr(), 1, flags, G.data_ptr(), J.data_ptr(), torch.cuda.current_stream().cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(G, Gd) and torch.equal(J, Jd)
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
        torch.cuda.synchronize()
    graph = 1e6 * (time.perf_counter() - t0) / reps
    # a host call on the same problem afterwards must still work
    Gh, Jh = p.eval_host(Z.cpu().numpy())
    ok_host = np.array_equal(Gh, Gd.cpu().numpy())
    print(f"{name:7s} direct {direct:7.1f} us   graph replay {graph:7.1f} us   identical {same}   host call afterwards ok {ok_host}", flush=True)
