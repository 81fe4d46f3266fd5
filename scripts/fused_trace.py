"""Where a wavefront of the fused legged core kernel spends its life (library built with -DLEG_G_TRACE: scripts/build_variant.sh
build/r3/lib_trace.so WORK -DLEG_G_TRACE; run with RAPTOR_HIP_LIB pointing at it).  The build leaves 100 MHz timestamp differences in the
torque values of every wavefront's first sample: phases 0-F2 | set-up of phase G (tables, weights) | torque rows | joint-limit rows |
contact rows | customized rows | waiting for the stores | whole life."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
import raptor_amd  # noqa: E402
from raptor_amd import _capi as A  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "digit"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
hip = raptor_amd.api()
hip.init(0)
keep = []
if which == "digit":
    p = H.digit_problem(hip, keep, N=14, customized=True, periodic=True)[0]
    scale, epw = 0.1, 2
else:
    p = H.legged_problem(hip, which, keep, N=14, customized=True, periodic=True)[0]
    scale, epw = 0.1, 4
g = torch.Generator(device="cpu").manual_seed(1)
Z = (scale * (torch.rand(B, p.n, generator=g, dtype=torch.float64) * 2 - 1)).cuda()
G = torch.empty(B, p.m, dtype=torch.float64, device="cuda")
J = torch.empty(B, p.m, p.n, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    p.eval_device(Z.data_ptr(), B, A.EVAL_G | A.EVAL_JAC, G.data_ptr(), J.data_ptr(), st)
torch.cuda.synchronize()
Gh = G.cpu().numpy()
ni = 12
rows = []
for e in range(0, B * 14, epw):  # first sample-eval of every wavefront
    b, i = divmod(e, 14)
    rows.append(Gh[b, i * ni:i * ni + 8])
r = np.array(rows) * 0.01  # us
names = ["phases 0-F2", "G set-up", "torque rows", "joint-limit rows", "contact rows", "customized rows", "store drain", "whole life"]
for k, nm in enumerate(names):
    print(f"{nm:18s} mean {r[:, k].mean():7.2f} us   median {np.median(r[:, k]):7.2f}   p90 {np.percentile(r[:, k], 90):7.2f}")
