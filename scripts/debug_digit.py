import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import raptor_amd
from raptor_amd import _capi as A
np.set_printoptions(linewidth=200, precision=5, suppress=True)
hip = raptor_amd.api(); hip.init(0)
oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
keep = []
N = 3
po, m, cfg = H.digit_problem(oracle, keep, N=N)
ph, _, _ = H.digit_problem(hip, keep, N=N)
rng = np.random.default_rng(20250523 + 4)
Z = 0.1 * rng.uniform(-1, 1, (1, po.n))
ro = po.constrained_id_host(Z, N, 36, 12)
rh = ph.constrained_id_host(Z, N, 36, 12)
for key in ("q", "v", "a", "tau", "lam"):
    d = np.abs(rh[key] - ro[key])
    print(key, "max err", d.max())
    print(" hip   ", rh[key][0, 0]); print(" oracle", ro[key][0, 0])
for key in ("ptau_pz", "plambda_pz"):
    d = np.abs(rh[key] - ro[key]) / (1 + np.abs(ro[key]))
    print(key, "max rel", d.max(), "at", np.unravel_index(d.argmax(), d.shape))
print(m.names)
