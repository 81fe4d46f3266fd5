"""Checks the reduced composite form used by the kernels:
   2B^C x = Q x + x x* H,  (2B^C)^T x = Q x - x x* H,  Q = [[Q_TL, [t]x], [-[t]x, 0]],  t = lin(H^C),
   Q_TL = sum_l W_l + W_l^T - (h u^T + u h^T) + 2 (u.h) 1,  W_l = [w]x I_O   (per link, world frame)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from proto_world_derivs import skew, crf_mat, crf

rng = np.random.default_rng(1)
for _ in range(5):
    Qfull = np.zeros((6, 6)); QTL = np.zeros((3, 3)); H = np.zeros(6)
    for l in range(4):
        m = rng.uniform(0.5, 2); c = rng.uniform(-1, 1, 3); A = rng.uniform(-1, 1, (3, 3)); Ic = A @ A.T
        v = rng.uniform(-1, 1, 6); w, u = v[:3], v[3:]
        hv = m * c; IO = Ic + m * skew(c) @ skew(c).T
        I = np.zeros((6, 6)); I[:3, :3] = IO; I[:3, 3:] = skew(hv); I[3:, :3] = -skew(hv); I[3:, 3:] = m * np.eye(3)
        P = crf_mat(v) @ I
        Qfull += P + P.T
        W = skew(w) @ IO
        QTL += W + W.T - (np.outer(hv, u) + np.outer(u, hv)) + 2 * (u @ hv) * np.eye(3)
        H += I @ v
    t = H[3:]
    Qred = np.zeros((6, 6)); Qred[:3, :3] = QTL; Qred[:3, 3:] = skew(t); Qred[3:, :3] = -skew(t)
    print(np.abs(Qfull - Qred).max())
