"""Checks of the legged (Talos / G1) oracle path and a numpy prototype of the re-rooted formulation the HIP kernel
uses: with the stance foot planted, the robot is a FIXED-BASE chain rooted at the foot, so
  tau      = joint torques of that chain (no P_dep, no lambda solve),
  lambda   = W6^-T X_p^-T F_root   (total wrench at the foot point, RPY rows through W = d rpy / d omega at R_des).
Run: python scripts/proto_legged.py [talos|g1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raptor_amd import _capi as A
from raptor_amd import robots
from raptor_amd.urdf import jtype_transform

from proto_world_derivs import crf, crm, skew


def build(oracle, which, N=5):
    if which == "talos":
        m = robots.talos_model(os.path.join(ROOT, "tests", "robots", "talos_reduced_armfixed_floatingbase.urdf"))
        cfg = robots.talos_config(m)
    else:
        m = robots.g1_model(os.path.join(ROOT, "tests", "robots", "g1_22dof_floatingbase.urdf"))
        cfg = robots.g1_config(m)
    keep = []
    mh = oracle.model_create(m)
    t = A.make_traj_desc(keep, A.TRAJ_BEZIER, cfg.T, N, 12, cfg.degree, A.TIME_CHEBYSHEV)
    p = A.ProblemHandle.legged(oracle, mh, t, cfg)
    return m, cfg, p, (keep, mh)


def rerooted(m, cfg, q_full, v_full, a_full):
    """tau (independent joints) and lambda from a fixed-base tree rooted at the stance foot, world frame."""
    n = m.nv
    # world kinematics of the floating-base tree at the oracle's solved q (the kernel derives the torso pose
    # itself; the prototype only checks the dynamics identity)
    R = [None] * n; o = [None] * n; s = [None] * n; v = [None] * n; a = [None] * n; f = [None] * n
    a0 = np.concatenate([np.zeros(3), -m.gravity])
    for i in range(n):
        p = int(m.parent[i])
        Rp, op = (np.eye(3), np.zeros(3)) if p < 0 else (R[p], o[p])
        vp, ap = (np.zeros(6), a0) if p < 0 else (v[p], a[p])
        Aw = Rp @ m.placement_R[i]
        oi = op + Rp @ m.placement_p[i]
        jt = int(m.jtype[i]); aj = abs(jt); sg = 1.0 if jt > 0 else -1.0
        Rq, pq = jtype_transform(jt, q_full[i])
        if aj <= 3:
            ax = sg * Aw[:, aj - 1]
            si = np.concatenate([ax, np.cross(oi, ax)])
        else:
            ax = sg * Aw[:, aj - 4]
            si = np.concatenate([np.zeros(3), ax])
            oi = oi + Aw @ pq
        R[i] = Aw @ Rq; o[i] = oi; s[i] = si
        v[i] = vp + si * v_full[i]
        a[i] = ap + si * a_full[i] + crm(vp, si) * v_full[i]
        c = oi + R[i] @ m.lever[i]
        mm = m.mass[i]
        Ic = R[i] @ m.inertia[i] @ R[i].T
        hh = mm * c
        IO = Ic + mm * skew(c) @ skew(c).T
        I = np.zeros((6, 6)); I[:3, :3] = IO; I[:3, 3:] = skew(hh); I[3:, :3] = -skew(hh); I[3:, 3:] = mm * np.eye(3)
        f[i] = I @ a[i] + crf(v[i], I @ v[i])
    foot = cfg.contact_joint_id - 1
    # stance chain: foot joint up to the first leg joint
    chain = []
    j = foot
    while j > 5:
        chain.append(j)
        j = int(m.parent[j])
    Ftot = sum(f)
    # subtree wrench sums in the original tree
    F = [x.copy() for x in f]
    for i in range(n - 1, -1, -1):
        p = int(m.parent[i])
        if p >= 0:
            F[p] += F[i]
    tau = np.zeros(n)
    for i in range(6, n):
        if i in chain:
            # re-rooted: successor side of joint i = everything except subtree(i); axis -s_i
            tau[i] = -s[i] @ (Ftot - F[i])
        else:
            tau[i] = s[i] @ F[i]
    # foot point and lambda
    p_foot = o[foot] + R[foot] @ np.asarray(cfg.stance_foot_endT_p)
    Rd = np.asarray(cfg.stance_foot_T_des_R)
    W = rpy_rate_matrix(Rd)
    lam_lin = Ftot[3:]
    lam_rot = np.linalg.solve(W.T, Ftot[:3] - np.cross(p_foot, Ftot[3:]))
    # foot must be stationary
    vf = v[foot]; af = a[foot]
    foot_motion = max(np.abs(vf[:3]).max(), np.abs(vf[3:] + np.cross(vf[:3], p_foot)).max())
    return tau[cfg.independent_ids], np.concatenate([lam_lin, lam_rot]), foot_motion


def rpy_rate_matrix(R):
    """d(roll,pitch,yaw)/d(omega) for the reference's getRPY (Transform.cpp:379-390) with dR = [omega]x R."""
    W = np.zeros((3, 3))
    for k in range(3):
        e = np.zeros(3); e[k] = 1
        dR = skew(e) @ R
        rollDen = R[1, 2] ** 2 + R[2, 2] ** 2
        yawDen = R[0, 0] ** 2 + R[0, 1] ** 2
        W[0, k] = (R[1, 2] * dR[2, 2] - R[2, 2] * dR[1, 2]) / rollDen
        W[1, k] = dR[0, 2] / np.sqrt(1 - R[0, 2] ** 2)
        W[2, k] = (R[0, 1] * dR[0, 0] - R[0, 0] * dR[0, 1]) / yawDen
    return W


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "talos"
    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    N = 5
    m, cfg, p, keep = build(oracle, which, N)
    print(which, "nv", m.nv, "n", p.n, "m", p.m)
    rng = np.random.default_rng(1)
    Z = 0.3 * rng.uniform(-1, 1, (2, p.n))
    out = p.constrained_id_host(Z, N, m.nv, 12, 6, True)
    # --- re-rooted identity
    worst_t = worst_l = 0
    for b in range(2):
        for i in range(N):
            t, lam, fm = rerooted(m, cfg, out["q"][b, i], out["v"][b, i], out["a"][b, i])
            worst_t = max(worst_t, np.abs(t - out["tau"][b, i]).max() / (1 + np.abs(out["tau"][b, i]).max()))
            worst_l = max(worst_l, np.abs(lam - out["lam"][b, i]).max() / (1 + np.abs(out["lam"][b, i]).max()))
            assert fm < 1e-9, fm
    print(f"re-rooted fixed-base chain vs oracle ConstrainedInverseDynamics: tau {worst_t:.2e}  lambda {worst_l:.2e}")
    # --- finite differences (the reference's own criterion, TestInverseDynamics.cpp recipe, h = 1e-7 central)
    h = 1e-6
    z0 = Z[0]
    fd_t = np.zeros_like(out["ptau_pz"][0]); fd_l = np.zeros_like(out["plambda_pz"][0])
    for c in range(84):
        zp = z0.copy(); zp[c] += h
        zm = z0.copy(); zm[c] -= h
        op = p.constrained_id_host(zp[None], N, m.nv, 12, 6, False)
        om = p.constrained_id_host(zm[None], N, m.nv, 12, 6, False)
        fd_t[:, :, c] = (op["tau"][0] - om["tau"][0]) / (2 * h)
        fd_l[:, :, c] = (op["lam"][0] - om["lam"][0]) / (2 * h)
    et = np.abs(fd_t - out["ptau_pz"][0]).max() / (1 + np.abs(fd_t).max())
    el = np.abs(fd_l - out["plambda_pz"][0]).max() / (1 + np.abs(fd_l).max())
    print(f"finite differences: ptau_pz {et:.2e}  plambda_pz {el:.2e}  (|ptau| max {np.abs(fd_t).max():.1f})")
    G, J = p.eval_host(Z)
    print("eval_host", G.shape, J.shape, "finite", np.isfinite(G).all(), np.isfinite(J).all())


if __name__ == "__main__":
    main()
