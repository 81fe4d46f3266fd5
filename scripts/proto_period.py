"""numpy model of the reset-map (periodicity) kernel, checked against the oracle's literal restatement of
DigitSingleStepPeriodicityConstraints.cpp:57-215 / TalosSingleStepPeriodicityConstraints.cpp.

The reference forms H = crba(q-), J and the tensors Jx_partial_dq, JTx_partial_dq after the stance swap, the RNEA
derivative with gravity off, and multiplies by pq_pz(last), pv_pz(last).  The kernel instead differentiates
    g1 = H(q)(v+ - v-(q)) - J(q)^T lambda,    g2 = J(q) v+
along the twelve tangent directions of the stance-phase manifold (q moves by Pfull[:, t] per unit actuator t) in
world-frame spatial algebra: with w_l = body twist per unit q_t,
    D_t s_k = w_k x s_k,   D_t (I_l a) = w_l x* (I_l a) - I_l (w_l x a) + I_l D_t a,
    D_t p_c = w_foot at p_c,   D_t E(R_c) from the closed form of rpy_rates,
and the dependent-velocity derivative Vq = d v-/d q_t never needs J_dep^-1: for the base joints
    sum_b s_b Vq_b = -(sum_{path(old stance foot)} (w_j x s_j) v-_j + s_j Vq_j)     (the planted foot stays planted).
Run: python scripts/proto_period.py [talos|g1|digit]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from raptor_amd import _capi as A
from raptor_amd.urdf import jtype_transform
from proto_world_derivs import crf, crm, skew
from proto_legged import rpy_rate_matrix


def kinematics(m, q):
    n = m.nv
    R = [None] * n; o = [None] * n; s = [None] * n; I6 = [None] * n
    for i in range(n):
        p = int(m.parent[i])
        Rp, op = (np.eye(3), np.zeros(3)) if p < 0 else (R[p], o[p])
        Aw = Rp @ m.placement_R[i]
        oi = op + Rp @ m.placement_p[i]
        jt = int(m.jtype[i]); aj = abs(jt); sg = 1.0 if jt > 0 else -1.0
        Rq, pq = jtype_transform(jt, q[i])
        if aj <= 3:
            ax = sg * Aw[:, aj - 1]
            si = np.concatenate([ax, np.cross(oi, ax)])
        else:
            ax = sg * Aw[:, aj - 4]
            si = np.concatenate([np.zeros(3), ax])
            oi = oi + Aw @ pq
        R[i] = Aw @ Rq; o[i] = oi; s[i] = si
        c = oi + R[i] @ m.lever[i]
        mass = m.mass[i]
        Ic = R[i] @ m.inertia[i] @ R[i].T
        hh = mass * c
        I = np.zeros((6, 6)); I[:3, :3] = Ic + mass * skew(c) @ skew(c).T; I[:3, 3:] = skew(hh); I[3:, :3] = -skew(hh); I[3:, 3:] = mass * np.eye(3)
        I6[i] = I
    return R, o, s, I6


def rpy_rates(R, w):
    dR = skew(w) @ R
    return np.array([(R[1, 2] * dR[2, 2] - R[2, 2] * dR[1, 2]) / (R[1, 2] ** 2 + R[2, 2] ** 2),
                     dR[0, 2] / np.sqrt(1 - R[0, 2] ** 2),
                     (R[0, 1] * dR[0, 0] - R[0, 0] * dR[0, 1]) / (R[0, 0] ** 2 + R[0, 1] ** 2)])


def d_rpy_rates(R, w, ww):
    """directional derivative of rpy_rates(R, w) with respect to R along dR = [ww]x R (w fixed)."""
    dR = skew(w) @ R
    vR = skew(ww) @ R          # variation of R
    vdR = skew(w) @ vR         # variation of dR
    D0 = R[1, 2] ** 2 + R[2, 2] ** 2
    N0 = R[1, 2] * dR[2, 2] - R[2, 2] * dR[1, 2]
    vN0 = vR[1, 2] * dR[2, 2] + R[1, 2] * vdR[2, 2] - vR[2, 2] * dR[1, 2] - R[2, 2] * vdR[1, 2]
    vD0 = 2 * (R[1, 2] * vR[1, 2] + R[2, 2] * vR[2, 2])
    D2 = R[0, 0] ** 2 + R[0, 1] ** 2
    N2 = R[0, 1] * dR[0, 0] - R[0, 0] * dR[0, 1]
    vN2 = vR[0, 1] * dR[0, 0] + R[0, 1] * vdR[0, 0] - vR[0, 0] * dR[0, 1] - R[0, 0] * vdR[0, 1]
    vD2 = 2 * (R[0, 0] * vR[0, 0] + R[0, 1] * vR[0, 1])
    x = R[0, 2]
    da = 1 / np.sqrt(1 - x * x)
    dda = x / (1 - x * x) ** 1.5
    return np.array([(vN0 * D0 - N0 * vD0) / D0 ** 2, dda * vR[0, 2] * dR[0, 2] + da * vdR[0, 2], (vN2 * D2 - N2 * vD2) / D2 ** 2])


MATRIX = False


def matrix_form(m, n, ni, nd, s, I6, anc, sub, F, W, aD, Vp, dv, vplus, vminus, Pfull, VqLoop, nf, of, pc, Rc, E, lt, lr, H):
    """D1 = A Pfull - H Vq,  D2 = -H Pfull,  G2_foot = B Pfull with per-joint vectors (what the HIP kernel evaluates):
         psi_k = a_parent(k) x s_k,  IS_k = I^C_k s_k,  X_k = s_k x* W - D_k W  (k on the new foot's path),
         cq_k = s_k x* F_k + I^C_k psi_k + X_k,
         A[j,k] = IS_j . psi_k + [foot in sub(j)] s_j . X_k   (k on the path of j),   A[j,k] = s_j . cq_k   (j strict ancestor of k)."""
    IC = [sum(I6[l] for l in sub[k]) for k in range(n)]
    on_nf = [nf in sub[k] for k in range(n)]
    T = np.array([[d_rpy_rates(Rc, ei, ej) for ej in np.eye(3)] for ei in np.eye(3)])  # T[i][j] = vector
    A9 = np.array([[lr @ T[i][j] for j in range(3)] for i in range(3)])
    Bm = np.array([[sum(T[i][j][r] * Vp[nf][i] for i in range(3)) for j in range(3)] for r in range(3)])
    psi = []; X = []; cq = []; IS = []; Bk = []
    for k in range(n):
        par = int(m.parent[k])
        apk = aD[par] if par >= 0 else np.zeros(6)
        psi.append(crm(apk, s[k]))
        IS.append(IC[k] @ s[k])
        if on_nf[k]:
            spc = s[k][3:] + np.cross(s[k][:3], pc)
            DkW = np.concatenate([np.cross(spc, lt) + A9 @ s[k][:3], np.zeros(3)])
            X.append(crf(s[k], W) - DkW)
            Vpk = Vp[par] if par >= 0 else np.zeros(6)
            Cp = crm(s[k], Vp[nf] - Vpk)
            Bk.append(np.concatenate([Cp[3:] + np.cross(Cp[:3], pc) + np.cross(Vp[nf][:3], spc), Bm @ s[k][:3] + E @ Cp[:3]]))
        else:
            X.append(np.zeros(6)); Bk.append(np.zeros(6))
        # F here already has -W folded in: F_k = sum f - [nf]W, so s x* (F_k + W) + ... - D_k W = s x* F_k + X_k
        cq.append(crf(s[k], F[k]) + IC[k] @ psi[k] + X[k])
    A = np.zeros((n, n))
    for j in range(n):
        for k in anc[j]:
            A[j, k] = IS[j] @ psi[k] + (s[j] @ X[k] if on_nf[j] else 0.0)
            if k != j:
                A[k, j] = s[k] @ cq[j]
    # d v-/d q_t: loop joints from Y2 (given), base joints from the planted old stance foot
    Vq = VqLoop.copy()
    base = anc[of][:6]
    Sb = np.column_stack([s[b] for b in base])
    for t in range(ni):
        u = Pfull[:, t]
        w = np.zeros(6); r = np.zeros(6)
        for j in anc[of]:
            w = w + s[j] * u[j]
            r += crm(w, s[j]) * vminus[j] + s[j] * Vq[j, t]
        Vq[base, t] = np.linalg.solve(Sb, -r)
    D1 = A @ Pfull - H @ Vq
    D2 = -H @ Pfull
    G2 = np.zeros((nd, ni))
    G2[nd - 6:] = np.column_stack(Bk) @ Pfull
    return D1, D2, G2


def period_block(m, ind, dep, q, vminus, Pfull, Vq, vplus, lam, old_foot, new_foot, endR, endp, id1, id2, sign, q0, v0,
                 w0, wN, cp, loops=None):
    """w0 / wN: Bernstein weights (B, dB) of the first / last sample, shape (2, K).  Returns g (m_p), J (m_p, n)."""
    n = m.nv; ni = len(ind); nd = len(dep); K = w0.shape[1]
    V = ni * K + n + nd
    R, o, s, I6 = kinematics(m, q)
    anc = []
    for l in range(n):
        pth = []; j = l
        while j >= 0:
            pth.append(j); j = int(m.parent[j])
        anc.append(pth[::-1])
    sub = [[l for l in range(n) if k in anc[l]] for k in range(n)]
    dv = vplus - vminus
    field = lambda x: [sum(s[j] * x[j] for j in anc[l]) for l in range(n)]
    aD = field(dv); Vp = field(vplus)
    f_ = new_foot
    Rc = R[f_] @ endR; pc = o[f_] + R[f_] @ endp
    E = rpy_rate_matrix(Rc)
    lt, lr = lam[nd - 6:nd - 3], lam[nd - 3:]
    W = np.concatenate([np.cross(pc, lt) + E.T @ lr, lt])
    fb = [I6[l] @ aD[l] for l in range(n)]
    fb[f_] = fb[f_] - W
    F = [sum(fb[l] for l in sub[k]) for k in range(n)]
    mp = n + nd + 2 * ni + 7
    g = np.zeros(mp); J = np.zeros((mp, V))
    cv, cl = V - n - nd, V - nd  # first column of v+, of lambda
    g[:n] = [s[k] @ F[k] + m.armature[k] * dv[k] for k in range(n)]
    # H and the foot rows of J
    for k in range(n):
        for c in range(n):
            common = [l for l in sub[k] if l in sub[c]]
            J[k, cv + c] = s[k] @ sum((I6[l] @ s[c] for l in common), np.zeros(6)) + (m.armature[k] if k == c else 0.0)
    Jf = np.zeros((6, n))
    for k in anc[f_]:
        Jf[:3, k] = s[k][3:] + np.cross(s[k][:3], pc)
        Jf[3:, k] = E @ s[k][:3]
    r0 = nd - 6
    g[n + r0:n + nd] = np.concatenate([Vp[f_][3:] + np.cross(Vp[f_][:3], pc), E @ Vp[f_][:3]])
    J[n + r0:n + nd, cv:cv + n] = Jf
    J[:n, cl + r0:cl + nd] = -Jf.T
    D1 = np.zeros((n, ni)); D2 = np.zeros((n, ni)); G2 = np.zeros((nd, ni))
    if MATRIX:
        D1, D2, G2 = matrix_form(m, n, ni, nd, s, I6, anc, sub, F, W, aD, Vp, dv, vplus, vminus, Pfull, Vq, f_, old_foot, pc, Rc, E,
                                 lt, lr, J[:n, cv:cv + n])
    for t in range(ni if not MATRIX else 0):
        u = Pfull[:, t]; vq = Vq[:, t]
        w = field(u)
        CD = []; Cp = []; Cm = []; Sq = []
        for l in range(n):
            cd = np.zeros(6); cpp = np.zeros(6); cm = np.zeros(6); sq = np.zeros(6)
            for j in anc[l]:
                c = crm(w[j], s[j])
                cd += c * dv[j]; cpp += c * vplus[j]; cm += c * vminus[j]; sq += s[j] * vq[j]
            CD.append(cd); Cp.append(cpp); Cm.append(cm); Sq.append(sq)
        r_of = Cm[old_foot] + Sq[old_foot]
        df = []
        for l in range(n):
            da = CD[l] - Sq[l] + r_of
            Ia = I6[l] @ aD[l]
            df.append(crf(w[l], Ia) - I6[l] @ crm(w[l], aD[l]) + I6[l] @ da)
        dpc = w[f_][3:] + np.cross(w[f_][:3], pc)
        dEtl = np.array([lr @ d_rpy_rates(Rc, e, w[f_][:3]) for e in np.eye(3)])
        df[f_] = df[f_] - np.concatenate([np.cross(dpc, lt) + dEtl, np.zeros(3)])
        for k in range(n):
            dF = sum(df[l] for l in sub[k])
            FP = sum(I6[l] @ w[l] for l in sub[k])
            D1[k, t] = crm(w[k], s[k]) @ F[k] + s[k] @ dF - m.armature[k] * vq[k]
            D2[k, t] = -(s[k] @ FP + m.armature[k] * u[k])
        dVp = Cp[f_]
        G2[r0:r0 + 3, t] = dVp[3:] + np.cross(dVp[:3], pc) + np.cross(Vp[f_][:3], dpc)
        G2[r0 + 3:, t] = d_rpy_rates(Rc, Vp[f_][:3], w[f_][:3]) + E @ dVp[:3]
    # closed loops (Digit): rows / columns of the loop joints only, in each loop's own frame
    if loops:
        for (lp, row0, Y1, cols) in loops:
            c, geo = lp.c_and_geo(q)
            joints = lp.joints
            nr = len(c)
            Jl = np.column_stack([lp.D(geo, [{j: 1.0}]) for j in joints])
            lam_l = lam[row0:row0 + nr]
            for a_, j in enumerate(joints):
                g[j] -= Jl[:, a_] @ lam_l
                J[j, cl + row0:cl + row0 + nr] = -Jl[:, a_]
                J[n + row0:n + row0 + nr, cv + j] = Jl[:, a_]
            vp_d = {j: vplus[j] for j in joints}
            g[n + row0:n + row0 + nr] = lp.D(geo, [vp_d])
            for x, t in enumerate(cols):
                U = {lp.drv[x]: 1.0}
                U.update({j: Y1[r, x] for r, j in enumerate(lp.dep)})
                G2[row0:row0 + nr, t] = lp.D(geo, [vp_d, U])
                for j in joints:
                    D1[j, t] -= lam_l @ lp.D(geo, [{j: 1.0}, U])
    # trajectory columns
    for t in range(ni):
        for k in range(K):
            J[:n, t * K + k] = D1[:, t] * wN[0, k] + D2[:, t] * wN[1, k]
            J[n:n + nd, t * K + k] = G2[:, t] * wN[0, k]
    o3, o4, o5 = n + nd, n + nd + ni, n + nd + 2 * ni
    ti = {j: t for t, j in enumerate(ind)}
    for i in range(ni):
        g[o3 + i] = q0[id1[i]] + sign * q[id2[i]]
        g[o4 + i] = v0[id1[i]] + sign * vplus[id2[i]]
        J[o3 + i, ti[id1[i]] * K:(ti[id1[i]] + 1) * K] += w0[0]
        J[o3 + i, ti[id2[i]] * K:(ti[id2[i]] + 1) * K] += sign * wN[0]
        J[o4 + i, ti[id1[i]] * K:(ti[id1[i]] + 1) * K] += w0[1]
        J[o4 + i, cv + id2[i]] += sign
    mu, gamma, Lx, Ly = cp
    c6 = lam[nd - 6:]
    fz = c6[2]
    g[o5:] = [fz, c6[0] ** 2 + c6[1] ** 2 - (mu * fz) ** 2, c6[5] ** 2 - (gamma * fz) ** 2, c6[3] - Lx * fz, -Lx * fz - c6[3],
              c6[4] - Ly * fz, -Ly * fz - c6[4]]
    cc = cl + nd - 6
    J[o5, cc + 2] = 1
    J[o5 + 1, cc:cc + 3] = [2 * c6[0], 2 * c6[1], -2 * mu ** 2 * fz]
    J[o5 + 2, cc + 5] = 2 * c6[5]; J[o5 + 2, cc + 2] = -2 * gamma ** 2 * fz
    J[o5 + 3, cc + 3] = 1; J[o5 + 3, cc + 2] = -Lx
    J[o5 + 4, cc + 3] = -1; J[o5 + 4, cc + 2] = -Lx
    J[o5 + 5, cc + 4] = 1; J[o5 + 5, cc + 2] = -Ly
    J[o5 + 6, cc + 4] = -1; J[o5 + 6, cc + 2] = -Ly
    return g, J


def run(which="talos", N=5, stance="L", verbose=False):
    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    keep = []
    if which == "digit":
        p, m, cfg = H.digit_problem(oracle, keep, N=N, stance=stance, periodic=True)
    else:
        p, m, cfg = H.legged_problem(oracle, which, keep, N=N, stance=stance, periodic=True)
    n = m.nv
    ind, dep = list(cfg.independent_ids), list(cfg.dependent_ids)
    ni, nd = len(ind), len(dep)
    K = cfg.degree + 1
    rng = np.random.default_rng(11)
    Z = 0.1 * rng.uniform(-1, 1, (1, p.n))
    ref = p.constrained_id_host(Z, N, n, ni, derivatives=False)
    Gref, Jref = p.eval_host(Z)
    mp = n + nd + 2 * ni + 7
    gref, jref = Gref[0, -mp:], Jref[0, -mp:]
    basis = np.zeros((N, 2, K))
    for k in range(K):
        zz = np.zeros((1, p.n)); zz[0, k] = 1.0
        qq, qd, _ = p.trajectory_host(zz, N, ni)
        basis[:, 0, k] = qq[0, :, 0]; basis[:, 1, k] = qd[0, :, 0]
    q = ref["q"][0, N - 1]; vminus = ref["v"][0, N - 1]
    # Pfull from the joint-limit rows of the last sample (column K-1 of each actuator has B = 1 at t = T)
    off_q = N * ni
    rows = Jref[0, off_q + (N - 1) * n: off_q + N * n]
    Pfull = np.column_stack([rows[:, t * K + K - 1] for t in range(ni)])
    vplus = Z[0, p.n - n - nd: p.n - nd]; lam = Z[0, p.n - nd:]
    sleg = 0 if stance == "L" else 1
    Vq = np.zeros((n, ni))
    loops = None
    if which == "digit":
        import proto_digit as PD
        old_foot = cfg.leg_ids[sleg][PD.R_["toe_roll"]]; new_foot = cfg.leg_ids[1 - sleg][PD.R_["toe_roll"]]
        s3 = np.sin(np.pi / 3)
        endR = np.array([[0, 1, 0], [-0.5, 0, s3], [s3, 0, 0.5]]) if sleg == 1 else np.array([[0, -1, 0], [0.5, 0, -s3], [s3, 0, 0.5]])
        endp = np.array([0, -0.05456, -0.0315]) if sleg == 1 else np.array([0, 0.05456, -0.0315])
        id1 = ind; id2 = ind[6:] + ind[:6]; sign = 1.0
        loops = []
        ti = {j: t for t, j in enumerate(ind)}
        for leg in range(2):
            knee, toe = PD.loops_of_leg(cfg, leg)
            for name, l in (("knee", knee), ("toe", toe)):
                lp = PD.Loop(m, *l)
                th = q.copy()
                Y1, Y2, Y3, _ = lp.solve(th)
                row0 = {("toe", 1): 0, ("toe", 0): 6, ("knee", 1): 12, ("knee", 0): 15}[(name, leg)]
                cols = [ti[j] for j in lp.drv]
                xd = np.array([vminus[j] for j in lp.drv])
                for r, j in enumerate(lp.dep):
                    for x, t in enumerate(cols):
                        Vq[j, t] = Y2[r, x] @ xd
                loops.append((lp, row0, Y1, cols))
    else:
        from raptor_amd import robots
        per = (robots.talos_period if which == "talos" else robots.g1_period)(m, stance)
        old_foot = cfg.contact_joint_id - 1; new_foot = per["swing_contact_joint_id"] - 1
        endR = np.eye(3); endp = np.array(per["swing_foot_endT_p"])
        id1, id2, sign = per["joint_id1"], per["joint_id2"], per["mirror_sign"]
    g, J = period_block(m, ind, dep, q, vminus, Pfull, Vq, vplus, lam, old_foot, new_foot, endR, endp, id1, id2, sign,
                        ref["q"][0, 0], ref["v"][0, 0], basis[0], basis[N - 1], (cfg.mu, cfg.gamma, cfg.Lx, cfg.Ly), loops)
    eg = np.abs(g - gref) / (1 + np.abs(gref)); ej = np.abs(J - jref) / (1 + np.abs(jref))
    if verbose:
        blocks = [("g1", 0, n), ("g2", n, n + nd), ("g3", n + nd, n + nd + ni), ("g4", n + nd + ni, n + nd + 2 * ni), ("g5", n + nd + 2 * ni, mp)]
        for nm, a, b in blocks:
            print(f"  {nm}: g {eg[a:b].max():.2e}  J traj {ej[a:b, :ni * K].max():.2e}  J v+ {ej[a:b, ni * K:ni * K + n].max():.2e}  J lam {ej[a:b, ni * K + n:].max():.2e}")
    return float(eg.max()), float(ej.max())


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "talos"
    if len(sys.argv) > 2 and sys.argv[2] == "matrix":
        MATRIX = True
    print(which, run(which, verbose=True))
