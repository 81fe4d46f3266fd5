"""numpy prototype of the world-frame analytic RNEA derivatives used by the HIP kernels.
Checks dtau/dq, dtau/dv, M against the oracle's forward-mode restatement on a chain (Kinova) and trees."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raptor_amd import _capi as A
from raptor_amd.urdf import load_urdf, jtype_transform


def crm(v, m):
    return np.concatenate([np.cross(v[:3], m[:3]), np.cross(v[:3], m[3:]) + np.cross(v[3:], m[:3])])


def crf(v, f):
    return np.concatenate([np.cross(v[:3], f[:3]) + np.cross(v[3:], f[3:]), np.cross(v[:3], f[3:])])


def skew(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0.0]])


def crf_mat(v):
    M = np.zeros((6, 6))
    M[:3, :3] = skew(v[:3]); M[:3, 3:] = skew(v[3:]); M[3:, 3:] = skew(v[:3])
    return M


def world_derivs(model, q, qd, qdd):
    n = model.nv
    R = [None] * n; o = [None] * n; s = [None] * n; v = [None] * n; a = [None] * n
    sd = [None] * n; psidd = [None] * n; I6 = [None] * n; f = [None] * n; P = [None] * n; h = [None] * n
    a0 = np.concatenate([np.zeros(3), -model.gravity])
    for i in range(n):
        p = int(model.parent[i])
        Rp, op = (np.eye(3), np.zeros(3)) if p < 0 else (R[p], o[p])
        vp, ap = (np.zeros(6), a0) if p < 0 else (v[p], a[p])
        Aw = Rp @ model.placement_R[i]
        oi = op + Rp @ model.placement_p[i]
        jt = int(model.jtype[i]); aj = abs(jt); sg = 1.0 if jt > 0 else -1.0
        Rq, pq = jtype_transform(jt, q[i])
        if aj <= 3:
            ax = sg * Aw[:, aj - 1]
            si = np.concatenate([ax, np.cross(oi, ax)])
        else:
            ax = sg * Aw[:, aj - 4]
            si = np.concatenate([np.zeros(3), ax])
            oi = oi + Aw @ pq
        R[i] = Aw @ Rq; o[i] = oi; s[i] = si
        v[i] = vp + si * qd[i]
        sd[i] = crm(vp, si)  # = v_i x s_i
        a[i] = ap + si * qdd[i] + sd[i] * qd[i]
        psidd[i] = crm(a[i], si) + crm(v[i], sd[i])
        c = oi + R[i] @ model.lever[i]
        m = model.mass[i]
        Ic = R[i] @ model.inertia[i] @ R[i].T
        hh = m * c
        IO = Ic + m * skew(c) @ skew(c).T
        I = np.zeros((6, 6)); I[:3, :3] = IO; I[:3, 3:] = skew(hh); I[3:, :3] = -skew(hh); I[3:, 3:] = m * np.eye(3)
        I6[i] = I
        h[i] = I @ v[i]
        f[i] = I @ a[i] + crf(v[i], h[i])
        P[i] = crf_mat(v[i]) @ I
    IC = [x.copy() for x in I6]; PC = [x.copy() for x in P]; HC = [x.copy() for x in h]; F = [x.copy() for x in f]
    for i in range(n - 1, -1, -1):
        p = int(model.parent[i])
        if p >= 0:
            IC[p] += IC[i]; PC[p] += PC[i]; HC[p] += HC[i]; F[p] += F[i]
    tau = np.array([s[i] @ F[i] for i in range(n)]) + model.armature * qdd
    dq = np.zeros((n, n)); dv = np.zeros((n, n)); M = np.zeros((n, n))

    def B2(k, x):  # 2 B^C_k x = P x + P^T x + x x* H
        return PC[k] @ x + PC[k].T @ x + crf(x, HC[k])

    def B2T(k, x):  # (2B)^T x ; (x x* H)-matrix is antisymmetric?  (x -> x x* H) = -crf-bar; compute via explicit matrix
        Hm = np.zeros((6, 6))
        for t in range(6):
            e = np.zeros(6); e[t] = 1
            Hm[:, t] = crf(e, HC[k])
        return PC[k].T @ x + PC[k] @ x + Hm.T @ x

    for k in range(n):
        IS = IC[k] @ s[k]; IdS = IC[k] @ sd[k]; Ipsi = IC[k] @ psidd[k]
        BS = B2(k, s[k]); BdS = B2(k, sd[k]); cF = crf(s[k], F[k])
        c_a = IS; c_v = 2 * IdS + BS; c_q = Ipsi + BdS + cF
        # ancestors (strict) of k
        j = int(model.parent[k])
        while j >= 0:
            M[j, k] = s[j] @ c_a; dv[j, k] = s[j] @ c_v; dq[j, k] = s[j] @ c_q
            j = int(model.parent[j])
        # j = k and descendants handled from j's side: entry (j, k) for k ancestor-or-equal of j
    for j in range(n):
        IS = IC[j] @ s[j]; BtS = B2T(j, s[j])
        k = j
        while k >= 0:
            M[j, k] = IS @ s[k]
            dv[j, k] = 2 * IS @ sd[k] + BtS @ s[k]
            dq[j, k] = IS @ psidd[k] + BtS @ sd[k]
            k = int(model.parent[k])
    M += np.diag(model.armature)
    return tau, dq, dv, M


def main():
    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    g = oracle.lib.oracle_rnea_derivatives
    g.argtypes = [C.c_void_p] + [A._dp] * 7
    for name in ("kinova.urdf", "talos_reduced_armfixed_floatingbase.urdf", "g1_22dof_floatingbase.urdf",
                 "digit-v3-armfixedspecific-floatingbase-springfixed.urdf"):
        m = load_urdf(os.path.join(ROOT, "tests", "robots", name))
        rng = np.random.default_rng(0)
        m.armature = rng.uniform(0, 0.2, m.nv)
        n = m.nv
        mh = oracle.model_create(m)
        q, qd, qdd = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        t0, dq0, dv0, M0 = np.zeros(n), np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, n))
        oracle.check(g(mh.handle, A._ptr(q), A._ptr(qd), A._ptr(qdd), A._ptr(t0), A._ptr(dq0), A._ptr(dv0), A._ptr(M0)))
        t, dq, dv, M = world_derivs(m, q, qd, qdd)
        sc = lambda X, Y: np.abs(X - Y).max() / (1 + np.abs(Y).max())
        print(f"{name[:24]:24s} nv={n:2d} tau {sc(t, t0):.1e} dq {sc(dq, dq0):.1e} dv {sc(dv, dv0):.1e} M {sc(M, M0):.1e}")


if __name__ == "__main__":
    main()
