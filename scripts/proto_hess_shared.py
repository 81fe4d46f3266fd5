#!/usr/bin/env python3
"""Prototype (numpy, one sample) of the shared-chain form of the torque part of Optimizer::eval_h for the Kinova problem:
the value chain and the seven first-order chains are walked once, every pair (a <= b) keeps only its mixed second-order
state from link b on.  Checked here against a hyper-dual evaluation of the same recursion (the arithmetic the present
kinova_hessian_kernel performs per pair) before the HIP version is written.  Not part of the product or the tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raptor_amd


class HD:
    __slots__ = ("v", "a", "b", "ab")

    def __init__(self, v, a=0.0, b=0.0, ab=0.0):
        self.v, self.a, self.b, self.ab = v, a, b, ab

    def __add__(self, o):
        o = o if isinstance(o, HD) else HD(o)
        return HD(self.v + o.v, self.a + o.a, self.b + o.b, self.ab + o.ab)

    __radd__ = __add__

    def __sub__(self, o):
        o = o if isinstance(o, HD) else HD(o)
        return HD(self.v - o.v, self.a - o.a, self.b - o.b, self.ab - o.ab)

    def __rsub__(self, o):
        return HD(o) - self

    def __neg__(self):
        return HD(-self.v, -self.a, -self.b, -self.ab)

    def __mul__(self, o):
        o = o if isinstance(o, HD) else HD(o)
        return HD(self.v * o.v, self.a * o.v + self.v * o.a, self.b * o.v + self.v * o.b,
                  self.ab * o.v + self.a * o.b + self.b * o.a + self.v * o.ab)

    __rmul__ = __mul__


def cross(x, y):
    return [x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0]]


def add(x, y):
    return [x[0] + y[0], x[1] + y[1], x[2] + y[2]]


def mulT(R, x):
    return [R[0] * x[0] + R[3] * x[1] + R[6] * x[2], R[1] * x[0] + R[4] * x[1] + R[7] * x[2], R[2] * x[0] + R[5] * x[1] + R[8] * x[2]]


def symmul(Ic, x):
    return [Ic[0] * x[0] + Ic[1] * x[1] + Ic[2] * x[2], Ic[1] * x[0] + Ic[3] * x[1] + Ic[4] * x[2], Ic[2] * x[0] + Ic[4] * x[1] + Ic[5] * x[2]]


def dot(x, y):
    return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]


class Model:
    def __init__(self, m):
        self.nv = m.nv
        self.jtype = [int(t) for t in m.jtype]
        self.plR = [list(np.asarray(m.placement_R).reshape(m.nv, 9)[i]) for i in range(m.nv)]
        self.plp = [list(np.asarray(m.placement_p).reshape(m.nv, 3)[i]) for i in range(m.nv)]
        self.mass = [float(x) for x in m.mass]
        self.lever = [list(np.asarray(m.lever).reshape(m.nv, 3)[i]) for i in range(m.nv)]
        I = np.asarray(m.inertia).reshape(m.nv, 9)
        self.Ic = [[I[i][0], I[i][1], I[i][2], I[i][4], I[i][5], I[i][8]] for i in range(m.nv)]
        self.ng = [-float(g) for g in m.gravity]


def sincos(x):
    if isinstance(x, HD):
        s, c = np.sin(x.v), np.cos(x.v)
        return (HD(s, c * x.a, c * x.b, c * x.ab - s * x.a * x.b), HD(c, -s * x.a, -s * x.b, -s * x.ab - c * x.a * x.b))
    return np.sin(x), np.cos(x)


def phi_generic(M, q, qd, qdd, mu):
    """mu^T tau by the forward recursion and the virtual power, any number type (the present kernel's arithmetic)."""
    va, vl, aa, wa, wl = [0.0] * 3, [0.0] * 3, [0.0] * 3, [0.0] * 3, [0.0] * 3
    al = list(M.ng)
    phi = 0.0
    for l in range(M.nv):
        p = M.plp[l]
        jt = M.jtype[l]
        sg = -1.0 if jt < 0 else 1.0
        K = abs(jt) - 1
        I, J = (K + 1) % 3, (K + 2) % 3
        sq, cq = sincos(q[l])
        C, Sn = cq, sg * sq
        out = []
        for ang, lin in ((va, vl), (aa, al), (wa, wl)):
            lin = mulT(M.plR[l], add(lin, cross(ang, p)))
            ang = mulT(M.plR[l], ang)
            for x in (ang, lin):
                xi = C * x[I] + Sn * x[J]
                xj = C * x[J] - Sn * x[I]
                x[I], x[J] = xi, xj
            out.append((ang, lin))
        (va, vl), (aa, al), (wa, wl) = out
        vj = sg * qd[l]
        va[K] = va[K] + vj
        aa[I] = aa[I] + va[J] * vj
        aa[J] = aa[J] - va[I] * vj
        al[I] = al[I] + vl[J] * vj
        al[J] = al[J] - vl[I] * vj
        aa[K] = aa[K] + sg * qdd[l]
        wa[K] = wa[K] + sg * mu[l]
        c, mass = M.lever[l], M.mass[l]
        hl = [mass * x for x in add(vl, cross(va, c))]
        gl = [mass * x for x in add(al, cross(aa, c))]
        ha = add(symmul(M.Ic[l], va), cross(c, hl))
        fl = add(gl, cross(va, hl))
        fa = add(add(symmul(M.Ic[l], aa), cross(c, gl)), add(cross(va, ha), cross(vl, hl)))
        phi = phi + dot(wl, fl) + dot(wa, fa)
    return phi


def hess_reference(M, q, qd, qdd, mu, w):
    n = M.nv
    H = np.zeros((n, n))
    for a in range(n):
        for b in range(a, n):
            Q = [HD(q[l], w[0] * (l == a), w[0] * (l == b)) for l in range(n)]
            QD = [HD(qd[l], w[1] * (l == a), w[1] * (l == b)) for l in range(n)]
            QDD = [HD(qdd[l], w[2] * (l == a), w[2] * (l == b)) for l in range(n)]
            H[a, b] = phi_generic(M, Q, QD, QDD, mu).ab
    return H


# ---- the shared form ----
def v3(x):
    return np.asarray(x, dtype=float)


def inertia_apply(M, l, ang, lin):
    """I [ang; lin] -> (angular, linear) force parts, I about the link origin from (mass, lever, Ic)."""
    c, mass = v3(M.lever[l]), M.mass[l]
    hl = mass * (lin + np.cross(ang, c))
    ha = v3(symmul(M.Ic[l], ang)) + np.cross(c, hl)
    return ha, hl


def hess_shared(M, q, qd, qdd, mu, w):
    n = M.nv
    w0, w1, w2 = w
    H = np.zeros((n, n))
    # chain state [c][vec] with c = 0 the value chain, c = 1 + a the first-order chain of z_a; vec: va vl aa al wa wl
    X = np.zeros((n + 1, 6, 3))
    X[0, 3] = M.ng
    AB = {}  # (a, b) -> mixed state [6][3]
    for l in range(n):
        p = v3(M.plp[l])
        R = v3(M.plR[l]).reshape(3, 3)
        jt = M.jtype[l]
        sg = -1.0 if jt < 0 else 1.0
        K = abs(jt) - 1
        e = np.zeros(3)
        e[K] = sg
        th = sg * q[l]
        # Rot: x[I] = C y[I] + Sn y[J], x[J] = C y[J] - Sn y[I]  == rotation about e_K by -theta ... as a matrix E:
        C, Sn = np.cos(th), np.sin(th)
        I_, J_ = (K + 1) % 3, (K + 2) % 3
        Rot = np.eye(3)
        Rot[I_, I_], Rot[I_, J_], Rot[J_, I_], Rot[J_, J_] = C, Sn, -Sn, C
        Et = Rot @ R.T

        def plain_step(S):  # Xup applied to the three motion vectors of one chain (any derivative order: linear)
            o = np.zeros((6, 3))
            for k in (0, 2, 4):
                o[k + 1] = Et @ (S[k + 1] + np.cross(S[k], p))
                o[k] = Et @ S[k]
            return o

        vj = qd[l]  # times e
        # value chain
        Y = plain_step(X[0])
        Y[0] += e * vj
        Y[2] += np.cross(Y[0], e) * vj + e * qdd[l]
        Y[3] += np.cross(Y[1], e) * vj
        Y[4] += e * mu[l]
        X[0] = Y
        Vv = Y
        # first-order chains: zero below their own link
        for a in range(n):
            if l < a:
                continue
            Ya = plain_step(X[1 + a])
            if l == a:  # seeded here: w0 d(Xup u)/dq = w0 (Xup u) x e on every vector, plus the joint's own motion
                Vt = Vv[0] - e * vj  # Xup V_parent, value
                At_a = Vv[2] - e * qdd[l] - np.cross(Vv[0], e) * vj
                At_l = Vv[3] - np.cross(Vv[1], e) * vj
                Wt = Vv[4] - e * mu[l]
                Ya[0] = w0 * np.cross(Vt, e) + e * w1
                Ya[1] = w0 * np.cross(Vv[1], e)
                Ya[2] = w0 * np.cross(At_a, e) + e * w2
                Ya[3] = w0 * np.cross(At_l, e)
                Ya[4] = w0 * np.cross(Wt, e)
                Ya[5] = w0 * np.cross(Vv[5], e)
                Ya[2] += np.cross(Ya[0], e) * vj + np.cross(Vv[0], e) * w1
                Ya[3] += np.cross(Ya[1], e) * vj + np.cross(Vv[1], e) * w1
            else:
                Ya[2] += np.cross(Ya[0], e) * vj
                Ya[3] += np.cross(Ya[1], e) * vj
            X[1 + a] = Ya
        # per-chain force quantities of this link
        Hv = inertia_apply(M, l, Vv[0], Vv[1])
        fv_a, fv_l = inertia_apply(M, l, Vv[2], Vv[3])
        fv_a = fv_a + np.cross(Vv[0], Hv[0]) + np.cross(Vv[1], Hv[1])
        fv_l = fv_l + np.cross(Vv[0], Hv[1])
        IW = inertia_apply(M, l, Vv[4], Vv[5])
        # cvec: W . (Vab x* I V + V x* I Vab) = Vab . cvec
        c1a = np.cross(Hv[0], Vv[4]) + np.cross(Hv[1], Vv[5])
        c1l = np.cross(Hv[1], Vv[4])
        mWa = np.cross(Vv[0], Vv[4])                       # crm(V) W
        mWl = np.cross(Vv[0], Vv[5]) + np.cross(Vv[1], Vv[4])
        c2 = inertia_apply(M, l, mWa, mWl)
        cva, cvl = c1a - c2[0], c1l - c2[1]
        per = {}
        for a in range(l + 1):
            S = X[1 + a]
            Ha = inertia_apply(M, l, S[0], S[1])
            fa_a, fa_l = inertia_apply(M, l, S[2], S[3])
            # f^a = I A^a + V^a x* I V + V x* I V^a
            fa_a = fa_a + np.cross(S[0], Hv[0]) + np.cross(S[1], Hv[1]) + np.cross(Vv[0], Ha[0]) + np.cross(Vv[1], Ha[1])
            fa_l = fa_l + np.cross(S[0], Hv[1]) + np.cross(Vv[0], Ha[1])
            Ua = np.cross(S[0], Vv[4])
            Ul = np.cross(S[0], Vv[5]) + np.cross(S[1], Vv[4])
            per[a] = (Ha, (fa_a, fa_l), (Ua, Ul))
        for a in range(l + 1):
            for b in range(a, l + 1):
                Sa = X[1 + a]
                if l == b:
                    delta = 1.0 if a == b else 0.0
                    T = np.zeros((6, 3))
                    T[0] = w0 * np.cross(Sa[0], e)
                    T[1] = w0 * np.cross(Sa[1], e)
                    T[4] = w0 * np.cross(Sa[4], e)
                    T[5] = w0 * np.cross(Sa[5], e)
                    Aa = Sa[2] - np.cross(Sa[0], e) * vj - delta * np.cross(Vv[0], e) * w1
                    Al = Sa[3] - np.cross(Sa[1], e) * vj - delta * np.cross(Vv[1], e) * w1
                    T[2] = w0 * np.cross(Aa, e) + np.cross(T[0], e) * vj + (1 + delta) * np.cross(Sa[0], e) * w1
                    T[3] = w0 * np.cross(Al, e) + np.cross(T[1], e) * vj + (1 + delta) * np.cross(Sa[1], e) * w1
                else:
                    T = plain_step(AB[(a, b)])
                    T[2] += np.cross(T[0], e) * vj
                    T[3] += np.cross(T[1], e) * vj
                AB[(a, b)] = T
                Sb = X[1 + b]
                Ha_, fa_, Ua_ = per[a]
                Hb_, fb_, Ub_ = per[b]
                t3 = dot(T[4], fv_a) + dot(T[5], fv_l) + dot(IW[0], T[2]) + dot(IW[1], T[3]) + dot(cva, T[0]) + dot(cvl, T[1])
                t1 = dot(Sa[4], fb_[0]) + dot(Sa[5], fb_[1]) + dot(Sb[4], fa_[0]) + dot(Sb[5], fa_[1])
                t2 = -(dot(Ua_[0], Hb_[0]) + dot(Ua_[1], Hb_[1]) + dot(Ub_[0], Ha_[0]) + dot(Ub_[1], Ha_[1]))
                H[a, b] += t1 + t2 + t3
    return H


def crm(xa, xl, ya, yl):
    return np.cross(xa, ya), np.cross(xa, yl) + np.cross(xl, ya)


def crf(xa, xl, fa, fl):
    return np.cross(xa, fa) + np.cross(xl, fl), np.cross(xa, fl)


def hess_stateless(M, q, qd, qdd, mu, w):
    """The form the kernel uses: no mixed state.  The mixed chain of (a, b) at link l is a cross product of chain a's state with
    joint b's axis carried to link l (J_b) and its rate (Jd_b):  V^ab = w0 V^a x J_b,  W^ab = w0 W^a x J_b,
    A^ab = (w0 A^a + w1 V^a) x J_b + w0 V^a x Jd_b,  and every pair term is a product of a covector of chain a with a vector of chain b."""
    n = M.nv
    w0, w1, w2 = w
    H = np.zeros((n, n))
    X = np.zeros((n + 1, 10, 3))  # va vl aa al wa wl ja jl jda jdl
    X[0, 3] = M.ng
    for l in range(n):
        p = v3(M.plp[l])
        R = v3(M.plR[l]).reshape(3, 3)
        jt = M.jtype[l]
        sg = -1.0 if jt < 0 else 1.0
        K = abs(jt) - 1
        e = np.zeros(3)
        e[K] = sg
        th = sg * q[l]
        C, Sn = np.cos(th), np.sin(th)
        I_, J_ = (K + 1) % 3, (K + 2) % 3
        Rot = np.eye(3)
        Rot[I_, I_], Rot[I_, J_], Rot[J_, I_], Rot[J_, J_] = C, Sn, -Sn, C
        Et = Rot @ R.T

        def step(S):
            o = np.zeros_like(S)
            for k in range(0, S.shape[0], 2):
                o[k + 1] = Et @ (S[k + 1] + np.cross(S[k], p))
                o[k] = Et @ S[k]
            return o

        vj = qd[l]
        Y = step(X[0])
        Y[0] += e * vj
        Y[2] += np.cross(Y[0], e) * vj + e * qdd[l]
        Y[3] += np.cross(Y[1], e) * vj
        Y[4] += e * mu[l]
        X[0] = Y
        Vv = Y
        for a in range(n):
            if l < a:
                continue
            Ya = step(X[1 + a])
            Ya[2] += np.cross(Ya[0], e) * vj
            Ya[3] += np.cross(Ya[1], e) * vj
            Ya[8] += np.cross(Ya[6], e) * vj
            Ya[9] += np.cross(Ya[7], e) * vj
            if l == a:
                Ya[0] = w0 * np.cross(Vv[0], e) + e * w1
                Ya[1] = w0 * np.cross(Vv[1], e)
                Ya[2] = w0 * np.cross(Vv[2] - np.cross(Vv[0], e) * vj, e) + e * w2 + np.cross(Ya[0], e) * vj + np.cross(Vv[0], e) * w1
                Ya[3] = w0 * np.cross(Vv[3] - np.cross(Vv[1], e) * vj, e) + np.cross(Ya[1], e) * vj + np.cross(Vv[1], e) * w1
                Ya[4] = w0 * np.cross(Vv[4], e)
                Ya[5] = w0 * np.cross(Vv[5], e)
                Ya[6], Ya[7], Ya[8], Ya[9] = e, 0 * e, 0 * e, 0 * e
            X[1 + a] = Ya
        # value-level quantities
        Hv = inertia_apply(M, l, Vv[0], Vv[1])
        fv = inertia_apply(M, l, Vv[2], Vv[3])
        t = crf(Vv[0], Vv[1], *Hv)
        fv = (fv[0] + t[0], fv[1] + t[1])
        IW = inertia_apply(M, l, Vv[4], Vv[5])
        mW = crm(Vv[0], Vv[1], Vv[4], Vv[5])
        c2 = inertia_apply(M, l, *mW)
        cv = (np.cross(Hv[0], Vv[4]) + np.cross(Hv[1], Vv[5]) - c2[0], np.cross(Hv[1], Vv[4]) - c2[1])
        rec = {}
        for a in range(l + 1):
            S = X[1 + a]
            Hc = inertia_apply(M, l, S[0], S[1])
            fc = inertia_apply(M, l, S[2], S[3])
            t1 = crf(S[0], S[1], *Hv)
            t2 = crf(Vv[0], Vv[1], *Hc)
            fc = (fc[0] + t1[0] + t2[0], fc[1] + t1[1] + t2[1])
            # G = -[w0 W^a x* fv + (w0 A^a + w1 V^a) x* IW + w0 V^a x* cv],  G2 = -w0 V^a x* IW
            g1 = crf(w0 * S[4], w0 * S[5], *fv)
            g2 = crf(w0 * S[2] + w1 * S[0], w0 * S[3] + w1 * S[1], *IW)
            g3 = crf(w0 * S[0], w0 * S[1], *cv)
            G = (-(g1[0] + g2[0] + g3[0]), -(g1[1] + g2[1] + g3[1]))
            g4 = crf(w0 * S[0], w0 * S[1], *IW)
            G2 = (-g4[0], -g4[1])
            # Q: U^a.H^b + U^b.H^a = V^b . Q^a,  Q^a = -I (W x V^a) + W x* H^a
            wxv = crm(Vv[4], Vv[5], S[0], S[1])
            q1 = inertia_apply(M, l, *wxv)
            q2 = crf(Vv[4], Vv[5], *Hc)
            Q = (q2[0] - q1[0], q2[1] - q1[1])
            rec[a] = dict(G=G, G2=G2, W=(S[4], S[5]), f=fc, Q=Q, J=(S[6], S[7]), Jd=(S[8], S[9]), V=(S[0], S[1]))

        def d6(x, y):
            return dot(x[0], y[0]) + dot(x[1], y[1])

        for a in range(l + 1):
            for b in range(a, l + 1):
                A_, B_ = rec[a], rec[b]
                H[a, b] += d6(B_["J"], A_["G"]) + d6(B_["Jd"], A_["G2"]) + d6(B_["f"], A_["W"]) + d6(B_["W"], A_["f"]) - d6(B_["V"], A_["Q"])
    return H


def run(trials=4, seed=3, verbose=False):
    """worst relative difference of the two shared forms from the hyper-dual recursion over `trials` random states"""
    m = raptor_amd.load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
    M = Model(m)
    rng = np.random.default_rng(seed)
    worst = 0.0
    for trial in range(trials):
        q, qd, qdd, mu = (rng.uniform(-1, 1, M.nv) for _ in range(4))
        w = rng.uniform(0.2, 1.5, 3)
        Href = hess_reference(M, q, qd, qdd, mu, w)
        scale = np.abs(Href).max()
        err = np.abs(hess_shared(M, q, qd, qdd, mu, w) - Href).max() / scale
        err2 = np.abs(hess_stateless(M, q, qd, qdd, mu, w) - Href).max() / scale
        if verbose:
            print(f"trial {trial}: max |H| {scale:.3e}  mixed chains kept as state {err:.2e}  stateless form (the kernel's) {err2:.2e}")
        worst = max(worst, err, err2)
    return worst


def main():
    print("worst", run(verbose=True))


if __name__ == "__main__":
    main()
