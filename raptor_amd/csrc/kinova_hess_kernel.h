// K_kinova_hess : Optimizer::eval_h for KinovaOptimizer, fused -- the Hessian of the Lagrangian of one decision vector,
//     H = obj_factor * hess_f + sum_r lambda_r * pg_pz_pz(r)                         (Optimization/src/Optimizer.cpp:499-635)
// leaves the kernel as its 28 upper-triangle entries; the per-row Hessians pg_pz_pz (m x 7 x 7 doubles per z in the
// reference: 1.9 MB at N = 128) never exist.
//
// MI355X design (not a translation of the reference's tensor contractions):
//   * the multiplier-weighted sum of the torque rows is the Hessian of ONE scalar per sample,
//         phi(z) = mu^T tau(q(z), q_d(z), q_dd(z)),   mu = lambda of the sample's TorqueLimits rows,
//     and mu^T tau = sum_l w_l . f_l with w_l the spatial velocity link l would have under joint rates mu
//     (virtual power), so phi needs only the FORWARD recursion of RNEA: v_l, a_l, w_l, f_l = I a + v x* I v.
//     pinocchio::ComputeRNEASecondOrderDerivatives' four nv^3 tensors and the dense products of
//     InverseDynamics.cpp:141-171 are never formed.  ArmourBezierCurves is linear in z with a diagonal pq_pz
//     (ArmourBezierCurves.cpp:147-151); damping, friction and armature are linear in (q_d, q_dd): no Hessian.
//     That part is kinova_hess_torque_kernel.h (round 3: every derivative chain of a sample walked once).  The hyper-dual
//     recursion below (first-order and hyper-dual numbers carried through the links in link-local frames) is what
//     kinova_hess_blocks_kernel.h builds the longer-horizon problem's per-sample blocks from;
//   * kinova_hessian_kernel here: the obstacle rows and hess_f.  One workgroup (4 wavefronts) per decision vector, 64 samples per
//     pass: trajectory values and sincos once per (sample, joint) into LDS (sample-minor: conflict-free), world-frame joint axes
//     and sphere-link poses by the first wavefront, then a wavefront owns one sphere (its joint is wave-uniform), lanes = samples
//     (KinovaCustomizedConstraints.cpp:117-155, BoxCollisionAvoidance.cpp:206-288, 451-474): arg-min box with the Jacobian
//     kernel's code and tie rules, the distance's gradient n and Hessian K = sign/d (P - n n^T) w.r.t. the centre (P = the
//     reference's clamped-axis projector), then  lambda s^2 (J_a^T K J_b + n . H_ab),  J_c = w_c x (p - o_c),  H_ab = w_a x J_b
//     (a <= b): the FK Hessian of ForwardKinematics.cpp:386-402 in closed form from the world-frame joint axes of that sample.
//     The 28 sums stay in the sample's lane over the passes; one wave reduction at the end, then a fixed-order sum over the
//     wavefronts: deterministic, no atomics.
#ifndef RAPTOR_KINOVA_HESS_KERNEL_H
#define RAPTOR_KINOVA_HESS_KERNEL_H

#include <type_traits>

#include "kinova_kernel.h"
#include "legged_kernel.h"  // wrap_to_pi

#define H_CH 64      // samples per pass = lanes of a wavefront
#define H_PAIRS 28   // 7 * 8 / 2
#define H_WAVES 4
// LDS (doubles), every table sample-minor: [field][64]
#define HL_X 0                        // per joint: q_d, q_dd, cos q, sin q, q        35 x 64
#define HL_MU (35 * H_CH)             // multipliers of the torque rows               7 x 64
#define HL_WS (HL_MU + 7 * H_CH)      // pq_pz, pq_d_pz, pq_dd_pz diagonal            3 x 64
#define HL_W (HL_WS + 3 * H_CH)       // per joint: world axis (3), world origin (3)  42 x 64
#define HL_RO (HL_W + 42 * H_CH)      // R, o of up to 4 sphere-carrying links        48 x 64
// the torque phase keeps each wavefront's hyper-dual virtual velocity (wa, wl: 6 x 4 doubles per sample) in LDS, [field][64],
// on top of the obstacle tables HL_W / HL_RO (filled after the torque phase): 4 x 24 = 96 fields over 90 + 6
#define HL_WST HL_W
#define HL_ACC (HL_W + H_WAVES * 24 * H_CH)  // per wavefront: 28 obstacle partial sums, then 28 torque sums
#define HL_TOTAL (HL_ACC + 2 * H_WAVES * H_PAIRS)
static_assert(HL_RO + 48 * H_CH <= HL_ACC, "obstacle tables fit under the w state");

// first-order dual and hyper-dual numbers
struct D1 {
    double v, a;
};
struct HD {
    double v, a, b, ab;
};
RD D1 operator+(D1 x, D1 y) { return D1{x.v + y.v, x.a + y.a}; }
RD D1 operator-(D1 x, D1 y) { return D1{x.v - y.v, x.a - y.a}; }
RD D1 operator*(D1 x, D1 y) { return D1{x.v * y.v, x.a * y.v + x.v * y.a}; }
RD D1 operator*(double s, D1 x) { return D1{s * x.v, s * x.a}; }
RD D1 operator*(D1 x, double s) { return D1{s * x.v, s * x.a}; }
RD D1 operator+(D1 x, double y) { return D1{x.v + y, x.a}; }
RD HD operator+(HD x, HD y) { return HD{x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab}; }
RD HD operator-(HD x, HD y) { return HD{x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab}; }
RD HD operator*(HD x, HD y) {
    return HD{x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b, x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab};
}
RD HD operator*(double s, HD x) { return HD{s * x.v, s * x.a, s * x.b, s * x.ab}; }
RD HD operator*(HD x, double s) { return HD{s * x.v, s * x.a, s * x.b, s * x.ab}; }
RD HD operator+(HD x, double y) { return HD{x.v + y, x.a, x.b, x.ab}; }
RD void add_value(double& x, double y) { x += y; }
RD void add_value(D1& x, double y) { x.v += y; }
RD void add_value(HD& x, double y) { x.v += y; }
RD D1 promote(double x, D1*) { return D1{x, 0.0}; }
RD HD promote(D1 x, HD*) { return HD{x.v, x.a, 0.0, 0.0}; }
RD HD promote(double x, HD*) { return HD{x, 0.0, 0.0, 0.0}; }

template <class S>
struct T3 {
    S c[3];  // compile-time indices only
};
template <class S>
RD T3<S> operator+(const T3<S>& x, const T3<S>& y) { return T3<S>{{x.c[0] + y.c[0], x.c[1] + y.c[1], x.c[2] + y.c[2]}}; }
template <class S>
RD T3<S> cross(const T3<S>& x, const T3<S>& y) {
    return T3<S>{{x.c[1] * y.c[2] - x.c[2] * y.c[1], x.c[2] * y.c[0] - x.c[0] * y.c[2], x.c[0] * y.c[1] - x.c[1] * y.c[0]}};
}
template <class S>
RD T3<S> cross(V3 p, const T3<S>& y) {  // constant x state
    return T3<S>{{p.y * y.c[2] - p.z * y.c[1], p.z * y.c[0] - p.x * y.c[2], p.x * y.c[1] - p.y * y.c[0]}};
}
template <class S>
RD T3<S> cross(const T3<S>& x, V3 p) {
    return T3<S>{{x.c[1] * p.z - x.c[2] * p.y, x.c[2] * p.x - x.c[0] * p.z, x.c[0] * p.y - x.c[1] * p.x}};
}
template <class S>
RD T3<S> mulT(const double* R, const T3<S>& x) {  // R^T x, constant row-major R
    return T3<S>{{R[0] * x.c[0] + R[3] * x.c[1] + R[6] * x.c[2], R[1] * x.c[0] + R[4] * x.c[1] + R[7] * x.c[2],
                  R[2] * x.c[0] + R[5] * x.c[1] + R[8] * x.c[2]}};
}
template <class S>
RD T3<S> symmul(const double* Ic, const T3<S>& x) {  // (xx, xy, xz, yy, yz, zz)
    return T3<S>{{Ic[0] * x.c[0] + Ic[1] * x.c[1] + Ic[2] * x.c[2], Ic[1] * x.c[0] + Ic[3] * x.c[1] + Ic[4] * x.c[2],
                  Ic[2] * x.c[0] + Ic[4] * x.c[1] + Ic[5] * x.c[2]}};
}
RD double dot_ab(const T3<HD>& x, const T3<HD>& y) {  // the mixed part of x . y
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) s += x.c[k].v * y.c[k].ab + x.c[k].a * y.c[k].b + x.c[k].b * y.c[k].a + x.c[k].ab * y.c[k].v;
    return s;
}

// spatial velocity, acceleration and virtual velocity of the link being walked, [angular; linear], link-local frame
template <class S>
struct Kin {
    T3<S> va, vl, aa, al, wa, wl;
};
template <class S, class P>
RD Kin<S> promote(const Kin<P>& k) {
    Kin<S> r;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        r.va.c[i] = promote(k.va.c[i], (S*)nullptr); r.vl.c[i] = promote(k.vl.c[i], (S*)nullptr);
        r.aa.c[i] = promote(k.aa.c[i], (S*)nullptr); r.al.c[i] = promote(k.al.c[i], (S*)nullptr);
        r.wa.c[i] = promote(k.wa.c[i], (S*)nullptr); r.wl.c[i] = promote(k.wl.c[i], (S*)nullptr);
    }
    return r;
}

template <class S>
RD T3<S> mulM(const double (&A)[9], const T3<S>& x) {  // A x, row-major A
    return T3<S>{{A[0] * x.c[0] + A[1] * x.c[1] + A[2] * x.c[2], A[3] * x.c[0] + A[4] * x.c[1] + A[5] * x.c[2],
                  A[6] * x.c[0] + A[7] * x.c[1] + A[8] * x.c[2]}};
}
// Past the seeded links the joint's sine / cosine are plain numbers: rotation about local axis K after the fixed placement
// is ONE matrix per (sample, link), Et = Rot_K(theta)^T plR^T, applied to every motion vector (36 instead of 52 multiply-adds
// per hyper-dual vector).
template <int K>
RD void make_Et(const double* R, double C, double Sn, double (&Et)[9]) {
    constexpr int I = (K + 1) % 3, J = (K + 2) % 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        Et[I * 3 + c] = C * R[c * 3 + I] + Sn * R[c * 3 + J];
        Et[J * 3 + c] = C * R[c * 3 + J] - Sn * R[c * 3 + I];
        Et[K * 3 + c] = R[c * 3 + K];
    }
}
RD void make_Et(const double* R, int aj, double C, double Sn, double (&Et)[9]) {
    if (aj == 3) make_Et<2>(R, C, Sn, Et);
    else if (aj == 2) make_Et<1>(R, C, Sn, Et);
    else make_Et<0>(R, C, Sn, Et);
}

// Joint about local axis K (Spatial.h:115-258 jcalc, Xup = XJ * Xtree): rotate the parent's motion vectors into the link
// frame, add the joint's own motion S = sg e_K.  C = cos(theta), Sn = sin(theta), theta = sg q; Tq = double past the seeded link.
template <int K, class S, class Tq>
RD void hess_motion(double sg, Tq qd, Tq qdd, double mu, Kin<S>& k) {
    constexpr int I = (K + 1) % 3, J = (K + 2) % 3;
    const Tq vj = sg * qd;
    k.va.c[K] = k.va.c[K] + vj;  // v_l = Xup v_parent + S q_d                 (CustomizedInverseDynamics.cpp:155)
    // a_l = Xup a_parent + crm(v_l) (S q_d) + S q_dd                            (:157)
    k.aa.c[I] = k.aa.c[I] + k.va.c[J] * vj;
    k.aa.c[J] = k.aa.c[J] - k.va.c[I] * vj;
    k.al.c[I] = k.al.c[I] + k.vl.c[J] * vj;
    k.al.c[J] = k.al.c[J] - k.vl.c[I] * vj;
    k.aa.c[K] = k.aa.c[K] + sg * qdd;
    add_value(k.wa.c[K], sg * mu);  // virtual velocity under joint rates mu
}
template <int K, class S, class Tq>
RD void hess_joint(Tq C, Tq Sn, double sg, Tq qd, Tq qdd, double mu, Kin<S>& k) {
    constexpr int I = (K + 1) % 3, J = (K + 2) % 3;
#define HROT(x) { const S xi_ = C * x.c[I] + Sn * x.c[J]; const S xj_ = C * x.c[J] - Sn * x.c[I]; x.c[I] = xi_; x.c[J] = xj_; }
    HROT(k.va) HROT(k.vl) HROT(k.aa) HROT(k.al) HROT(k.wa) HROT(k.wl)
#undef HROT
    hess_motion<K>(sg, qd, qdd, mu, k);
}
template <class S, class Tq>
RD void hess_link(const DevModel* __restrict__ M, int l, Tq C, Tq Sn, Tq qd, Tq qdd, double mu, Kin<S>& k) {
    // Xtree = plux(R^T, p): [w; v] -> [R^T w; R^T (v + w x p)]   (CustomizedInverseDynamics.cpp:43-47, Utils.h:185-191)
    const V3 p = v3(M->plp[l][0], M->plp[l][1], M->plp[l][2]);
    const int jt = M->jtype[l];
    const double sg = jt < 0 ? -1.0 : 1.0;
    const int aj = jt < 0 ? -jt : jt;
    if constexpr (std::is_same<Tq, double>::value) {  // past the seeded links: placement and rotation as one matrix
        double Et[9];
        make_Et(M->plR[l], aj, C, Sn, Et);
        k.vl = mulM(Et, k.vl + cross(k.va, p));
        k.va = mulM(Et, k.va);
        k.al = mulM(Et, k.al + cross(k.aa, p));
        k.aa = mulM(Et, k.aa);
        k.wl = mulM(Et, k.wl + cross(k.wa, p));
        k.wa = mulM(Et, k.wa);
        if (aj == 3) hess_motion<2>(sg, qd, qdd, mu, k);
        else if (aj == 2) hess_motion<1>(sg, qd, qdd, mu, k);
        else hess_motion<0>(sg, qd, qdd, mu, k);
    } else {
        k.vl = mulT(M->plR[l], k.vl + cross(k.va, p));
        k.va = mulT(M->plR[l], k.va);

        k.al = mulT(M->plR[l], k.al + cross(k.aa, p));
        k.aa = mulT(M->plR[l], k.aa);

        k.wl = mulT(M->plR[l], k.wl + cross(k.wa, p));
        k.wa = mulT(M->plR[l], k.wa);

        if (aj == 3) hess_joint<2>(C, Sn, sg, qd, qdd, mu, k);
        else if (aj == 2) hess_joint<1>(C, Sn, sg, qd, qdd, mu, k);
        else hess_joint<0>(C, Sn, sg, qd, qdd, mu, k);
    }
}
// The hyper-dual regime keeps v, a in registers and the virtual velocity w in LDS (`ws`: this lane's column of the wavefront's
// [24][64] area: field = (vector * 3 + component) * 4 + part): 72 hyper-dual doubles do not fit 256 registers with the force
// temporaries beside them, and two wavefronts per SIMD need that.
RD T3<HD> w_load(const double* ws, int vec) {
    T3<HD> x;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double* f = ws + ((vec * 3 + i) * 4) * H_CH;
        x.c[i] = HD{f[0], f[H_CH], f[2 * H_CH], f[3 * H_CH]};
    }
    return x;
}
RD void w_store(double* ws, int vec, const T3<HD>& x) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double* f = ws + ((vec * 3 + i) * 4) * H_CH;
        f[0] = x.c[i].v; f[H_CH] = x.c[i].a; f[2 * H_CH] = x.c[i].b; f[3 * H_CH] = x.c[i].ab;
    }
}
template <class S>
struct KinVA {
    T3<S> va, vl, aa, al;
};
template <int K, class X>
RD void hess_rot(const HD C, const HD Sn, X& x) {
    constexpr int I = (K + 1) % 3, J = (K + 2) % 3;
    const HD xi_ = C * x.c[I] + Sn * x.c[J];
    const HD xj_ = C * x.c[J] - Sn * x.c[I];
    x.c[I] = xi_;
    x.c[J] = xj_;
}
template <int K, class Tq>
RD void hess_motion_va(double sg, Tq qd, Tq qdd, KinVA<HD>& k) {  // the joint's own motion, after the rotation (:155-157)
    constexpr int I = (K + 1) % 3, J = (K + 2) % 3;
    const Tq vj = sg * qd;
    k.va.c[K] = k.va.c[K] + vj;
    k.aa.c[I] = k.aa.c[I] + k.va.c[J] * vj;
    k.aa.c[J] = k.aa.c[J] - k.va.c[I] * vj;
    k.al.c[I] = k.al.c[I] + k.vl.c[J] * vj;
    k.al.c[J] = k.al.c[J] - k.vl.c[I] * vj;
    k.aa.c[K] = k.aa.c[K] + sg * qdd;
}
// one link of the hyper-dual regime: w in LDS first (its registers are free again before the forces are formed), then v, a
RD void hess_link_hd(const DevModel* __restrict__ M, int l, double C, double Sn, double qd, double qdd, double mu, KinVA<HD>& k,
          double* ws) {
    const V3 p = v3(M->plp[l][0], M->plp[l][1], M->plp[l][2]);
    const int jt = M->jtype[l];
    const double sg = jt < 0 ? -1.0 : 1.0;
    const int aj = jt < 0 ? -jt : jt;
    double Et[9];
    make_Et(M->plR[l], aj, C, Sn, Et);
    {
        T3<HD> wa = w_load(ws, 0), wl = w_load(ws, 1);
        wl = mulM(Et, wl + cross(wa, p));
        wa = mulM(Et, wa);
        if (aj == 3) add_value(wa.c[2], sg * mu);
        else if (aj == 2) add_value(wa.c[1], sg * mu);
        else add_value(wa.c[0], sg * mu);
        w_store(ws, 0, wa);
        w_store(ws, 1, wl);
    }
    asm volatile("" ::: "memory");  // the forces below read w back from LDS: no forwarding of the 24 values through registers
    k.vl = mulM(Et, k.vl + cross(k.va, p));
    k.va = mulM(Et, k.va);
    k.al = mulM(Et, k.al + cross(k.aa, p));
    k.aa = mulM(Et, k.aa);
    if (aj == 3) hess_motion_va<2>(sg, qd, qdd, k);
    else if (aj == 2) hess_motion_va<1>(sg, qd, qdd, k);
    else hess_motion_va<0>(sg, qd, qdd, k);
}
// the seeded link: sine / cosine carry derivative parts, placement and rotation stay two steps
RD void hess_link_hd(const DevModel* __restrict__ M, int l, HD C, HD Sn, HD qd, HD qdd, double mu, KinVA<HD>& k, double* ws) {
    const V3 p = v3(M->plp[l][0], M->plp[l][1], M->plp[l][2]);
    const int jt = M->jtype[l];
    const double sg = jt < 0 ? -1.0 : 1.0;
    const int aj = jt < 0 ? -jt : jt;
    {
        T3<HD> wa = w_load(ws, 0), wl = w_load(ws, 1);
        wl = mulT(M->plR[l], wl + cross(wa, p));
        wa = mulT(M->plR[l], wa);
        if (aj == 3) { hess_rot<2>(C, Sn, wa); hess_rot<2>(C, Sn, wl); add_value(wa.c[2], sg * mu); }
        else if (aj == 2) { hess_rot<1>(C, Sn, wa); hess_rot<1>(C, Sn, wl); add_value(wa.c[1], sg * mu); }
        else { hess_rot<0>(C, Sn, wa); hess_rot<0>(C, Sn, wl); add_value(wa.c[0], sg * mu); }
        w_store(ws, 0, wa);
        w_store(ws, 1, wl);
    }
    asm volatile("" ::: "memory");
    k.vl = mulT(M->plR[l], k.vl + cross(k.va, p));
    k.va = mulT(M->plR[l], k.va);
    k.al = mulT(M->plR[l], k.al + cross(k.aa, p));
    k.aa = mulT(M->plR[l], k.aa);
    if (aj == 3) { hess_rot<2>(C, Sn, k.va); hess_rot<2>(C, Sn, k.vl); hess_rot<2>(C, Sn, k.aa); hess_rot<2>(C, Sn, k.al); hess_motion_va<2>(sg, qd, qdd, k); }
    else if (aj == 2) { hess_rot<1>(C, Sn, k.va); hess_rot<1>(C, Sn, k.vl); hess_rot<1>(C, Sn, k.aa); hess_rot<1>(C, Sn, k.al); hess_motion_va<1>(sg, qd, qdd, k); }
    else { hess_rot<0>(C, Sn, k.va); hess_rot<0>(C, Sn, k.vl); hess_rot<0>(C, Sn, k.aa); hess_rot<0>(C, Sn, k.al); hess_motion_va<0>(sg, qd, qdd, k); }
}
RD double dot_ab_lds(const double* ws, int vec, const T3<HD>& y) {  // the mixed part of w . y, w read from LDS
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double* f = ws + ((vec * 3 + i) * 4) * H_CH;
        s += f[0] * y.c[i].ab + f[H_CH] * y.c[i].b + f[2 * H_CH] * y.c[i].a + f[3 * H_CH] * y.c[i].v;
    }
    return s;
}
RD double hess_power_ab_lds(const DevModel* __restrict__ M, int l, const KinVA<HD>& k, const double* ws) {
    const V3 c = v3(M->lever[l][0], M->lever[l][1], M->lever[l][2]);
    const double mass = M->mass[l];
    T3<HD> hl = k.vl + cross(k.va, c);  // momentum, linear part
    T3<HD> gl = k.al + cross(k.aa, c);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        hl.c[i] = mass * hl.c[i];
        gl.c[i] = mass * gl.c[i];
    }
    const T3<HD> fl = gl + cross(k.va, hl);
    double s = dot_ab_lds(ws, 1, fl);
    const T3<HD> ha = symmul(M->Ic[l], k.va) + cross(c, hl);
    const T3<HD> fa = symmul(M->Ic[l], k.aa) + cross(c, gl) + cross(k.va, ha) + cross(k.vl, hl);
    return s + dot_ab_lds(ws, 0, fa);
}

// mixed part of w . f,  f = I a + v x* (I v),  I = (mass, lever, inertia about the COM) in the link frame (:244-246)
RD double hess_power_ab(const DevModel* __restrict__ M, int l, const Kin<HD>& k) {
    const V3 c = v3(M->lever[l][0], M->lever[l][1], M->lever[l][2]);
    const double mass = M->mass[l];
    T3<HD> hl = k.vl + cross(k.va, c);  // momentum, linear part
    T3<HD> gl = k.al + cross(k.aa, c);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        hl.c[i] = mass * hl.c[i];
        gl.c[i] = mass * gl.c[i];
    }

    const T3<HD> ha = symmul(M->Ic[l], k.va) + cross(c, hl);
    const T3<HD> fl = gl + cross(k.va, hl);
    double s = dot_ab(k.wl, fl);

    const T3<HD> fa = symmul(M->Ic[l], k.aa) + cross(c, gl) + cross(k.va, ha) + cross(k.vl, hl);
    return s + dot_ab(k.wa, fa);
}

// ---- d2 phi / dq_a d(q_dd)_b: tau is linear in q_dd with the coefficient M(q), so  d phi / d(q_dd)_b = mu^T M(q) e_b = the virtual
// power of the inertial forces I_l b_l, b = the acceleration field of a unit q_dd at joint b alone (no velocities, no gravity).
// One first-order dual pass (seed q_a) over four motion vectors instead of a hyper-dual pass over six.
template <class S>
struct KinB {
    T3<S> ba, bl, wa, wl;
};
template <int K, class S, class Tq>
RD void qa_joint(Tq C, Tq Sn, double sg, double bseed, double mu, KinB<S>& k) {
    constexpr int I = (K + 1) % 3, J = (K + 2) % 3;
#define HROT(x) { const S xi_ = C * x.c[I] + Sn * x.c[J]; const S xj_ = C * x.c[J] - Sn * x.c[I]; x.c[I] = xi_; x.c[J] = xj_; }
    HROT(k.ba) HROT(k.bl) HROT(k.wa) HROT(k.wl)
#undef HROT
    add_value(k.ba.c[K], sg * bseed);
    add_value(k.wa.c[K], sg * mu);
}
template <class S, class Tq>
RD void qa_link(const DevModel* __restrict__ M, int l, Tq C, Tq Sn, double bseed, double mu, KinB<S>& k) {
    const V3 p = v3(M->plp[l][0], M->plp[l][1], M->plp[l][2]);
    k.bl = mulT(M->plR[l], k.bl + cross(k.ba, p));
    k.ba = mulT(M->plR[l], k.ba);
    k.wl = mulT(M->plR[l], k.wl + cross(k.wa, p));
    k.wa = mulT(M->plR[l], k.wa);
    const int jt = M->jtype[l];
    const double sg = jt < 0 ? -1.0 : 1.0;
    const int aj = jt < 0 ? -jt : jt;
    if (aj == 3) qa_joint<2>(C, Sn, sg, bseed, mu, k);
    else if (aj == 2) qa_joint<1>(C, Sn, sg, bseed, mu, k);
    else qa_joint<0>(C, Sn, sg, bseed, mu, k);
}
RD double dot_a(const T3<D1>& x, const T3<D1>& y) {  // the dual part of x . y
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) s += x.c[k].v * y.c[k].a + x.c[k].a * y.c[k].v;
    return s;
}
RD double qa_power_a(const DevModel* __restrict__ M, int l, const KinB<D1>& k) {  // dual part of w . (I b)
    const V3 c = v3(M->lever[l][0], M->lever[l][1], M->lever[l][2]);
    const double mass = M->mass[l];
    T3<D1> gl = k.bl + cross(k.ba, c);
#pragma unroll
    for (int i = 0; i < 3; i++) gl.c[i] = mass * gl.c[i];
    const T3<D1> fa = symmul(M->Ic[l], k.ba) + cross(c, gl);
    return dot_a(k.wl, gl) + dot_a(k.wa, fa);
}

// ---- d2 phi / d(q_d)_a d(q_d)_b: tau is quadratic in q_d, so the entry is the symmetric bilinear part of RNEA under the two
// unit velocity fields (alpha: unit q_d at joint a, beta: at joint b) -- plain doubles, nothing to differentiate:
//   c_l = Xup c_parent + crm(v_alpha) S [l == b] + crm(v_beta) S [l == a]          (the q_d q_d part of :157)
//   f_l = I c_l + v_alpha x* I v_beta + v_beta x* I v_alpha                         (:244-246)
struct KinV {
    T3<double> xa, xl, ya, yl, ca, cl, wa, wl;  // alpha field, beta field, mixed acceleration, virtual velocity
};
template <int K>
RD void vv_joint(double C, double Sn, double sa, double sb, double smu, KinV& k) {
    constexpr int I = (K + 1) % 3, J = (K + 2) % 3;
#define HROT(x) { const double xi_ = C * x.c[I] + Sn * x.c[J]; const double xj_ = C * x.c[J] - Sn * x.c[I]; x.c[I] = xi_; x.c[J] = xj_; }
    HROT(k.xa) HROT(k.xl) HROT(k.ya) HROT(k.yl) HROT(k.ca) HROT(k.cl) HROT(k.wa) HROT(k.wl)
#undef HROT
    k.xa.c[K] += sa;
    k.ya.c[K] += sb;
    k.ca.c[I] += k.xa.c[J] * sb + k.ya.c[J] * sa;
    k.ca.c[J] -= k.xa.c[I] * sb + k.ya.c[I] * sa;
    k.cl.c[I] += k.xl.c[J] * sb + k.yl.c[J] * sa;
    k.cl.c[J] -= k.xl.c[I] * sb + k.yl.c[I] * sa;
    k.wa.c[K] += smu;
}
RD void vv_link(const DevModel* __restrict__ M, int l, double C, double Sn, double ua, double ub, double mu, KinV& k) {
    const V3 p = v3(M->plp[l][0], M->plp[l][1], M->plp[l][2]);
#define HXT(a, lin) lin = mulT(M->plR[l], lin + cross(a, p)); a = mulT(M->plR[l], a);
    HXT(k.xa, k.xl) HXT(k.ya, k.yl) HXT(k.ca, k.cl) HXT(k.wa, k.wl)
#undef HXT
    const int jt = M->jtype[l];
    const double sg = jt < 0 ? -1.0 : 1.0;
    const int aj = jt < 0 ? -jt : jt;
    if (aj == 3) vv_joint<2>(C, Sn, sg * ua, sg * ub, sg * mu, k);
    else if (aj == 2) vv_joint<1>(C, Sn, sg * ua, sg * ub, sg * mu, k);
    else vv_joint<0>(C, Sn, sg * ua, sg * ub, sg * mu, k);
}
RD double dot3(const T3<double>& x, const T3<double>& y) { return x.c[0] * y.c[0] + x.c[1] * y.c[1] + x.c[2] * y.c[2]; }
RD double vv_power(const DevModel* __restrict__ M, int l, const KinV& k) {
    const V3 c = v3(M->lever[l][0], M->lever[l][1], M->lever[l][2]);
    const double mass = M->mass[l];
    T3<double> hx = k.xl + cross(k.xa, c), hy = k.yl + cross(k.ya, c), gl = k.cl + cross(k.ca, c);
#pragma unroll
    for (int i = 0; i < 3; i++) { hx.c[i] *= mass; hy.c[i] *= mass; gl.c[i] *= mass; }
    const T3<double> ax = symmul(M->Ic[l], k.xa) + cross(c, hx);  // angular momentum of each field
    const T3<double> ay = symmul(M->Ic[l], k.ya) + cross(c, hy);
    const T3<double> fl = gl + cross(k.xa, hy) + cross(k.ya, hx);
    const T3<double> fa = symmul(M->Ic[l], k.ca) + cross(c, gl) + cross(k.xa, ay) + cross(k.ya, ax) + cross(k.xl, hy) + cross(k.yl, hx);
    return dot3(k.wl, fl) + dot3(k.wa, fa);
}

// ---- d2 phi / dq_a d(q_d)_b: the q_d-directional derivative of RNEA is written out (variation fields d = delta v, e = delta a
// under a unit q_d at joint b; tau is quadratic in q_d), then differentiated once in q_a by first-order duals:
//   e_l = Xup e_parent + crm(d_l) S q_d_l + crm(v_l) S [l == b],   delta f = I e + d x* I v + v x* I d
template <class S>
struct KinQ {
    T3<S> va, vl, da, dl, ea, el, wa, wl;
};
template <int K, class S, class Tq>
RD void qv_joint(Tq C, Tq Sn, double vj, double dvj, double smu, KinQ<S>& k) {
    constexpr int I = (K + 1) % 3, J = (K + 2) % 3;
#define HROT(x) { const S xi_ = C * x.c[I] + Sn * x.c[J]; const S xj_ = C * x.c[J] - Sn * x.c[I]; x.c[I] = xi_; x.c[J] = xj_; }
    HROT(k.va) HROT(k.vl) HROT(k.da) HROT(k.dl) HROT(k.ea) HROT(k.el) HROT(k.wa) HROT(k.wl)
#undef HROT
    add_value(k.va.c[K], vj);
    add_value(k.da.c[K], dvj);
    k.ea.c[I] = k.ea.c[I] + k.da.c[J] * vj + k.va.c[J] * dvj;
    k.ea.c[J] = k.ea.c[J] - (k.da.c[I] * vj + k.va.c[I] * dvj);
    k.el.c[I] = k.el.c[I] + k.dl.c[J] * vj + k.vl.c[J] * dvj;
    k.el.c[J] = k.el.c[J] - (k.dl.c[I] * vj + k.vl.c[I] * dvj);
    add_value(k.wa.c[K], smu);
}
template <class S, class Tq>
RD void qv_link(const DevModel* __restrict__ M, int l, Tq C, Tq Sn, double qd, double ub, double mu, KinQ<S>& k) {
    const V3 p = v3(M->plp[l][0], M->plp[l][1], M->plp[l][2]);
#define HXT(a, lin) lin = mulT(M->plR[l], lin + cross(a, p)); a = mulT(M->plR[l], a);
    HXT(k.va, k.vl) HXT(k.da, k.dl) HXT(k.ea, k.el) HXT(k.wa, k.wl)
#undef HXT
    const int jt = M->jtype[l];
    const double sg = jt < 0 ? -1.0 : 1.0;
    const int aj = jt < 0 ? -jt : jt;
    if (aj == 3) qv_joint<2>(C, Sn, sg * qd, sg * ub, sg * mu, k);
    else if (aj == 2) qv_joint<1>(C, Sn, sg * qd, sg * ub, sg * mu, k);
    else qv_joint<0>(C, Sn, sg * qd, sg * ub, sg * mu, k);
}
RD double qv_power_a(const DevModel* __restrict__ M, int l, const KinQ<D1>& k) {  // dual part of w . delta f
    const V3 c = v3(M->lever[l][0], M->lever[l][1], M->lever[l][2]);
    const double mass = M->mass[l];
    T3<D1> hl = k.vl + cross(k.va, c), xl = k.dl + cross(k.da, c), gl = k.el + cross(k.ea, c);
#pragma unroll
    for (int i = 0; i < 3; i++) { hl.c[i] = mass * hl.c[i]; xl.c[i] = mass * xl.c[i]; gl.c[i] = mass * gl.c[i]; }
    const T3<D1> ha = symmul(M->Ic[l], k.va) + cross(c, hl);
    const T3<D1> xa = symmul(M->Ic[l], k.da) + cross(c, xl);
    const T3<D1> fl = gl + cross(k.da, hl) + cross(k.va, xl);
    const T3<D1> fa = symmul(M->Ic[l], k.ea) + cross(c, gl) + cross(k.da, ha) + cross(k.va, xa) + cross(k.dl, hl) + cross(k.vl, xl);
    return dot_a(k.wl, fl) + dot_a(k.wa, fa);
}

RD double wave_sum(double x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

// One sphere of one sample (lane): arg-min box, gradient n and Hessian K of the distance w.r.t. the centre, and the sphere's
// contribution  wgt (J_a^T K J_b + n . H_ab)  to every pair a <= b < joint id, added into acc (upper triangle, row by row).
// The joint id is wave-uniform (a wavefront owns one sphere): scalar branches.
RD void hess_sphere_terms(const double* lds, const int lane, const DevSpheres* __restrict__ S, const DevBox* __restrict__ boxes,
          const int O, const int sp, const double wgt, double (&acc)[H_PAIRS]) {
    constexpr int NV = 7;
    const int jid = S->joint_id[sp];
    const double* ro = lds + HL_RO + (S->ro_slot[jid - 1] * 12) * H_CH + lane;
    M3 R;
#pragma unroll
    for (int k = 0; k < 9; k++) R.m[k] = ro[k * H_CH];
    const V3 point = v3(ro[9 * H_CH], ro[10 * H_CH], ro[11 * H_CH]) +
                     mul(R, v3(S->offset[sp][0], S->offset[sp][1], S->offset[sp][2]));
    BoxBest best;
    best.key = 1e38; best.d2 = 0.0; best.sign = 1.0; best.dl = v3(0, 0, 0); best.idx = 0;
    for (int ob = 0; ob < O; ob++) box_candidate(boxes + ob, ob, point, best);
    V3 ng = v3(0, 0, 0);
    double K0 = 0, K1 = 0, K2 = 0, K3 = 0, K4 = 0, K5 = 0;
    if (best.d2 > 1e-12) {  // SQUARE_ROOT_THRESHOLD, BoxCollisionAvoidance.cpp:458
        const DevBox* bx = boxes + best.idx;
        M3 Rb;
#pragma unroll
        for (int k = 0; k < 9; k++) Rb.m[k] = bx->R[k];
        const double inv = 1.0 / sqrt(best.d2);
        const V3 nh = mul(Rb, best.dl) * inv;  // diff / |diff| in the world frame
        const double ws = wgt * best.sign;
        ng = nh * ws;
        // P = sum over the clamped axes (dl != 0 exactly there: :233-250, :269-273) of r_k r_k^T
        const double c0 = (best.dl.x != 0.0) ? 1.0 : 0.0, c1 = (best.dl.y != 0.0) ? 1.0 : 0.0, c2 = (best.dl.z != 0.0) ? 1.0 : 0.0;
        const V3 r0 = col(Rb, 0), r1 = col(Rb, 1), r2 = col(Rb, 2);
        const double kk = ws * inv;
        K0 = kk * (c0 * r0.x * r0.x + c1 * r1.x * r1.x + c2 * r2.x * r2.x - nh.x * nh.x);
        K1 = kk * (c0 * r0.x * r0.y + c1 * r1.x * r1.y + c2 * r2.x * r2.y - nh.x * nh.y);
        K2 = kk * (c0 * r0.x * r0.z + c1 * r1.x * r1.z + c2 * r2.x * r2.z - nh.x * nh.z);
        K3 = kk * (c0 * r0.y * r0.y + c1 * r1.y * r1.y + c2 * r2.y * r2.y - nh.y * nh.y);
        K4 = kk * (c0 * r0.y * r0.z + c1 * r1.y * r1.z + c2 * r2.y * r2.z - nh.y * nh.z);
        K5 = kk * (c0 * r0.z * r0.z + c1 * r1.z * r1.z + c2 * r2.z * r2.z - nh.z * nh.z);
    }
    // columns c < joint id move the sphere (jid is wave-uniform: scalar branches)
    V3 om[NV], Jc[NV];
#pragma unroll
    for (int c = 0; c < NV; c++) {
        if (c < jid) {
            const double* wv = lds + HL_W + (c * 6) * H_CH + lane;
            om[c] = v3(wv[0], wv[H_CH], wv[2 * H_CH]);
            Jc[c] = cross(om[c], point - v3(wv[3 * H_CH], wv[4 * H_CH], wv[5 * H_CH]));
        }
    }
#pragma unroll
    for (int cb = 0; cb < NV; cb++) {
        if (cb < jid) {
            const V3 KJ = symmul(K0, K1, K2, K3, K4, K5, Jc[cb]);
            const V3 nJ = cross(Jc[cb], ng);  // n . (w_a x J_b) = w_a . (J_b x n)
#pragma unroll
            for (int ca = 0; ca <= cb; ca++) acc[ca * 7 - (ca * (ca - 1)) / 2 + (cb - ca)] += dot(Jc[ca], KJ) + dot(om[ca], nJ);
        }
    }
}

// pairs dealt to the four wavefronts by cost (hyper-dual links (7 - b) dominate); entries = a * 8 + b
__device__ const unsigned char H_PAIR_OF[H_WAVES][7] = {
    {0 * 8 + 0, 1 * 8 + 3, 1 * 8 + 4, 0 * 8 + 5, 1 * 8 + 5, 2 * 8 + 6, 3 * 8 + 6},
    {0 * 8 + 1, 0 * 8 + 3, 3 * 8 + 3, 2 * 8 + 4, 4 * 8 + 5, 5 * 8 + 5, 6 * 8 + 6},
    {1 * 8 + 1, 2 * 8 + 2, 0 * 8 + 4, 3 * 8 + 4, 3 * 8 + 5, 0 * 8 + 6, 5 * 8 + 6},
    {0 * 8 + 2, 1 * 8 + 2, 2 * 8 + 3, 4 * 8 + 4, 2 * 8 + 5, 1 * 8 + 6, 4 * 8 + 6}};
RD int pair_index(int a, int b) { return a * 7 - (a * (a - 1)) / 2 + (b - a); }  // upper triangle row by row (Optimizer.cpp:531-538)

// SPLIT = false: one workgroup per decision vector (batches).  SPLIT = true: seven workgroups per decision vector
// (blockIdx.y = slot): each wavefront takes every 28th sphere, the 28 partial sums go to `Hout` = part[B][7][28],
// kinova_hess_torque_kernel<true> adds its share of the samples to the same slots and kinova_hessian_reduce_kernel adds the
// slots in a fixed order -- the single-problem (Ipopt-driven) case, where one workgroup per z would leave the chip idle.
#ifndef H_OCC
#define H_OCC 2  // waves per SIMD the register allocation must leave room for
#endif
template <bool SPLIT>
__global__ __launch_bounds__(RAPTOR_BLOCK, H_OCC) void kinova_hessian_kernel(
    const DevModel* __restrict__ M, const DevTraj* __restrict__ T, const double* __restrict__ basis,
    const double* __restrict__ wsum, const DevSpheres* __restrict__ S, const DevBox* __restrict__ boxes, const DevLayout L,
    const double* __restrict__ Z, const double* __restrict__ obj_factor, const double* __restrict__ lambda, const int tplan_n,
    double* __restrict__ Hout) {
    __shared__ double lds[HL_TOTAL];
    constexpr int NV = 7;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long b = blockIdx.x;
    const int N = T->N;
    const int ns = S->num_spheres;
    const int O = S->num_obstacles;
    const double* z = Z + b * NV;
    const double* lam = lambda + b * (long long)L.m;
    if (lane < 2 * H_PAIRS) lds[HL_ACC + wave * 2 * H_PAIRS + lane] = 0.0;
    double acc[H_PAIRS];  // obstacle rows: this lane's sample of every pass, this wavefront's spheres
#pragma unroll
    for (int k = 0; k < H_PAIRS; k++) acc[k] = 0.0;

#pragma unroll 1
    for (int i0 = 0; i0 < N; i0 += H_CH) {
        // ---- pass set-up: trajectory values, sincos, multipliers, basis weights of the 64 samples ----
        for (int idx = tid; idx < H_CH * NV; idx += RAPTOR_BLOCK) {
            const int s = idx & (H_CH - 1), j = idx >> 6;
            const int i = min(i0 + s, N - 1);
            double q, qd, qdd, sq, cq;
            traj_joint(T, basis, z, i, j, q, qd, qdd);
            sincos_joint(q, &sq, &cq);
            double* x = lds + HL_X + (j * 5) * H_CH + s;
            x[0] = qd; x[H_CH] = qdd; x[2 * H_CH] = cq; x[3 * H_CH] = sq; x[4 * H_CH] = q;
        }
        if (tid < H_CH) lds[HL_WS + tid] = wsum[min(i0 + tid, N - 1) * 3];  // pq_pz of the sample
        __syncthreads();
        const bool live = (i0 + lane) < N;
        const double w0 = lds[HL_WS + lane];

        // ---- world-frame joint axes / origins and sphere-link poses of every sample (first wavefront, lane = sample) ----
        if (wave == 0 && O > 0) {
            M3 Rw;
#pragma unroll
            for (int k = 0; k < 9; k++) Rw.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
            V3 ow = v3(0, 0, 0);
#pragma unroll 1
            for (int l = 0; l < NV; l++) {
                const double* x = lds + HL_X + (l * 5) * H_CH + lane;
                const M3 A = mul(Rw, M->plR[l]);
                V3 o = ow + mul(Rw, v3(M->plp[l][0], M->plp[l][1], M->plp[l][2]));
                M3 R;
                SV s;
                joint_world(M->jtype[l], x[4 * H_CH], x[2 * H_CH], x[3 * H_CH], A, R, o, s);
                Rw = R;
                ow = o;
                double* wv = lds + HL_W + (l * 6) * H_CH + lane;
                wv[0] = s.a.x; wv[H_CH] = s.a.y; wv[2 * H_CH] = s.a.z;
                wv[3 * H_CH] = ow.x; wv[4 * H_CH] = ow.y; wv[5 * H_CH] = ow.z;
                const int slot = S->ro_slot[l];
                if (slot >= 0) {
                    double* ro = lds + HL_RO + (slot * 12) * H_CH + lane;
#pragma unroll
                    for (int k = 0; k < 9; k++) ro[k * H_CH] = Rw.m[k];
                    ro[9 * H_CH] = ow.x; ro[10 * H_CH] = ow.y; ro[11 * H_CH] = ow.z;
                }
            }
        }

        __syncthreads();  // world-frame tables complete

        // ---- obstacle rows: a wavefront per sphere, lane = sample ----
        if (O > 0) {
#pragma unroll 1
            for (int sp = SPLIT ? wave + H_WAVES * (int)blockIdx.y : wave; sp < ns; sp += SPLIT ? 7 * H_WAVES : H_WAVES) {
                const double wgt = live ? lam[L.off_obs + (i0 + lane) * ns + sp] * w0 * w0 : 0.0;
                hess_sphere_terms(lds, lane, S, boxes, O, sp, wgt, acc);
            }
        }
        __syncthreads();  // the sample tables are rewritten by the next pass
    }
    if (O > 0) {  // the obstacle sums of this wavefront's spheres over all passes: one reduction over the 64 samples
#pragma unroll
        for (int k = 0; k < H_PAIRS; k++) {
            const double tot = wave_sum(acc[k]);
            if (lane == 0) lds[HL_ACC + wave * 2 * H_PAIRS + k] = tot;
        }
    }
    __syncthreads();

    // ---- the partial sums in a fixed order; hess_f = diag(pq_pz(tplan_n)^2) (KinovaOptimizer.cpp:243-247) ----
    if (tid < H_PAIRS) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < H_WAVES; w++) s += lds[HL_ACC + w * 2 * H_PAIRS + tid] + lds[HL_ACC + w * 2 * H_PAIRS + H_PAIRS + tid];
        int pa = 0, rem = tid;
        while (rem >= NV - pa) {
            rem -= NV - pa;
            pa++;
        }
        if (SPLIT) {
            Hout[(b * 7 + blockIdx.y) * H_PAIRS + tid] = s;
        } else {
            if (obj_factor != nullptr && rem == 0) {
                const double w = wsum[tplan_n * 3];
                s += obj_factor[b] * (w * w);
            }
            Hout[b * H_PAIRS + tid] = s;
        }
    }
}

// second step of the split form: H[b][p] = sum over the seven slots (fixed order) + obj_factor * hess_f
__global__ __launch_bounds__(64) void kinova_hessian_reduce_kernel(const double* __restrict__ part, const double* __restrict__ wsum,
                                                                    const double* __restrict__ obj_factor, const int tplan_n,
                                                                    const int B, double* __restrict__ Hout) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= B * H_PAIRS) return;
    const int b = idx / H_PAIRS, p = idx - b * H_PAIRS;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; k++) s += part[((long long)b * 7 + k) * H_PAIRS + p];
    int pa = 0, rem = p;
    while (rem >= 7 - pa) {
        rem -= 7 - pa;
        pa++;
    }
    if (obj_factor != nullptr && rem == 0) {
        const double w = wsum[tplan_n * 3];
        s += obj_factor[b] * (w * w);
    }
    Hout[idx] = s;
}

// KinovaOptimizer::eval_f / eval_grad_f (KinovaOptimizer.cpp:152-218): one lane per decision vector.
__global__ __launch_bounds__(RAPTOR_BLOCK) void kinova_objective_kernel(const DevTraj* __restrict__ T, const double* __restrict__ basis,
                                                                         const double* __restrict__ wsum, const double* __restrict__ Z,
                                                                         const double* __restrict__ q_des, const int tplan_n, const int B,
                                                                         double* __restrict__ F, double* __restrict__ GRAD) {
    const int b = blockIdx.x * RAPTOR_BLOCK + threadIdx.x;
    if (b >= B) return;
    constexpr int NV = 7;
    const double* z = Z + (long long)b * NV;
    double d[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        double q, qd, qdd;
        traj_joint(T, basis, z, tplan_n, j, q, qd, qdd);
        double r = q - q_des[j];
        if (j % 2 == 0) r = wrap_to_pi(r);  // continuous joints (Utils.h:59-68)
        d[j] = r;
    }
    // :169-175 sums the continuous joints first
    double obj = d[0] * d[0];
    obj += d[2] * d[2];
    obj += d[4] * d[4];
    obj += d[6] * d[6];
    obj += d[1] * d[1];
    obj += d[3] * d[3];
    obj += d[5] * d[5];
    F[b] = 0.5 * obj;
    if (GRAD != nullptr) {
        const double w = wsum[tplan_n * 3];
#pragma unroll
        for (int j = 0; j < NV; j++) GRAD[(long long)b * NV + j] = d[j] * w;
    }
}

#endif
