// =====================================================================================
// TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of the reference's hot path:
//     z -> Trajectories::compute -> InverseDynamics::compute -> Constraints::compute
//       -> Optimizer::eval_g / eval_jac_g packing.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
// and only as the checker / reported baseline.  The product (raptor_amd/, libraptor_hip.so)
// never links, imports or calls it.
//
// Every function cites the reference file:line it follows (paths relative to the reference
// checkout).  The reference's third-party arithmetic that is NOT in the checkout --
// pinocchio::rnea / computeRNEADerivatives (Pinocchio >= 3, unpinned) -- is restated through the
// reference's own dependency-free Featherstone recursion (CustomizedInverseDynamics.cpp), which
// the reference's tests pin to pinocchio::rnea at 1e-10
// (Tests/KinematicsDynamics/TestCustomizedInverseDynamics.cpp:49-53); its derivatives are the
// forward-mode sweep of the same file seeded with identity (what computeRNEADerivatives returns
// mathematically).  Pinning: the nine TaperedCapsule known answers of the reference's tests plus the
// reference's finite-difference recipes (tests/test_oracle_*.py); for d(tau)/d(q,v,a), model
// building and anything else without a golden vector in the reference: PARITY UNPINNED -- except the Digit path, which the
// reference's one shipped data file (Examples/Digit/data/initial-digit-two-step-5.txt, a converged two-step gait) pins at the
// reference solver's tolerance: every equality row of both steps and both links <= 5e-5 (tests/test_reference_solution_pin.py).
// =====================================================================================
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "../include/raptor_hip.h"
#include "oracle_math.h"

namespace oracle {

static thread_local std::string g_err;

// ------------------------------------------------------------------ Utils (Utils/include/Utils.h)
// Utils.h:85-95
static inline double sign(double val, double eps = 1e-8) {
    if (val > eps) return 1.0;
    if (val < -eps) return -1.0;
    return 0.0;
}
// Utils.h:59-68
static inline double wrapToPi(double angle) {
    double res = angle;
    // the reference's two loops, bounded like the kernels' (legged_kernel.h wrap_to_pi) so that the checker terminates on the
    // angles an unclosable z produces: beyond 64 turns one remainder step comes first, a non-finite angle (the reference would
    // never return) is handed back as it is
    if (!(std::fabs(res) <= 128.0 * M_PI)) {
        if (!std::isfinite(res)) return res;
        res = std::remainder(res, 2.0 * M_PI);
    }
    while (res > M_PI) res -= 2.0 * M_PI;
    while (res < -M_PI) res += 2.0 * M_PI;
    return res;
}
// Utils.h:185-191   plux(R,p) = [R 0; -R*skew(p) R]
static Mat6 plux(const Mat3& R, const Vec3& p) {
    Mat6 X;
    Mat3 mRS = (R * skew(p)) * -1.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            X(i, j) = R(i, j);
            X(i + 3, j + 3) = R(i, j);
            X(i + 3, j) = mRS(i, j);
        }
    return X;
}

// ------------------------------------------------------------------ model
struct Model {
    int nv = 0;
    std::vector<int> parent, jtype;
    std::vector<Mat3> plR;
    std::vector<Vec3> plp;
    std::vector<double> mass;
    std::vector<Vec3> lever;
    std::vector<Mat3> inertia;
    Vec3 gravity;
    std::vector<double> armature, damping, friction;
    // CustomizedInverseDynamics.cpp:43-59
    std::vector<Mat6> Xtree, I;
    Vec6 a_grav;
};

static void model_from_desc(const raptor_model_desc* d, Model& m) {
    if (d == nullptr || d->nv <= 0) throw std::invalid_argument("model: nv must be positive");
    m.nv = d->nv;
    m.parent.assign(d->parent, d->parent + d->nv);
    m.jtype.assign(d->jtype, d->jtype + d->nv);
    m.plR.resize(m.nv);
    m.plp.resize(m.nv);
    m.mass.resize(m.nv);
    m.lever.resize(m.nv);
    m.inertia.resize(m.nv);
    m.armature.assign(m.nv, 0.0);
    m.damping.assign(m.nv, 0.0);
    m.friction.assign(m.nv, 0.0);
    for (int i = 0; i < m.nv; i++) {
        if (m.parent[i] >= i || m.parent[i] < -1) throw std::invalid_argument("model: parents must precede children");
        if (m.jtype[i] == 0 || std::abs(m.jtype[i]) > 6) throw std::invalid_argument("model: jtype must be +-1..6");
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) {
                m.plR[i](r, c) = d->placement_R[i * 9 + r * 3 + c];
                m.inertia[i](r, c) = d->inertia[i * 9 + r * 3 + c];
            }
            m.plp[i](r) = d->placement_p[i * 3 + r];
            m.lever[i](r) = d->lever[i * 3 + r];
        }
        m.mass[i] = d->mass[i];
        if (d->armature) m.armature[i] = d->armature[i];
        if (d->damping) m.damping[i] = d->damping[i];
        if (d->friction) m.friction[i] = d->friction[i];
    }
    m.gravity = Vec3(d->gravity[0], d->gravity[1], d->gravity[2]);
    // CustomizedInverseDynamics.cpp:43-59
    m.Xtree.resize(m.nv);
    m.I.resize(m.nv);
    for (int i = 0; i < m.nv; i++) {
        m.Xtree[i] = plux(m.plR[i].transpose(), m.plp[i]);
        const Mat3 CC = skew(m.lever[i]);
        const double mm = m.mass[i];
        const Mat3 A = (CC * CC.transpose()) * mm + m.inertia[i];
        const Mat3 Bm = CC * mm;
        const Mat3 Ct = CC.transpose() * mm;
        Mat6 I6;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                I6(r, c) = A(r, c);
                I6(r, c + 3) = Bm(r, c);
                I6(r + 3, c) = Ct(r, c);
                I6(r + 3, c + 3) = (r == c) ? mm : 0.0;
            }
        m.I[i] = I6;
    }
    for (int k = 0; k < 3; k++) {
        m.a_grav(k) = 0.0;  // model.gravity.angular()
        m.a_grav(k + 3) = m.gravity(k);
    }
}

// ------------------------------------------------------------------ Spatial.h
// KinematicsDynamics/include/Spatial.h:115-258 (jcalc with derivative). jtype -6 quirk kept (:242-248).
static void jcalc(Mat6& Xj, Mat6& dXjdq, Vec6& S, int jtyp, double q) {
    Xj = Mat6::Identity();
    dXjdq = Mat6();
    S = Vec6();
    double c = std::cos(q), s = std::sin(q);
    double dcdq = -s, dsdq = c;
    if (jtyp < 0) {
        s = -s;
        dsdq = -dsdq;
    }
    switch (jtyp) {
        case 1:
        case -1:
            Xj(1, 1) = c; Xj(1, 2) = s; Xj(2, 1) = -s; Xj(2, 2) = c;
            dXjdq(1, 1) = dcdq; dXjdq(1, 2) = dsdq; dXjdq(2, 1) = -dsdq; dXjdq(2, 2) = dcdq;
            S(0) = (jtyp > 0) ? 1 : -1;
            break;
        case 2:
        case -2:
            Xj(0, 0) = c; Xj(0, 2) = -s; Xj(2, 0) = s; Xj(2, 2) = c;
            dXjdq(0, 0) = dcdq; dXjdq(0, 2) = -dsdq; dXjdq(2, 0) = dsdq; dXjdq(2, 2) = dcdq;
            S(1) = (jtyp > 0) ? 1 : -1;
            break;
        case 3:
        case -3:
            Xj(0, 0) = c; Xj(0, 1) = s; Xj(1, 0) = -s; Xj(1, 1) = c;
            dXjdq(0, 0) = dcdq; dXjdq(0, 1) = dsdq; dXjdq(1, 0) = -dsdq; dXjdq(1, 1) = dcdq;
            S(2) = (jtyp > 0) ? 1 : -1;
            break;
        case 4:
            Xj(4, 2) = q; Xj(5, 1) = -q; dXjdq(4, 2) = 1; dXjdq(5, 1) = -1; S(3) = 1;
            break;
        case 5:
            Xj(3, 2) = -q; Xj(5, 0) = q; dXjdq(3, 2) = -1; dXjdq(5, 0) = 1; S(4) = 1;
            break;
        case 6:
            Xj(3, 1) = q; Xj(4, 0) = -q; dXjdq(3, 1) = 1; dXjdq(4, 0) = -1; S(5) = 1;
            break;
        case -4:
            Xj(4, 2) = -q; Xj(5, 1) = q; dXjdq(4, 2) = -1; dXjdq(5, 1) = 1; S(3) = -1;
            break;
        case -5:
            Xj(3, 2) = q; Xj(5, 0) = -q; dXjdq(3, 2) = 1; dXjdq(5, 0) = -1; S(4) = -1;
            break;
        case -6:  // reference writes the Px entries here (Spatial.h:242-248); reproduced, unused by the configs
            Xj(4, 2) = q; Xj(5, 1) = -q; dXjdq(4, 2) = 1; dXjdq(5, 1) = -1; S(5) = -1;
            break;
        default:
            throw std::invalid_argument("spatial.hpp: jcalc(): unknown joint type!");
    }
    if (std::abs(jtyp) <= 3) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                Xj(i + 3, j + 3) = Xj(i, j);
                dXjdq(i + 3, j + 3) = dXjdq(i, j);
            }
    }
}

// Spatial.h:406-434
static Mat6 crm(const Vec6& v) {
    Mat6 vc;
    vc(0, 1) = -v(2); vc(0, 2) = v(1); vc(1, 0) = v(2); vc(1, 2) = -v(0); vc(2, 0) = -v(1); vc(2, 1) = v(0);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) vc(i + 3, j + 3) = vc(i, j);
    vc(3, 1) = -v(5); vc(3, 2) = v(4); vc(4, 0) = v(5); vc(4, 2) = -v(3); vc(5, 0) = -v(4); vc(5, 1) = v(3);
    return vc;
}
// Spatial.h:436-440
static Mat6 crf(const Vec6& v) {
    Mat6 t = crm(v).transpose();
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) t(i, j) = -t(i, j);
    return t;
}

static inline void row_axpy(MatX& dst, int rd, double alpha, const MatX& src, int rs) {
    if (alpha == 0.0) return;
    for (int j = 0; j < dst.c; j++) dst(rd, j) += alpha * src(rs, j);
}

// ------------------------------------------------------------------ RNEA (+ forward-mode derivative)
// Restates KinematicsDynamics/src/CustomizedInverseDynamics.cpp:133-302 for a model whose joints are
// all active.  pq_pz/pq_d_pz/pq_dd_pz are nv x W seed matrices (W == 0: values only).
// Output: tau[nv] WITHOUT damping/friction, but WITH armature*q_dd (what pinocchio::rnea (>= 3) returns,
// SURVEY.md section 8a row a4), ptau[nv x W], link wrenches f[nv] (for the reference's identity test).
static void featherstone_rnea(const Model& m, const double* q, const double* q_d, const double* q_dd,
                              const MatX* pq_pz, const MatX* pq_d_pz, const MatX* pq_dd_pz,
                              double* tau, MatX* ptau_pz, std::vector<Vec6>* f_out) {
    const int nv = m.nv;
    const bool deriv = (pq_pz != nullptr);
    const int W = deriv ? pq_pz->c : 0;
    std::vector<Mat6> Xup(nv), dXupdq(nv);
    std::vector<Vec6> S(nv), v(nv), a(nv), f(nv);
    std::vector<MatX> pv(nv), pa(nv), pf(nv);
    if (deriv) *ptau_pz = MatX::Zero(nv, W);

    Vec6 neg_a_grav = -m.a_grav;

    for (int j = 0; j < nv; j++) {
        const int parent_id = m.parent[j];
        Mat6 XJ, dXJdq;
        jcalc(XJ, dXJdq, S[j], m.jtype[j], q[j]);  // :140-145
        Vec6 vJ = S[j] * q_d[j];                    // :147
        Xup[j] = XJ * m.Xtree[j];                   // :148
        if (deriv) dXupdq[j] = dXJdq * m.Xtree[j];  // :151

        if (parent_id > -1) {
            v[j] = Xup[j] * v[parent_id] + vJ;  // :155
            Mat6 crm_v_j = crm(v[j]);
            a[j] = Xup[j] * a[parent_id] + crm_v_j * vJ + S[j] * q_dd[j];  // :157
            if (deriv) {
                pv[j] = Xup[j] * pv[parent_id];  // :161
                for (int k = 0; k < 6; k++)
                    if (S[j](k) != 0) row_axpy(pv[j], k, S[j](k), *pq_d_pz, j);  // :165-169
                Vec6 dXv = dXupdq[j] * v[parent_id];                                // :172
                for (int k = 0; k < 6; k++) row_axpy(pv[j], k, dXv(k), *pq_pz, j);

                pa[j] = Xup[j] * pa[parent_id];  // :179
                // :182-187   d(crm(v_j)) * vJ
                MatX& A = pa[j];
                const MatX& P = pv[j];
                for (int c = 0; c < W; c++) {
                    A(0, c) += vJ(2) * P(1, c) - vJ(1) * P(2, c);
                    A(1, c) += vJ(0) * P(2, c) - vJ(2) * P(0, c);
                    A(2, c) += vJ(1) * P(0, c) - vJ(0) * P(1, c);
                    A(3, c) += vJ(2) * P(4, c) - vJ(1) * P(5, c) - vJ(4) * P(2, c) + vJ(5) * P(1, c);
                    A(4, c) += vJ(0) * P(5, c) - vJ(2) * P(3, c) + vJ(3) * P(2, c) - vJ(5) * P(0, c);
                    A(5, c) += vJ(1) * P(3, c) - vJ(0) * P(4, c) - vJ(3) * P(1, c) + vJ(4) * P(0, c);
                }
                Vec6 crm_S = crm_v_j * S[j];  // :190-193
                for (int k = 0; k < 6; k++) row_axpy(pa[j], k, crm_S(k), *pq_d_pz, j);
                for (int k = 0; k < 6; k++)
                    if (S[j](k) != 0) row_axpy(pa[j], k, S[j](k), *pq_dd_pz, j);  // :196-200
                Vec6 dXa = dXupdq[j] * a[parent_id];                                // :203-206
                for (int k = 0; k < 6; k++) row_axpy(pa[j], k, dXa(k), *pq_pz, j);
            }
        } else {
            v[j] = vJ;                                      // :211
            a[j] = Xup[j] * neg_a_grav + S[j] * q_dd[j];  // :212
            if (deriv) {
                pv[j] = MatX::Zero(6, W);
                for (int k = 0; k < 6; k++)
                    if (S[j](k) != 0) row_axpy(pv[j], k, S[j](k), *pq_d_pz, j);  // :219-223
                pa[j] = MatX::Zero(6, W);
                for (int k = 0; k < 6; k++)
                    if (S[j](k) != 0) row_axpy(pa[j], k, S[j](k), *pq_dd_pz, j);  // :230-234
                Vec6 dXg = dXupdq[j] * neg_a_grav;                                  // :236-239
                for (int k = 0; k < 6; k++) row_axpy(pa[j], k, dXg(k), *pq_pz, j);
            }
        }

        Mat6 crf_v_j = crf(v[j]);  // :244-246
        Vec6 Iv = m.I[j] * v[j];
        f[j] = m.I[j] * a[j] + crf_v_j * Iv;
        if (deriv) {
            pf[j] = m.I[j] * pa[j] + (crf_v_j * m.I[j]) * pv[j];  // :250
            MatX& F = pf[j];
            const MatX& P = pv[j];
            for (int c = 0; c < W; c++) {  // :253-258
                F(0, c) += Iv(2) * P(1, c) - Iv(1) * P(2, c) - Iv(4) * P(5, c) + Iv(5) * P(4, c);
                F(1, c) += Iv(0) * P(2, c) - Iv(2) * P(0, c) + Iv(3) * P(5, c) - Iv(5) * P(3, c);
                F(2, c) += Iv(1) * P(0, c) - Iv(0) * P(1, c) - Iv(3) * P(4, c) + Iv(4) * P(3, c);
                F(3, c) += Iv(5) * P(1, c) - Iv(4) * P(2, c);
                F(4, c) += Iv(3) * P(2, c) - Iv(5) * P(0, c);
                F(5, c) += Iv(4) * P(0, c) - Iv(3) * P(1, c);
            }
        }
    }

    if (f_out) {
        // link-local wrench before children are accumulated is not what pinocchio::rnea stores in data.f;
        // data.f[i] is the accumulated one, as is the reference's f(j) after its backward pass.
    }

    for (int j = nv - 1; j >= 0; j--) {  // :263-302
        const int parent_id = m.parent[j];
        tau[j] = S[j].dot(f[j]) + m.armature[j] * q_dd[j];
        if (deriv) {
            for (int c = 0; c < W; c++) {
                double s = 0;
                for (int k = 0; k < 6; k++) s += S[j](k) * pf[j](k, c);
                (*ptau_pz)(j, c) = s + m.armature[j] * (*pq_dd_pz)(j, c);
            }
        }
        if (parent_id > -1) {
            Mat6 XT = Xup[j].transpose();
            f[parent_id] = f[parent_id] + XT * f[j];  // :288
            if (deriv) {
                MatX add = XT * pf[j];  // :291
                for (size_t t = 0; t < add.a.size(); t++) pf[parent_id].a[t] += add.a[t];
                Vec6 dXTf = dXupdq[j].transpose() * f[j];  // :295-298
                for (int k = 0; k < 6; k++) row_axpy(pf[parent_id], k, dXTf(k), *pq_pz, j);
            }
        }
    }
    if (f_out) *f_out = f;
}

// What pinocchio::computeRNEADerivatives hands back (InverseDynamics.cpp:106-108): identity seeds.
static void rnea_derivatives(const Model& m, const double* q, const double* v, const double* a,
                             double* tau, MatX& dq, MatX& dv, MatX& da) {
    const int nv = m.nv;
    MatX sq(nv, 3 * nv), sv(nv, 3 * nv), sa(nv, 3 * nv);
    for (int i = 0; i < nv; i++) {
        sq(i, i) = 1.0;
        sv(i, nv + i) = 1.0;
        sa(i, 2 * nv + i) = 1.0;
    }
    MatX pt;
    featherstone_rnea(m, q, v, a, &sq, &sv, &sa, tau, &pt, nullptr);
    dq = MatX(nv, nv);
    dv = MatX(nv, nv);
    da = MatX(nv, nv);
    for (int i = 0; i < nv; i++)
        for (int j = 0; j < nv; j++) {
            dq(i, j) = pt(i, j);
            dv(i, j) = pt(i, nv + j);
            da(i, j) = pt(i, 2 * nv + j);
        }
}

// ------------------------------------------------------------------ Trajectories
struct Trajectory {
    int type = 0, degree = 0, N = 0, Nact = 0, varLength = 0;
    double T = 0;
    std::vector<double> tspan;
    VecX q0, q_d0, q_dd0, qT;
    // PiecewiseBezierCurves.h:36-54
    bool optimize_begin_position = false, optimize_end_position = false;
    int begin_position_offset = 0, end_position_offset = 0;
    int fixed_rows = 0;  // BezierCurves: 1 after constrainInitialPosition, 2 after constrainInitialVelocity
    MatX coefficients;  // (degree+1) x Nact
    VecX Bionomials;
    // results (Trajectories.h:71-83)
    std::vector<VecX> q, q_d, q_dd;
    std::vector<MatX> pq_pz, pq_d_pz, pq_dd_pz;

    void init(const raptor_traj_desc* d) {
        if (d == nullptr) throw std::invalid_argument("traj: null descriptor");
        type = d->type;
        T = d->T;
        N = d->N;
        Nact = d->Nact;
        if (N < 1 || Nact < 1) throw std::invalid_argument("traj: N and Nact must be positive");
        degree = (type == RAPTOR_TRAJ_ARMOUR_BEZIER) ? 5 : d->degree;  // ARMOUR_BEZIER_CURVE_DEGREE
        if (degree < 1) throw std::invalid_argument("traj: degree must be positive");
        // Trajectories.cpp:54-64
        tspan.assign(N, 0.0);
        if (type == RAPTOR_TRAJ_PIECEWISE_BEZIER) {  // PiecewiseBezierCurves.cpp:18-22: always LinSpaced(N, 0, T)
            if (N % (degree + 1) != 0) throw std::invalid_argument("PiecewiseBezierCurves: N must be a multiple of (degree + 1)");
            for (int i = 0; i < N; i++) tspan[i] = (N == 1) ? T : (0.0 + i * ((T - 0.0) / (N - 1)));
            if (N > 1) tspan[N - 1] = T;
        } else if (d->time_discretization == RAPTOR_TIME_UNIFORM) {
            // Eigen LinSpaced(N, 0, T): low + i*step, last element forced to high
            for (int i = 0; i < N; i++) tspan[i] = (N == 1) ? T : (0.0 + i * ((T - 0.0) / (N - 1)));
            if (N > 1) tspan[N - 1] = T;
        } else if (d->time_discretization == RAPTOR_TIME_CHEBYSHEV) {
            for (int i = 1; i < N - 1; i++) tspan[i] = 0.5 * T * (1 + std::cos(M_PI * (2 * (N - i) - 3) / (2 * (N - 2))));
            tspan[0] = 0;
            tspan[N - 1] = T;
        } else {
            throw std::invalid_argument("traj: unknown time discretization");
        }
        if (type == RAPTOR_TRAJ_ARMOUR_BEZIER) {
            if (!d->q0 || !d->q_d0 || !d->q_dd0) throw std::invalid_argument("ArmourBezierCurves: q0.size() != Nact");
            q0.assign(d->q0, d->q0 + Nact);
            q_d0.assign(d->q_d0, d->q_d0 + Nact);
            q_dd0.assign(d->q_dd0, d->q_dd0 + Nact);
            varLength = Nact;  // ArmourBezierCurves.cpp:61
            coefficients = MatX(degree + 1, Nact);
            for (int j = 0; j < Nact; j++) {  // :63-69
                coefficients(0, j) = q0[j];
                coefficients(1, j) = q0[j] + (T * q_d0[j]) / 5.0;
                coefficients(2, j) = q0[j] + (2.0 * T * q_d0[j]) / 5.0 + (T * T * q_dd0[j]) / 20.0;
            }
        } else if (type == RAPTOR_TRAJ_PIECEWISE_BEZIER) {  // PiecewiseBezierCurves.cpp:34-47
            varLength = Nact * degree * 3;
            if (d->q0) q0.assign(d->q0, d->q0 + Nact);
            if (d->qT) qT.assign(d->qT, d->qT + Nact);
            if ((int)q0.size() != Nact) {
                optimize_begin_position = true;
                begin_position_offset = varLength;
                varLength += Nact;
            }
            if ((int)qT.size() != Nact) {
                optimize_end_position = true;
                end_position_offset = varLength;
                varLength += Nact;
            }
            coefficients = MatX(6, Nact);
        } else if (type == RAPTOR_TRAJ_BEZIER || type == RAPTOR_TRAJ_POLYNOMIAL) {
            varLength = (degree + 1) * Nact;
            coefficients = MatX(degree + 1, Nact);
            if (type == RAPTOR_TRAJ_BEZIER) {
                // BezierCurves::constrainInitialPosition / constrainInitialVelocity (BezierCurves.cpp:39-69), as the single-step
                // optimizers call them when q0_input / q_d0_input are given (TalosSingleStepOptimizer.cpp:41-46)
                if (d->q_d0 && !d->q0) throw std::invalid_argument("BezierCurves: initial position must be constrained first");
                if (d->q0) {
                    q0.assign(d->q0, d->q0 + Nact);
                    fixed_rows = 1;
                    varLength -= Nact;
                }
                if (d->q_d0) {
                    q_d0.assign(d->q_d0, d->q_d0 + Nact);
                    fixed_rows = 2;
                    varLength -= Nact;
                }
                if (degree + 1 - fixed_rows < 1) throw std::invalid_argument("traj: degree too low for the constrained initial state");
            }
        } else {
            throw std::invalid_argument("traj: unknown trajectory type");
        }
        if (type == RAPTOR_TRAJ_PIECEWISE_BEZIER) {  // PiecewiseBezierCurves.cpp:28-32: quintic pieces
            Bionomials.assign(6, 1.0);
            for (int j = 1; j <= 2; j++) {
                Bionomials[j] = Bionomials[j - 1] * (6 - j) / j;
                Bionomials[5 - j] = Bionomials[j];
            }
        } else if (type != RAPTOR_TRAJ_POLYNOMIAL) {
            // BezierCurves.cpp:32-36
            Bionomials.assign(degree + 1, 1.0);
            for (int j = 1; j <= degree / 2; j++) {
                Bionomials[j] = Bionomials[j - 1] * (degree + 1 - j) / j;
                Bionomials[degree - j] = Bionomials[j];
            }
        }
        q.assign(N, VecX(Nact, 0.0));
        q_d = q;
        q_dd = q;
        pq_pz.assign(N, MatX(Nact, varLength));
        pq_d_pz = pq_pz;
        pq_dd_pz = pq_pz;
    }

    // basis values at sample x
    void basis(int x, VecX& B, VecX& dB, VecX& ddB) const {
        B.assign(degree + 1, 0.0);
        dB = B;
        ddB = B;
        if (type == RAPTOR_TRAJ_POLYNOMIAL) {
            // Polynomials.cpp:42-68
            const double t = tspan[x];
            B[0] = 1; dB[0] = 0; ddB[0] = 0;
            B[1] = t; dB[1] = 1; ddB[1] = 0;
            double powt = 1;
            for (int i = 2; i <= degree; i++) {
                ddB[i] = powt;
                dB[i] = t * ddB[i] / (i - 1);
                B[i] = t * dB[i] / i;
                powt *= t;
            }
            return;
        }
        // BezierCurves.cpp:103-141 == ArmourBezierCurves.cpp:102-140
        const double t = tspan[x] / T;
        VecX tA(degree + 1, 1.0), tB(degree + 1, 1.0), dtA(degree + 1, 0.0), dtB(degree + 1, 0.0),
            ddtA(degree + 1, 0.0), ddtB(degree + 1, 0.0);
        for (int j = 1; j <= degree; j++) {
            tA[j] = t * tA[j - 1];
            tB[degree - j] = (1 - t) * tB[degree - j + 1];
            dtA[j] = j * tA[j - 1];
            dtB[degree - j] = -j * tB[degree - j + 1];
            ddtA[j] = j * dtA[j - 1];
            ddtB[degree - j] = -j * dtB[degree - j + 1];
        }
        for (int k = 0; k <= degree; k++) {
            B[k] = Bionomials[k] * tA[k] * tB[k];
            dB[k] = Bionomials[k] * (dtA[k] * tB[k] + tA[k] * dtB[k]) / T;
            ddB[k] = Bionomials[k] * (ddtA[k] * tB[k] + 2 * dtA[k] * dtB[k] + tA[k] * ddtB[k]) / (T * T);
        }
    }

    // PiecewiseBezierCurves::compute (PiecewiseBezierCurves.cpp:52-255)
    void compute_piecewise(const double* z, bool compute_derivatives) {
        const int seg_size = N / (degree + 1);
        const double localT = T / (degree + 1);
        for (int x = 0; x < N; x++) {
            const int id = x / seg_size;                              // :68
            const double t = (tspan[x] - id * localT) / localT;       // :71
            const double velocity_factor = localT / 5.0;              // :74-75
            const double acceleration_factor = localT * localT / 20.0;
            VecX begin_q(Nact, 0.0), begin_q_d(Nact, 0.0), begin_q_dd(Nact, 0.0), end_q(Nact, 0.0), end_q_d(Nact, 0.0), end_q_dd(Nact, 0.0);
            for (int j = 0; j < Nact; j++) {                          // :81-123
                if (id == 0) {
                    begin_q[j] = optimize_begin_position ? z[begin_position_offset + j] : q0[j];
                } else {
                    const double* cp = z + (id - 1) * 3 * Nact;
                    begin_q[j] = cp[j]; begin_q_d[j] = cp[Nact + j]; begin_q_dd[j] = cp[2 * Nact + j];
                }
                if (id == degree) {
                    end_q[j] = optimize_end_position ? z[end_position_offset + j] : qT[j];
                } else {
                    const double* cp = z + id * 3 * Nact;
                    end_q[j] = cp[j]; end_q_d[j] = cp[Nact + j]; end_q_dd[j] = cp[2 * Nact + j];
                }
                coefficients(0, j) = begin_q[j];                       // :127-138
                coefficients(1, j) = begin_q[j] + begin_q_d[j] * velocity_factor;
                coefficients(2, j) = begin_q[j] + begin_q_d[j] * 2.0 * velocity_factor + begin_q_dd[j] * acceleration_factor;
                coefficients(3, j) = end_q[j] - end_q_d[j] * 2.0 * velocity_factor + end_q_dd[j] * acceleration_factor;
                coefficients(4, j) = end_q[j] - end_q_d[j] * velocity_factor;
                coefficients(5, j) = end_q[j];
            }
            double tA[6], tB[6], dtA[6], dtB[6], ddtA[6], ddtB[6], B[6], dB[6], ddB[6];  // :143-167
            for (int k = 0; k < 6; k++) { tA[k] = 1; tB[k] = 1; dtA[k] = 0; dtB[k] = 0; ddtA[k] = 0; ddtB[k] = 0; }
            for (int j = 1; j <= 5; j++) {
                tA[j] = t * tA[j - 1];
                tB[5 - j] = (1 - t) * tB[5 - j + 1];
                dtA[j] = j * tA[j - 1];
                dtB[5 - j] = -j * tB[5 - j + 1];
                ddtA[j] = j * dtA[j - 1];
                ddtB[5 - j] = -j * dtB[5 - j + 1];
            }
            for (int k = 0; k < 6; k++) {
                B[k] = Bionomials[k] * tA[k] * tB[k];
                dB[k] = Bionomials[k] * (dtA[k] * tB[k] + tA[k] * dtB[k]) / localT;
                ddB[k] = Bionomials[k] * (ddtA[k] * tB[k] + 2 * dtA[k] * dtB[k] + tA[k] * ddtB[k]) / (localT * localT);
            }
            for (int j = 0; j < Nact; j++) {  // :175-177
                double s0 = 0, s1 = 0, s2 = 0;
                for (int k = 0; k < 6; k++) {
                    s0 += coefficients(k, j) * B[k];
                    s1 += coefficients(k, j) * dB[k];
                    s2 += coefficients(k, j) * ddB[k];
                }
                q[x][j] = s0; q_d[x][j] = s1; q_dd[x][j] = s2;
            }
            if (!compute_derivatives) continue;
            pq_pz[x].setZero(); pq_d_pz[x].setZero(); pq_dd_pz[x].setZero();
            MatX* P[3] = {&pq_pz[x], &pq_d_pz[x], &pq_dd_pz[x]};
            const double* Bs[3] = {B, dB, ddB};
            for (int o = 0; o < 3; o++) {
                const double* b = Bs[o];
                MatX& M = *P[o];
                for (int i = 0; i < Nact; i++) {
                    if (id < degree) {  // columns of the control point that ends this piece (:186-194, :234-242)
                        const int c0 = id * 3 * Nact + i;
                        M(i, c0) = b[3] + b[4] + b[5];
                        M(i, c0 + Nact) = -b[3] * 2.0 * velocity_factor - b[4] * velocity_factor;
                        M(i, c0 + 2 * Nact) = b[3] * acceleration_factor;
                    } else if (optimize_end_position) {  // :215-219
                        M(i, end_position_offset + i) = b[3] + b[4] + b[5];
                    }
                    if (id > 0) {  // columns of the control point that begins this piece (:205-213, :224-232)
                        const int c0 = (id - 1) * 3 * Nact + i;
                        M(i, c0) = b[0] + b[1] + b[2];
                        M(i, c0 + Nact) = b[1] * velocity_factor + b[2] * 2.0 * velocity_factor;
                        M(i, c0 + 2 * Nact) = b[2] * acceleration_factor;
                    } else if (optimize_begin_position) {  // :196-200
                        M(i, begin_position_offset + i) = b[0] + b[1] + b[2];
                    }
                }
            }
        }
    }

    void compute(const double* z, bool compute_derivatives) {
        if (type == RAPTOR_TRAJ_PIECEWISE_BEZIER) {
            compute_piecewise(z, compute_derivatives);
            return;
        }
        if (type == RAPTOR_TRAJ_ARMOUR_BEZIER) {
            for (int j = 0; j < Nact; j++) {  // ArmourBezierCurves.cpp:96-99
                const double qfinal = q0[j] + z[j];
                coefficients(5, j) = qfinal;
                coefficients(4, j) = qfinal;
                coefficients(3, j) = qfinal;
            }
        } else {
            // Utils::reshape is a column-major map (Utils.h:142-146): coefficients(k, j) = z[j*(degree+1)+k];
            // BezierCurves.cpp:84-100: the first rows are pinned by the constrained initial state (T / 5.0 as written there)
            const int free_rows = degree + 1 - fixed_rows;
            for (int j = 0; j < Nact; j++) {
                if (fixed_rows >= 1) coefficients(0, j) = q0[j];
                if (fixed_rows == 2) coefficients(1, j) = q0[j] + q_d0[j] * T / 5.0;
                for (int k = 0; k < free_rows; k++) coefficients(fixed_rows + k, j) = z[j * free_rows + k];
            }
        }
        VecX B, dB, ddB;
        for (int x = 0; x < N; x++) {
            basis(x, B, dB, ddB);
            for (int j = 0; j < Nact; j++) {  // q = coefficients^T * B
                double s0 = 0, s1 = 0, s2 = 0;
                for (int k = 0; k <= degree; k++) {
                    s0 += coefficients(k, j) * B[k];
                    s1 += coefficients(k, j) * dB[k];
                    s2 += coefficients(k, j) * ddB[k];
                }
                q[x][j] = s0;
                q_d[x][j] = s1;
                q_dd[x][j] = s2;
            }
            if (compute_derivatives) {
                pq_pz[x].setZero();
                pq_d_pz[x].setZero();
                pq_dd_pz[x].setZero();
                if (type == RAPTOR_TRAJ_ARMOUR_BEZIER) {
                    for (int i = 0; i < Nact; i++) {  // ArmourBezierCurves.cpp:147-151
                        pq_pz[x](i, i) = B[3] + B[4] + B[5];
                        pq_d_pz[x](i, i) = dB[3] + dB[4] + dB[5];
                        pq_dd_pz[x](i, i) = ddB[3] + ddB[4] + ddB[5];
                    }
                } else {
                    const int free_rows = degree + 1 - fixed_rows;
                    for (int i = 0; i < Nact; i++)  // BezierCurves.cpp:148-169 (B.tail(...) blocks), Polynomials.cpp:75-79
                        for (int k = 0; k < free_rows; k++) {
                            pq_pz[x](i, i * free_rows + k) = B[fixed_rows + k];
                            pq_d_pz[x](i, i * free_rows + k) = dB[fixed_rows + k];
                            pq_dd_pz[x](i, i * free_rows + k) = ddB[fixed_rows + k];
                        }
                }
            }
        }
    }
};

// ------------------------------------------------------------------ InverseDynamics
struct InverseDynamics {
    const Model* model = nullptr;
    Trajectory* traj = nullptr;
    std::vector<VecX> tau;
    std::vector<MatX> ptau_pz;

    // KinematicsDynamics/src/InverseDynamics.cpp:73-139
    void compute(const double* z, bool compute_derivatives) {
        const int nv = model->nv;
        traj->compute(z, compute_derivatives);
        tau.assign(traj->N, VecX(nv, 0.0));
        if (compute_derivatives) ptau_pz.assign(traj->N, MatX(nv, traj->varLength));
        for (int i = 0; i < traj->N; i++) {
            const VecX& q = traj->q[i];
            const VecX& v = traj->q_d[i];
            const VecX& a = traj->q_dd[i];
            MatX prnea_pq, prnea_pv, prnea_pa;
            VecX t(nv, 0.0);
            if (compute_derivatives) {
                rnea_derivatives(*model, q.data(), v.data(), a.data(), t.data(), prnea_pq, prnea_pv, prnea_pa);
            } else {
                featherstone_rnea(*model, q.data(), v.data(), a.data(), nullptr, nullptr, nullptr, t.data(), nullptr, nullptr);
            }
            for (int j = 0; j < nv; j++)  // :115-121
                tau[i][j] = t[j] + model->damping[j] * v[j] + model->friction[j] * sign(v[j]);
            if (compute_derivatives) {
                for (int mi = 0; mi < nv; mi++)  // :126-130 (pinocchio fills the upper triangle only)
                    for (int mj = 0; mj <= mi; mj++) prnea_pa(mi, mj) = prnea_pa(mj, mi);
                for (int mi = 0; mi < nv; mi++) prnea_pv(mi, mi) += model->damping[mi];  // :133-135
                ptau_pz[i] = prnea_pq * traj->pq_pz[i] + prnea_pv * traj->pq_d_pz[i] + prnea_pa * traj->pq_dd_pz[i];  // :137-139
            }
        }
    }
};

// ------------------------------------------------------------------ Transform / ForwardKinematics
// KinematicsDynamics/src/Transform.cpp
struct Transform {
    Mat3 R = Mat3::Identity();
    Vec3 p;
    bool ifDerivative = false;
    Transform() {}
    Transform(const Mat3& R_, const Vec3& p_, bool d = false) : R(R_), p(p_), ifDerivative(d) {}
    explicit Transform(const Vec3& p_) : p(p_) {}
    // Transform.cpp:70-77  Rx*Ry*Rz
    static Transform fromRPY(const Vec3& rpy, const Vec3& p_) {
        Transform t;
        t.p = p_;
        const double cr = std::cos(rpy(0)), sr = std::sin(rpy(0));
        const double cp = std::cos(rpy(1)), sp = std::sin(rpy(1));
        const double cy = std::cos(rpy(2)), sy = std::sin(rpy(2));
        Mat3 Rx = Mat3::Identity(), Ry = Mat3::Identity(), Rz = Mat3::Identity();
        Rx(1, 1) = cr; Rx(1, 2) = -sr; Rx(2, 1) = sr; Rx(2, 2) = cr;
        Ry(0, 0) = cp; Ry(0, 2) = sp; Ry(2, 0) = -sp; Ry(2, 2) = cp;
        Rz(0, 0) = cy; Rz(0, 1) = -sy; Rz(1, 0) = sy; Rz(1, 1) = cy;
        t.R = Rx * Ry * Rz;
        return t;
    }
    // Transform.cpp:85-322
    Transform(int jtype, double theta, int order) {
        R = (order == 0) ? Mat3::Identity() : Mat3();
        p = Vec3();
        ifDerivative = (order > 0);
        const double c = std::cos(theta), s = std::sin(theta);
        // value and derivatives of (cos, sin): order 0: (c, s); 1: (-s, c); 2: (-c, -s); 3: (s, -c)
        double C, S;
        switch (order) {
            case 0: C = c; S = s; break;
            case 1: C = -s; S = c; break;
            case 2: C = -c; S = -s; break;
            case 3: C = s; S = -c; break;
            default: throw std::runtime_error("spatial.cpp: Transform(): Unsupported differentiate order!");
        }
        const int aj = std::abs(jtype);
        const double sg = (jtype > 0) ? 1.0 : -1.0;
        if (aj == 1) {
            R(1, 1) = C; R(1, 2) = -sg * S; R(2, 1) = sg * S; R(2, 2) = C;
        } else if (aj == 2) {
            R(0, 0) = C; R(0, 2) = sg * S; R(2, 0) = -sg * S; R(2, 2) = C;
        } else if (aj == 3) {
            R(0, 0) = C; R(0, 1) = -sg * S; R(1, 0) = sg * S; R(1, 1) = C;
        } else if (aj >= 4 && aj <= 6) {
            if (order == 0) p(aj - 4) = sg * theta;
            else if (order == 1) p(aj - 4) = sg;
        } else {
            throw std::runtime_error("spatial.cpp: Transform(): Unrecognized jtype!");
        }
    }
    // Transform.cpp:331-335
    Transform operator*(const Transform& x) const {
        const bool i = ifDerivative | x.ifDerivative;
        if (x.ifDerivative) return Transform(R * x.R, R * x.p, i);
        return Transform(R * x.R, R * x.p + p, i);
    }
    // Transform.cpp:379-390
    Vec3 getRPY() const;
};

// HigherOrderDerivatives.h:10-25
static inline double safeasin(double x) {
    if (x > 1.0) return M_PI_2;
    if (x < -1.0) return -M_PI_2;
    return std::asin(x);
}
Vec3 Transform::getRPY() const { return Vec3(std::atan2(-R(1, 2), R(2, 2)), safeasin(R(0, 2)), std::atan2(-R(0, 1), R(0, 0))); }

// SE3 * Transform (Transform.cpp:361-372)
static Transform placement_times(const Mat3& R, const Vec3& p, const Transform& t) {
    if (t.ifDerivative) return Transform(R * t.R, R * t.p, true);
    return Transform(R * t.R, R * t.p + p, false);
}

// KinematicsDynamics/src/ForwardKinematics.cpp:153-312 (orders 0 and 1; higher orders are added with the
// legged-robot rows).  start/end are Pinocchio joint ids (0 = universe).
struct ForwardKinematicsSolver {
    const Model* model = nullptr;
    std::vector<int> chain;
    Transform T;
    std::vector<Transform> dTdq;  // indexed by joint (0-based), valid for joints in chain

    void compute(int start, int end, const double* q, const Transform* startT, const Transform* endT, int order) {
        if (order < 0) throw std::invalid_argument("order has to be non-negative!");
        if (order > 1) throw std::invalid_argument("oracle FK: order > 1 not restated yet");
        chain.clear();
        int search_id = end;
        while (search_id != start) {  // :181-190
            chain.push_back(search_id - 1);
            if (search_id < 0 || search_id > model->nv || (int)chain.size() > model->nv)
                throw std::runtime_error("Can not find the end joint in the modelPtr_!");
            search_id = model->parent[search_id - 1] + 1;
        }
        std::reverse(chain.begin(), chain.end());
        const int n = (int)chain.size();
        const Transform T_start = startT ? *startT : Transform();
        // getTransformChain(q, a, b) = prod_{i=a..b} placement_i * Transform(jtype_i, q_i), built right-to-left (:135-150)
        std::vector<Transform> L(n), D(n);
        for (int i = 0; i < n; i++) {
            const int id = chain[i];
            L[i] = placement_times(model->plR[id], model->plp[id], Transform(model->jtype[id], q[id], 0));
            if (order >= 1) D[i] = placement_times(model->plR[id], model->plp[id], Transform(model->jtype[id], q[id], 1));
        }
        // suffix[i] = chain(i..n-1), suffix[n] = identity; prefix(0..i-1) uses the same right-to-left nesting
        std::vector<Transform> suffix(n + 1);
        for (int i = n - 1; i >= 0; i--) suffix[i] = (i == n - 1) ? L[i] : L[i] * suffix[i + 1];
        auto chain_prod = [&](int a, int b) -> Transform {  // a..b inclusive
            if (a > b) return Transform();
            Transform t = L[b];
            for (int i = b - 1; i >= a; i--) t = L[i] * t;
            return t;
        };
        T = T_start * (n > 0 ? suffix[0] : Transform());  // :206
        dTdq.assign(model->nv, Transform());
        if (order >= 1) {
            for (int i = 0; i < n; i++)  // :209-214
                dTdq[chain[i]] = T_start * chain_prod(0, i - 1) * D[i] * suffix[i + 1];
        }
        if (endT) {  // :290-311  (operator*=, Transform.cpp:337-347)
            T = T * (*endT);
            if (order >= 1)
                for (int id : chain) dTdq[id] = dTdq[id] * (*endT);
        }
    }
    Vec3 getTranslation() const { return T.p; }
    // :330-341
    MatX getTranslationJacobian() const {
        MatX J(3, model->nv);
        for (int id : chain)
            for (int k = 0; k < 3; k++) J(k, id) = dTdq[id].p(k);
        return J;
    }
};

// ------------------------------------------------------------------ BoxCollisionAvoidance
// Constraints/src/BoxCollisionAvoidance.cpp
struct BoxCollisionAvoidance {
    static constexpr double SQUARE_ROOT_THRESHOLD = 1e-12;  // Constraints/include/BoxCollisionAvoidance.h:12
    int numObstacles = 0;
    std::vector<Vec3> boxCenters, boxSize;
    std::vector<Mat3> boxR;
    double minimumDistance = 0;
    int minimumDistanceIndex = 0;
    MatX pdistances_pz;

    void init(int O, const double* c, const double* rpy, const double* size) {
        numObstacles = O;
        boxCenters.resize(O);
        boxSize.resize(O);
        boxR.resize(O);
        for (int o = 0; o < O; o++) {
            boxCenters[o] = Vec3(c[3 * o], c[3 * o + 1], c[3 * o + 2]);
            boxSize[o] = Vec3(size[3 * o], size[3 * o + 1], size[3 * o + 2]);
            // :59-61  R = Rz(yaw) * Ry(pitch) * Rx(roll)
            const double cr = std::cos(rpy[3 * o]), sr = std::sin(rpy[3 * o]);
            const double cp = std::cos(rpy[3 * o + 1]), sp = std::sin(rpy[3 * o + 1]);
            const double cy = std::cos(rpy[3 * o + 2]), sy = std::sin(rpy[3 * o + 2]);
            Mat3 Rx = Mat3::Identity(), Ry = Mat3::Identity(), Rz = Mat3::Identity();
            Rx(1, 1) = cr; Rx(1, 2) = -sr; Rx(2, 1) = sr; Rx(2, 2) = cr;
            Ry(0, 0) = cp; Ry(0, 2) = sp; Ry(2, 0) = -sp; Ry(2, 2) = cp;
            Rz(0, 0) = cy; Rz(0, 1) = -sy; Rz(1, 0) = sy; Rz(1, 1) = cy;
            boxR[o] = Rz * Ry * Rx;
        }
    }

    // :92-137
    Vec3 diffClosest(const Vec3& point, int obs_id, double& isInside) const {
        Vec3 localPoint = boxR[obs_id].transpose() * (point - boxCenters[obs_id]);
        Vec3 localClosestPoint;
        isInside = -1.0;
        double dface[3][2];
        for (int i = 0; i < 3; i++) {
            dface[i][0] = localPoint(i) + boxSize[obs_id](i) / 2.0;
            dface[i][1] = boxSize[obs_id](i) / 2.0 - localPoint(i);
            localClosestPoint(i) = std::fmax(-boxSize[obs_id](i) / 2.0, std::fmin(localPoint(i), boxSize[obs_id](i) / 2.0));
            if (localClosestPoint(i) != localPoint(i)) isInside = 1.0;
        }
        if (isInside == -1.0) {
            double minDist = 1e19;
            int faceIndex = 0, directionIndex = 0;
            for (int i = 0; i < 3; i++) {
                if (dface[i][0] < minDist) { minDist = dface[i][0]; faceIndex = i; directionIndex = 0; }
                if (dface[i][1] < minDist) { minDist = dface[i][1]; faceIndex = i; directionIndex = 1; }
            }
            localClosestPoint(faceIndex) = (directionIndex == 0) ? -boxSize[obs_id](faceIndex) / 2.0 : boxSize[obs_id](faceIndex) / 2.0;
        }
        Vec3 closestPoint = boxR[obs_id] * localClosestPoint + boxCenters[obs_id];
        return point - closestPoint;
    }

    // :139-204
    Vec3 diffClosest(const Vec3& point, const MatX& ppoint_pz, int obs_id, MatX& pdiff_pz, double& isInside) const {
        const int W = ppoint_pz.c;
        const Mat3 Rt = boxR[obs_id].transpose();
        Vec3 localPoint = Rt * (point - boxCenters[obs_id]);
        MatX plocal(3, W);
        for (int i = 0; i < 3; i++)
            for (int c = 0; c < W; c++) plocal(i, c) = Rt(i, 0) * ppoint_pz(0, c) + Rt(i, 1) * ppoint_pz(1, c) + Rt(i, 2) * ppoint_pz(2, c);
        Vec3 localClosestPoint;
        MatX plc(3, W);
        isInside = -1.0;
        double dface[3][2];
        for (int i = 0; i < 3; i++) {
            dface[i][0] = localPoint(i) + boxSize[obs_id](i) / 2.0;
            dface[i][1] = boxSize[obs_id](i) / 2.0 - localPoint(i);
            if (localPoint(i) < -boxSize[obs_id](i) / 2.0) {
                localClosestPoint(i) = -boxSize[obs_id](i) / 2.0;
                isInside = 1.0;
            } else if (localPoint(i) > boxSize[obs_id](i) / 2.0) {
                localClosestPoint(i) = boxSize[obs_id](i) / 2.0;
                isInside = 1.0;
            } else {
                localClosestPoint(i) = localPoint(i);
                for (int c = 0; c < W; c++) plc(i, c) = plocal(i, c);
            }
        }
        if (isInside == -1.0) {
            double minDist = 1e19;
            int faceIndex = 0, directionIndex = 0;
            for (int i = 0; i < 3; i++) {
                if (dface[i][0] < minDist) { minDist = dface[i][0]; faceIndex = i; directionIndex = 0; }
                if (dface[i][1] < minDist) { minDist = dface[i][1]; faceIndex = i; directionIndex = 1; }
            }
            localClosestPoint(faceIndex) = (directionIndex == 0) ? -boxSize[obs_id](faceIndex) / 2.0 : boxSize[obs_id](faceIndex) / 2.0;
            for (int c = 0; c < W; c++) plc(faceIndex, c) = 0.0;
        }
        Vec3 closestPoint = boxR[obs_id] * localClosestPoint + boxCenters[obs_id];
        pdiff_pz = MatX(3, W);
        const Mat3& R = boxR[obs_id];
        for (int i = 0; i < 3; i++)
            for (int c = 0; c < W; c++)
                pdiff_pz(i, c) = ppoint_pz(i, c) - (R(i, 0) * plc(0, c) + R(i, 1) * plc(1, c) + R(i, 2) * plc(2, c));
        return point - closestPoint;
    }

    // :290-310
    void computeDistance(const Vec3& point) {
        minimumDistance = 1e19;
        double isInside = 1.0;
        for (int o = 0; o < numObstacles; o++) {
            Vec3 diff = diffClosest(point, o, isInside);
            const double d2 = diff.dot(diff);
            const double dist = (d2 > SQUARE_ROOT_THRESHOLD) ? isInside * std::sqrt(d2) : 0.0;
            if (dist < minimumDistance) {
                minimumDistance = dist;
                minimumDistanceIndex = o;
            }
        }
    }
    // :312-373 with onlyComputeDerivativesForMinimumDistance == true (KinovaCustomizedConstraints.cpp:51)
    void computeDistance(const Vec3& point, const MatX& ppoint_pz) {
        if (ppoint_pz.r != 3) throw std::invalid_argument("ppoint_pz should have 3 rows");
        computeDistance(point);
        pdistances_pz = MatX(numObstacles, ppoint_pz.c);
        MatX pdiff_pz;
        double isInside = 1.0;
        Vec3 diff = diffClosest(point, ppoint_pz, minimumDistanceIndex, pdiff_pz, isInside);
        const double d2 = diff.dot(diff);
        if (d2 > SQUARE_ROOT_THRESHOLD) {
            const double nrm = std::sqrt(d2);
            for (int c = 0; c < ppoint_pz.c; c++)
                pdistances_pz(minimumDistanceIndex, c) =
                    isInside * (diff(0) * pdiff_pz(0, c) + diff(1) * pdiff_pz(1, c) + diff(2) * pdiff_pz(2, c)) / nrm;
        }
    }
};

// ------------------------------------------------------------------ Constraints
struct Constraints {
    int m = 0;
    VecX g, g_lb, g_ub;
    MatX pg_pz;
    virtual ~Constraints() {}
    virtual void compute(const double* z, bool compute_derivatives) = 0;
    virtual void compute_bounds() = 0;
    void initialize_memory(int m_, int varLength) {  // Constraints.cpp:38-56
        m = m_;
        g.assign(m, 0.0);
        g_lb.assign(m, 0.0);
        g_ub.assign(m, 0.0);
        pg_pz = MatX(m, varLength);
    }
};

// Constraints/src/JointLimits.cpp:23-53 and VelocityLimits.cpp:23-52 (same shape, q vs q_d)
struct StateLimits : Constraints {
    Trajectory* traj;
    VecX lower, upper;
    bool velocity;
    StateLimits(Trajectory* t, const VecX& lo, const VecX& up, bool vel) : traj(t), lower(lo), upper(up), velocity(vel) {
        if ((int)lo.size() != t->Nact || (int)up.size() != t->Nact)
            throw std::invalid_argument("lowerLimits and upperLimits must be the same size as the number of actuated joints");
        initialize_memory(t->N * t->Nact, t->varLength);
    }
    void compute(const double* z, bool d) override {
        traj->compute(z, d);
        const int Nact = traj->Nact;
        for (int i = 0; i < traj->N; i++) {
            const VecX& src = velocity ? traj->q_d[i] : traj->q[i];
            for (int j = 0; j < Nact; j++) g[i * Nact + j] = src[j];
            if (d) {
                const MatX& P = velocity ? traj->pq_d_pz[i] : traj->pq_pz[i];
                for (int j = 0; j < Nact; j++)
                    for (int c = 0; c < traj->varLength; c++) pg_pz(i * Nact + j, c) = P(j, c);
            }
        }
    }
    void compute_bounds() override {
        for (int i = 0; i < traj->N; i++)
            for (int j = 0; j < traj->Nact; j++) {
                g_lb[i * traj->Nact + j] = lower[j];
                g_ub[i * traj->Nact + j] = upper[j];
            }
    }
};

// Constraints/src/TorqueLimits.cpp:18-50
struct TorqueLimits : Constraints {
    Trajectory* traj;
    InverseDynamics* id;
    VecX lower, upper;
    TorqueLimits(Trajectory* t, InverseDynamics* i, const VecX& lo, const VecX& up) : traj(t), id(i), lower(lo), upper(up) {
        initialize_memory(t->N * t->Nact, t->varLength);
    }
    void compute(const double* z, bool d) override {
        id->compute(z, d);
        const int Nact = traj->Nact;
        for (int i = 0; i < traj->N; i++) {
            for (int j = 0; j < Nact; j++) g[i * Nact + j] = id->tau[i][j];
            if (d)
                for (int j = 0; j < Nact; j++)
                    for (int c = 0; c < traj->varLength; c++) pg_pz(i * Nact + j, c) = id->ptau_pz[i](j, c);
        }
    }
    void compute_bounds() override {
        for (int i = 0; i < traj->N; i++)
            for (int j = 0; j < traj->Nact; j++) {
                g_lb[i * traj->Nact + j] = lower[j];
                g_ub[i * traj->Nact + j] = upper[j];
            }
    }
};

// Examples/Kinova/CollisionAvoidanceTrajectory/include/KinovaCustomizedConstraints.h:21-63
static const int NUM_SPHERES = 17;
static const int SPHERE_JOINT_ID[NUM_SPHERES] = {2, 2, 2, 2, 2, 2, 2, 4, 4, 4, 4, 4, 6, 6, 7, 7, 7};
static const double SPHERE_OFFSET[NUM_SPHERES][3] = {
    {0.0, -0.05, -0.01}, {0.0, -0.11, -0.01}, {0.0, -0.17, -0.01}, {0.0, -0.24, -0.01}, {0.0, -0.30, -0.01},
    {0.0, -0.36, -0.01}, {0.0, -0.43, -0.01}, {0.0, -0.07, 0.0},   {0.0, -0.14, 0.0},   {0.0, -0.21, 0.0},
    {0.0, -0.28, 0.0},   {0.0, -0.32, 0.0},   {0.0, -0.07, 0.0},   {0.0, -0.14, 0.0},   {-0.03, -0.06, -0.055},
    {0.0, -0.06, -0.055}, {0.03, -0.06, -0.055}};
static const double SPHERE_RADIUS[NUM_SPHERES] = {0.06, 0.06, 0.06, 0.06, 0.06, 0.06, 0.07, 0.07, 0.06,
                                                  0.06, 0.06, 0.06, 0.05, 0.05, 0.02, 0.02, 0.02};

// Examples/Kinova/CollisionAvoidanceTrajectory/src/KinovaCustomizedConstraints.cpp
struct KinovaCustomizedConstraints : Constraints {
    Trajectory* traj;
    const Model* model;
    ForwardKinematicsSolver fk;
    BoxCollisionAvoidance box;
    double collision_buffer;
    int num_spheres;
    std::vector<int> sphere_joint_id;
    std::vector<Vec3> sphere_offset;
    std::vector<double> sphere_radius;
    std::vector<std::pair<int, int>> tapered_capsules;  // KinovaCustomizedConstraints.h:58-63
    KinovaCustomizedConstraints(Trajectory* t, const Model* mdl, int O, const double* bc, const double* brpy,
                                const double* bs, bool include_gripper, double buffer)
        : traj(t), model(mdl), collision_buffer(buffer) {
        fk.model = mdl;
        box.init(O, bc, brpy, bs);
        num_spheres = NUM_SPHERES;
        for (int i = 0; i < NUM_SPHERES; i++) {  // :23-33
            sphere_joint_id.push_back(SPHERE_JOINT_ID[i]);
            sphere_offset.push_back(Vec3(SPHERE_OFFSET[i][0], SPHERE_OFFSET[i][1], SPHERE_OFFSET[i][2]));
            sphere_radius.push_back(SPHERE_RADIUS[i]);
        }
        tapered_capsules = {{0, 6}, {7, 11}, {12, 13}};  // :36-39
        if (include_gripper) tapered_capsules.back().second = 19;  // :46
        if (include_gripper) {  // :42-47
            const double zs[3] = {-0.10, -0.15, -0.20};
            for (int k = 0; k < 3; k++) {
                num_spheres++;
                sphere_joint_id.push_back(7);
                sphere_offset.push_back(Vec3(0.0, 0.0, zs[k]));
                sphere_radius.push_back(0.05);
            }
        }
        initialize_memory(t->N * num_spheres, t->varLength);
    }
    // :73-158
    void compute(const double* z, bool d) override {
        traj->compute(z, d);
        if (box.numObstacles == 0) return;  // :80-82, g / pg_pz untouched
        for (int i = 0; i < traj->N; i++) {
            const VecX& q = traj->q[i];
            const MatX& pq_pz = traj->pq_pz[i];
            for (int j = 0; j < num_spheres; j++) {
                Transform endT(sphere_offset[j]);
                fk.compute(0, sphere_joint_id[j], q.data(), nullptr, &endT, d ? 1 : 0);
                Vec3 c = fk.getTranslation();
                if (d) {
                    MatX pc = fk.getTranslationJacobian() * pq_pz;  // :112
                    box.computeDistance(c, pc);
                } else {
                    box.computeDistance(c);
                }
                g[i * num_spheres + j] = box.minimumDistance - sphere_radius[j];  // :145
                if (d)
                    for (int cc = 0; cc < traj->varLength; cc++)
                        pg_pz(i * num_spheres + j, cc) = box.pdistances_pz(box.minimumDistanceIndex, cc);  // :148-149
            }
        }
    }
    void compute_bounds() override {  // :160-164
        std::fill(g_lb.begin(), g_lb.end(), collision_buffer);
        std::fill(g_ub.begin(), g_ub.end(), 1e19);
    }
    // sphere_centers_copy(j, i) and sphere_centers_gradient_copy(j, i) (:109, :114) for a trajectory already computed
    void sphere_center(int i, int j, bool d, Vec3& c, MatX& pc) {
        Transform endT(sphere_offset[j]);
        fk.compute(0, sphere_joint_id[j], traj->q[i].data(), nullptr, &endT, d ? 1 : 0);
        c = fk.getTranslation();
        if (d) pc = fk.getTranslationJacobian() * traj->pq_pz[i];
    }
};

#include "../raptor_amd/csrc/digit_constants.h"
#include "../raptor_amd/csrc/digit_tables.h"
#include "legged_oracle.inc"
#include "hessian_oracle.inc"

// ------------------------------------------------------------------ problem = Optimizer's constraint list
struct Problem {
    Model model;
    Trajectory traj;
    InverseDynamics id;
    ContactDynamicsConstraints dc;
    ConstrainedInverseDynamics cid;
    std::vector<std::unique_ptr<Constraints>> cons;  // constraintsPtrVec_ order
    std::unique_ptr<Constraints> link;  // reset-map tables kept for the multiple-step link only (period.enabled == RAPTOR_PERIOD_LINK_ONLY)
    bool is_kinova = false;
    VecX q_des;        // KinovaOptimizer::q_des, tplan_n (KinovaOptimizer.cpp:39-40); tplan_n < 0: objective not set
    int tplan_n = -1;
    int n = 0, m = 0;
    std::vector<int> offsets;
    void finish() {
        id.model = &model;
        id.traj = &traj;
        n = traj.varLength;
        offsets.assign(1, 0);
        for (auto& c : cons) offsets.push_back(offsets.back() + c->m);
        m = offsets.back();
    }
};

// Examples/Kinova/KinovaConstants.h:7-29
static const double JOINT_LIMITS_LOWER[7] = {-1e19, -128.9, -1e19, -147.8, -1e19, -120.3, -1e19};
static const double JOINT_LIMITS_UPPER[7] = {1e19, 128.9, 1e19, 147.8, 1e19, 120.3, 1e19};
static const double VELOCITY_LIMITS_LOWER[7] = {-100, -100, -100, -100, -180, -180, -180};
static const double VELOCITY_LIMITS_UPPER[7] = {100, 100, 100, 100, 180, 180, 180};
static const double TORQUE_LIMITS_LOWER[7] = {-90.0, -90.0, -90.0, -90.0, -45.0, -45.0, -45.0};
static const double TORQUE_LIMITS_UPPER[7] = {90.0, 90.0, 90.0, 90.0, 45.0, 45.0, 45.0};
static inline double deg2rad(double deg) { return deg * M_PI / 180.0; }  // Utils.h:17-19

// ------------------------------------------------------------------ TaperedCapsuleCollision
// Constraints/src/TaperedCapsuleCollision.cpp:7-22
static inline void solve_quadratic(double a, double b, double c, double* sol1, double* sol2) {
    double discriminant = (b * b) - (4 * a * c);
    if (discriminant < 0) {
        *sol1 = -1.0;
        *sol2 = -1.0;
        return;
    }
    double sqrt_discriminant = std::sqrt(discriminant);
    double twoa = 2 * a;
    *sol1 = (-b + sqrt_discriminant) / twoa;
    *sol2 = (-b - sqrt_discriminant) / twoa;
}

// batchDot (:31-41): result[i] = vector . matrix.col(i); matrices are 3 x F row-major here
static inline void batchDot(const Vec3& v, const double* M, int F, double* out) {
    for (int i = 0; i < F; i++) out[i] = v(0) * M[i] + v(1) * M[F + i] + v(2) * M[2 * F + i];
}

// :43-274.  P = 4 endpoints, dP = four 3xF Jacobians (NULL: distance only), R = 4 radii.
// All reference quirks kept: un-rooted radicand in t_test[8],[9] (:131-134), pu_pz == pt_pz in case 8/9 (:257-258).
static double capsule_distance(const double* P, const double* dP, const double* Rr, int F, double* pdist_pz) {
    const Vec3 tc1_point_1(P[0], P[1], P[2]), tc1_point_2(P[3], P[4], P[5]);
    const Vec3 tc2_point_1(P[6], P[7], P[8]), tc2_point_2(P[9], P[10], P[11]);
    const double tc1_radius_1 = Rr[0], tc1_radius_2 = Rr[1], tc2_radius_1 = Rr[2], tc2_radius_2 = Rr[3];

    double r1 = tc1_radius_2 - tc1_radius_1;
    double r2 = tc2_radius_2 - tc2_radius_1;
    Vec3 d1 = tc1_point_2 - tc1_point_1;
    Vec3 d2 = tc2_point_2 - tc2_point_1;
    Vec3 d12 = tc2_point_1 - tc1_point_1;
    double a = d1.dot(d1), b = d1.dot(d2), c = d1.dot(d12), e = d2.dot(d2), f = d2.dot(d12), g = d12.dot(d12);
    double distances[14], u_test[14], t_test[14];
    u_test[10] = 0.0; u_test[11] = 0.0; u_test[12] = 1.0; u_test[13] = 1.0;
    t_test[10] = 0.0; t_test[11] = 1.0; t_test[12] = 0.0; t_test[13] = 1.0;
    if (std::abs(r2) > 1e-10) {
        solve_quadratic(e * (e - r2 * r2), 2 * f * (e - r2 * r2), f * f - r2 * r2 * g, &u_test[0], &u_test[1]);
        solve_quadratic(e * (e - r2 * r2), 2 * (f - b) * (e - r2 * r2), (f - b) * (f - b) - r2 * r2 * (g - 2 * c + a), &u_test[2], &u_test[3]);
    } else {
        u_test[0] = -f * (e - r2 * r2) / (e * (e - r2 * r2));
        u_test[1] = -1;
        u_test[2] = -(f - b) * (e - r2 * r2) / (e * (e - r2 * r2));
        u_test[3] = -1;
    }
    t_test[0] = 0; t_test[1] = 0; t_test[2] = 1; t_test[3] = 1;
    u_test[4] = 0; u_test[5] = 0; u_test[6] = 1; u_test[7] = 1;
    if (std::abs(r1) > 1e-10) {
        solve_quadratic(a * (a - std::pow(r1, 2)), -2 * c * (a - std::pow(r1, 2)), std::pow(c, 2) - std::pow(r1, 2) * g, &t_test[4], &t_test[5]);
        solve_quadratic(a * (a - std::pow(r1, 2)), -2 * (b + c) * (a - std::pow(r1, 2)), std::pow((b + c), 2) - std::pow(r1, 2) * (e + 2 * f + g), &t_test[6], &t_test[7]);
    } else {
        t_test[4] = c * (a - std::pow(r1, 2)) / (a * (a - std::pow(r1, 2)));
        t_test[5] = -1;
        t_test[6] = (b + c) * (a - std::pow(r1, 2)) / (a * (a - std::pow(r1, 2)));
        t_test[7] = -1;
    }
    double radicand = (g * b * b - 2 * b * c * f + e * c * c + a * f * f - a * e * g) * (b * b + 2 * b * r1 * r2 + e * r1 * r1 + a * r2 * r2 - a * e);
    if (radicand < 0) {
        t_test[8] = -1.0; t_test[9] = -1.0; u_test[8] = -1.0; u_test[9] = -1.0;
    } else {
        double sqrt_radicand = std::sqrt(radicand);
        const double den = (a * a * e * e - a * a * e * r2 * r2 - 2 * a * b * b * e + a * b * b * r2 * r2 - 2 * a * b * e * r1 * r2 - a * e * e * r1 * r1 + b * b * b * b + 2 * b * b * b * r1 * r2 + b * b * e * r1 * r1);
        const double rad2 = ((g * b * b - 2 * b * c * f + e * c * c + a * f * f - a * e * g) * (b * b + 2 * b * r1 * r2 + e * r1 * r1 + a * r2 * r2 - a * e));
        t_test[8] = (c * r2 + f * r1) / (a * r2 + b * r1) + ((b * r2 + e * r1) * (a * r2 * sqrt_radicand - b * b * b * c + b * r1 * rad2
                    + a * a * f * r2 * r2 + a * b * b * f - a * a * e * f - a * b * c * r2 * r2 - b * c * e * r1 * r1 + a * e * f * r1 * r1 - 2 * b * b * c * r1 * r2 + a * b * c * e + 2 * a * b * f * r1 * r2)) / ((a * r2 + b * r1) * den);
        t_test[9] = (c * r2 + f * r1) / (a * r2 + b * r1) - ((b * r2 + e * r1) * (b * b * b * c + a * r2 * sqrt_radicand + b * r1 * rad2
                    - a * a * f * r2 * r2 - a * b * b * f + a * a * e * f + a * b * c * r2 * r2 + b * c * e * r1 * r1 - a * e * f * r1 * r1 + 2 * b * b * c * r1 * r2 - a * b * c * e - 2 * a * b * f * r1 * r2)) / ((a * r2 + b * r1) * den);
        u_test[8] = (a * r2 * sqrt_radicand - b * b * b * c + b * r1 * sqrt_radicand
                     + a * a * f * r2 * r2 + a * b * b * f - a * a * e * f - a * b * c * r2 * r2 - b * c * e * r1 * r1 + a * e * f * r1 * r1 - 2 * b * b * c * r1 * r2 + a * b * c * e + 2 * a * b * f * r1 * r2) / den;
        u_test[9] = -(b * b * b * c + a * r2 * sqrt_radicand + b * r1 * sqrt_radicand
                      - a * a * f * r2 * r2 - a * b * b * f + a * a * e * f + a * b * c * r2 * r2 + b * c * e * r1 * r1 - a * e * f * r1 * r1 + 2 * b * b * c * r1 * r2 - a * b * c * e - 2 * a * b * f * r1 * r2) / den;
    }
    for (int i = 0; i < 14; i++) {
        if (u_test[i] <= 1 && u_test[i] >= 0 && t_test[i] <= 1 && t_test[i] >= 0)
            distances[i] = (d1 * t_test[i] - d2 * u_test[i] - d12).norm() - r1 * t_test[i] - r2 * u_test[i];
        else
            distances[i] = INFINITY;
    }
    size_t ind = 0;
    for (int i = 1; i < 14; i++)
        if (distances[i] < distances[ind]) ind = i;
    double finalDistance = distances[ind] - tc1_radius_1 - tc2_radius_1;
    double uStar = u_test[ind], tStar = t_test[ind];

    if (dP != nullptr && pdist_pz != nullptr) {
        const double* p11 = dP;
        const double* p12 = dP + 3 * F;
        const double* p21 = dP + 6 * F;
        const double* p22 = dP + 9 * F;
        std::vector<double> pd1(3 * F), pd2(3 * F), pd12(3 * F);
        for (int i = 0; i < 3 * F; i++) {
            pd1[i] = p12[i] - p11[i];
            pd2[i] = p22[i] - p21[i];
            pd12[i] = p21[i] - p11[i];
        }
        std::vector<double> t1(F), t2(F), pa_pz(F), pb_pz(F), pc_pz(F), pe_pz(F), pf_pz(F);
        batchDot(d1, pd1.data(), F, t1.data());
        for (int i = 0; i < F; i++) pa_pz[i] = 2 * t1[i];
        batchDot(d1, pd2.data(), F, t1.data());
        batchDot(d2, pd1.data(), F, t2.data());
        for (int i = 0; i < F; i++) pb_pz[i] = t1[i] + t2[i];
        batchDot(d12, pd1.data(), F, t1.data());
        batchDot(d1, pd12.data(), F, t2.data());
        for (int i = 0; i < F; i++) pc_pz[i] = t1[i] + t2[i];
        batchDot(d2, pd2.data(), F, t1.data());
        for (int i = 0; i < F; i++) pe_pz[i] = 2 * t1[i];
        batchDot(d2, pd12.data(), F, t1.data());
        batchDot(d12, pd2.data(), F, t2.data());
        for (int i = 0; i < F; i++) pf_pz[i] = t1[i] + t2[i];

        std::vector<double> lgrey(F), red(F), grey(F);
        for (int i = 0; i < F; i++) lgrey[i] = pa_pz[i] * tStar - pb_pz[i] * uStar - pc_pz[i];
        Vec3 dist_vec = d1 * tStar - d2 * uStar - d12;
        double purple = dist_vec.norm();
        if (purple == 0) purple = 1e-6;
        double orange = a * tStar - b * uStar - c;
        double green = -b * tStar - e * uStar + f;
        for (int i = 0; i < F; i++) red[i] = -pb_pz[i] * tStar + pe_pz[i] * uStar + pf_pz[i];
        Vec3 grey_intermediate = d1 * tStar * tStar - d2 * tStar * uStar - d12 * tStar;
        {
            std::vector<double> g1(F), g2(F), g3(F);
            batchDot(grey_intermediate, pd1.data(), F, g1.data());
            batchDot(grey_intermediate, pd2.data(), F, g2.data());
            batchDot((d1 - d2 - d12), pd12.data(), F, g3.data());
            for (int i = 0; i < F; i++) grey[i] = (g1[i] - g2[i] - g3[i]) / (purple * purple);
        }
        double gradient_factor = 1 / std::pow(purple, 3);
        std::vector<double> dL1(F, 0.0), dL2(F, 0.0), dL3(F, 0.0), dL4(F, 0.0);
        for (int i = 0; i < F; i++) {
            if (tStar == 1) dL1[i] = -(lgrey[i] * purple - orange * grey[i]);
            if (tStar == 0) dL2[i] = -(red[i] * purple - green * grey[i]);
            if (uStar == 1) dL3[i] = (lgrey[i] * purple - orange * grey[i]);
            if (uStar == 0) dL4[i] = (red[i] * purple - green * grey[i]);
        }
        std::vector<double> Hin(F), comb(3 * F);
        for (int i = 0; i < 3 * F; i++) comb[i] = -pd1[i] * tStar + pd2[i] * uStar + pd12[i];
        batchDot(d1 * -tStar + d2 * uStar + d12, comb.data(), F, Hin.data());
        std::vector<double> H1(F, 0.0), H2(F, 0.0), pt_pz(F, 0.0), pu_pz(F, 0.0);
        double L11, L22, L12;
        switch (ind) {
            case 0: case 1: case 2: case 3:
                L22 = e / purple - gradient_factor * green * green;
                for (int i = 0; i < F; i++) {
                    H2[i] = red[i] / purple + dL2[i] - dL4[i] - green * Hin[i] * gradient_factor;
                    pu_pz[i] = -H2[i] / L22;
                }
                break;
            case 4: case 5: case 6: case 7:
                L11 = a / purple - gradient_factor * orange * orange;
                for (int i = 0; i < F; i++) {
                    H1[i] = lgrey[i] / purple + dL1[i] - dL3[i] - orange * Hin[i] * gradient_factor;
                    pt_pz[i] = -H1[i] / L11;
                }
                break;
            case 8: case 9:
                L11 = a / purple - gradient_factor * orange * orange;
                L22 = e / purple - gradient_factor * green * green;
                L12 = -b / purple - gradient_factor * orange * green;
                for (int i = 0; i < F; i++) {
                    H2[i] = red[i] / purple + dL2[i] - dL4[i] - green * Hin[i] * gradient_factor;
                    H1[i] = lgrey[i] / purple + dL1[i] - dL3[i] - orange * Hin[i] * gradient_factor;
                    pt_pz[i] = (L12 * H2[i] - L22 * H1[i]) / (L11 * L22 - L12 * L12);
                    pu_pz[i] = (L12 * H2[i] - L22 * H1[i]) / (L11 * L22 - L12 * L12);
                }
                break;
            default:
                break;
        }
        // :268-270
        for (int i = 0; i < F; i++) {
            double s = 0;
            for (int k = 0; k < 3; k++) {
                const double col = pd1[k * F + i] * tStar + d1(k) * pt_pz[i] - pd2[k * F + i] * uStar - d2(k) * pu_pz[i] - pd12[k * F + i];
                s += dist_vec(k) * col;
            }
            pdist_pz[i] = s / purple - r1 * pt_pz[i] - r2 * pu_pz[i];
        }
    }
    return finalDistance;
}

}  // namespace oracle

// =====================================================================================
// C ABI (same signatures as include/raptor_hip.h, prefix oracle_, host pointers)
// =====================================================================================
using namespace oracle;

struct raptor_model {
    Model m;
};
struct raptor_problem {
    Problem p;
};

#define ORACLE_TRY try {
#define ORACLE_CATCH                                \
    }                                               \
    catch (const std::invalid_argument& e) {        \
        g_err = e.what();                           \
        return RAPTOR_ERR_INVALID_ARGUMENT;         \
    }                                               \
    catch (const std::exception& e) {               \
        g_err = e.what();                           \
        return RAPTOR_ERR_RUNTIME;                  \
    }                                               \
    return RAPTOR_OK;

extern "C" {

const char* oracle_last_error(void) { return g_err.c_str(); }

int oracle_model_create(const raptor_model_desc* desc, raptor_model** out) {
    ORACLE_TRY
    if (!out) throw std::invalid_argument("out is null");
    auto* mm = new raptor_model();
    try {
        model_from_desc(desc, mm->m);
    } catch (...) {
        delete mm;
        throw;
    }
    *out = mm;
    ORACLE_CATCH
}
void oracle_model_destroy(raptor_model* m) { delete m; }

int oracle_id_problem_create(const raptor_model* model, const raptor_traj_desc* traj, raptor_problem** out) {
    ORACLE_TRY
    if (!model || !out) throw std::invalid_argument("null argument");
    std::unique_ptr<raptor_problem> pr(new raptor_problem());
    Problem& p = pr->p;
    p.model = model->m;
    p.traj.init(traj);
    if (p.traj.Nact != p.model.nv) throw std::invalid_argument("InverseDynamics: trajectory Nact must equal model nv");
    VecX lo(p.traj.Nact, -1e19), up(p.traj.Nact, 1e19);
    p.cons.emplace_back(new TorqueLimits(&p.traj, &p.id, lo, up));
    p.finish();
    *out = pr.release();
    ORACLE_CATCH
}

int oracle_kinova_problem_create(const raptor_model* model, const raptor_kinova_desc* d, raptor_problem** out) {
    ORACLE_TRY
    if (!model || !d || !out) throw std::invalid_argument("null argument");
    std::unique_ptr<raptor_problem> pr(new raptor_problem());
    Problem& p = pr->p;
    p.model = model->m;
    if (d->traj.type != RAPTOR_TRAJ_ARMOUR_BEZIER && d->traj.type != RAPTOR_TRAJ_PIECEWISE_BEZIER)
        throw std::invalid_argument("KinovaOptimizer uses ArmourBezierCurves, KinovaLongerHorizonOptimizer PiecewiseBezierCurves");
    p.traj.init(&d->traj);
    if (p.model.nv != 7 || p.traj.Nact != 7) throw std::invalid_argument("KinovaOptimizer: NUM_JOINTS is 7");
    if (d->num_obstacles < 0) throw std::invalid_argument("num_obstacles < 0");
    VecX jl(7), ju(7), vl(7), vu(7), tl(7), tu(7);
    for (int j = 0; j < 7; j++) {  // KinovaOptimizer.cpp:52-80
        const double jb = d->joint_limits_buffer ? d->joint_limits_buffer[j] : 0.0;
        const double vb = d->velocity_limits_buffer ? d->velocity_limits_buffer[j] : 0.0;
        const double tb = d->torque_limits_buffer ? d->torque_limits_buffer[j] : 0.0;
        jl[j] = deg2rad(JOINT_LIMITS_LOWER[j]) + jb;
        ju[j] = deg2rad(JOINT_LIMITS_UPPER[j]) - jb;
        vl[j] = deg2rad(VELOCITY_LIMITS_LOWER[j]) + vb;
        vu[j] = deg2rad(VELOCITY_LIMITS_UPPER[j]) - vb;
        tl[j] = TORQUE_LIMITS_LOWER[j] + tb;
        tu[j] = TORQUE_LIMITS_UPPER[j] - tb;
    }
    p.cons.emplace_back(new StateLimits(&p.traj, jl, ju, false));  // :83-86
    p.cons.emplace_back(new StateLimits(&p.traj, vl, vu, true));   // :89-92
    p.cons.emplace_back(new TorqueLimits(&p.traj, &p.id, tl, tu)); // :95-99
    p.cons.emplace_back(new KinovaCustomizedConstraints(&p.traj, &p.model, d->num_obstacles, d->box_center, d->box_rpy,
                                                        d->box_size, d->include_gripper != 0, d->collision_buffer));  // :102-109
    p.is_kinova = true;
    p.finish();
    *out = pr.release();
    ORACLE_CATCH
}

int oracle_legged_problem_create(const raptor_model* model, const raptor_legged_desc* d, raptor_problem** out) {
    ORACLE_TRY
    if (!model || !d || !out) throw std::invalid_argument("null argument");
    std::unique_ptr<raptor_problem> pr(new raptor_problem());
    Problem& p = pr->p;
    p.model = model->m;
    const int nv = p.model.nv;
    if (d->traj.type != RAPTOR_TRAJ_BEZIER) throw std::invalid_argument("single-step optimizers use BezierCurves");
    p.traj.init(&d->traj);
    if (d->extra_variables < 0) throw std::invalid_argument("extra_variables < 0");
    p.traj.varLength += d->extra_variables;  // TalosSingleStepOptimizer.cpp:50-51
    p.traj.pq_pz.assign(p.traj.N, MatX(p.traj.Nact, p.traj.varLength));
    p.traj.pq_d_pz = p.traj.pq_pz;
    p.traj.pq_dd_pz = p.traj.pq_pz;
    if (d->num_dependent != 6) throw std::invalid_argument("contact constraint: six dependent joints expected");
    if (d->num_independent != p.traj.Nact || d->num_dependent + d->num_independent != nv)
        throw std::invalid_argument("Trajectory and DynamicsConstraints must have the same number of actuated joints");
    ContactDynamicsConstraints& dc = p.dc;
    dc.model = &p.model;
    dc.fk.model = &p.model;
    dc.nv = nv;
    dc.nd = d->num_dependent;
    dc.ni = d->num_independent;
    dc.dep.assign(d->dependent_ids, d->dependent_ids + dc.nd);
    dc.ind.assign(d->independent_ids, d->independent_ids + dc.ni);
    std::vector<int> seen(nv, 0);
    for (int id : dc.dep) { if (id < 0 || id >= nv || seen[id]++) throw std::invalid_argument("bad dependent joint id"); }
    for (int id : dc.ind) { if (id < 0 || id >= nv || seen[id]++) throw std::invalid_argument("bad independent joint id"); }
    dc.fk_start = d->fk_start_joint;
    dc.contact_joint_id = d->contact_joint_id;
    if (dc.contact_joint_id < 1 || dc.contact_joint_id > nv || dc.fk_start < 0 || dc.fk_start > nv)
        throw std::runtime_error("Can not find joint");
    dc.stance_foot_endT = Transform(Vec3(d->stance_foot_endT_p[0], d->stance_foot_endT_p[1], d->stance_foot_endT_p[2]));
    Mat3 Rd;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Rd(r, c) = d->stance_foot_T_des_R[3 * r + c];
    dc.stance_foot_T_des = Transform(Rd, Vec3(d->stance_foot_T_des_p[0], d->stance_foot_T_des_p[1], d->stance_foot_T_des_p[2]));
    p.cid.model = &p.model;
    p.cid.traj = &p.traj;
    p.cid.dc = &p.dc;
    VecX tl(d->torque_limits_lower, d->torque_limits_lower + dc.ni), tu(d->torque_limits_upper, d->torque_limits_upper + dc.ni);
    VecX jl(d->joint_limits_lower, d->joint_limits_lower + nv), ju(d->joint_limits_upper, d->joint_limits_upper + nv);
    p.cons.emplace_back(new ConstrainedTorqueLimits(&p.cid, tl, tu));                                   // TalosSingleStepOptimizer.cpp:69-73
    p.cons.emplace_back(new ConstrainedJointLimits(&p.cid, jl, ju));                                    // :82-86
    p.cons.emplace_back(new RectangleSurfaceContactConstraints(&p.cid, d->mu, d->gamma, d->Lx, d->Ly)); // :89-92
    if (d->gait.enabled) {  // :95-99
        if (d->stance_leg != 'L' && d->stance_leg != 'R') throw std::runtime_error("Invalid stance leg input");
        if (d->gait.swing_joint_id < 1 || d->gait.swing_joint_id > nv) throw std::runtime_error("Can not find joint");
        p.cons.emplace_back(new CustomizedConstraints(
            &p.cid, false, d->stance_leg == 'L', d->gait.swing_joint_id,
            Transform(Vec3(d->gait.swingfoot_endT_p[0], d->gait.swingfoot_endT_p[1], d->gait.swingfoot_endT_p[2])), d->gait));
    }
    if (d->period.enabled) {  // :101-106
        const raptor_period_desc& pd = d->period;
        if (pd.swing_contact_joint_id < 1 || pd.swing_contact_joint_id > nv) throw std::runtime_error("Can not find joint");
        if (!pd.joint_id1 || !pd.joint_id2) throw std::invalid_argument("null argument");
        std::vector<int> id1(pd.joint_id1, pd.joint_id1 + dc.ni), id2(pd.joint_id2, pd.joint_id2 + dc.ni);
        for (int k = 0; k < dc.ni; k++)
            if (id1[k] < 0 || id1[k] >= nv || id2[k] < 0 || id2[k] >= nv) throw std::runtime_error("Can not find joint");
        std::unique_ptr<Constraints> pc(new PeriodicityConstraints(
            &p.cid, d->mu, d->gamma, d->Lx, d->Ly, pd.mirror_sign, id1, id2, pd.swing_contact_joint_id,
            Transform(Vec3(pd.swing_foot_endT_p[0], pd.swing_foot_endT_p[1], pd.swing_foot_endT_p[2]))));
        if (pd.enabled == RAPTOR_PERIOD_LINK_ONLY) p.link = std::move(pc); else p.cons.push_back(std::move(pc));
    }
    p.finish();
    *out = pr.release();
    ORACLE_CATCH
}

int oracle_digit_problem_create(const raptor_model* model, const raptor_digit_desc* d, raptor_problem** out) {
    ORACLE_TRY
    if (!model || !d || !out) throw std::invalid_argument("null argument");
    std::unique_ptr<raptor_problem> pr(new raptor_problem());
    Problem& p = pr->p;
    p.model = model->m;
    const int nv = p.model.nv;
    if (nv != 36) throw std::invalid_argument("Digit: NUM_JOINTS is 36");
    if (d->traj.type != RAPTOR_TRAJ_BEZIER) throw std::invalid_argument("single-step optimizers use BezierCurves");
    p.traj.init(&d->traj);
    if (p.traj.Nact != 12) throw std::invalid_argument("Digit: NUM_INDEPENDENT_JOINTS is 12");
    if (d->extra_variables < 0) throw std::invalid_argument("extra_variables < 0");
    p.traj.varLength += d->extra_variables;  // DigitSingleStepOptimizer.cpp:50-51
    p.traj.pq_pz.assign(p.traj.N, MatX(p.traj.Nact, p.traj.varLength));
    p.traj.pq_d_pz = p.traj.pq_pz;
    p.traj.pq_dd_pz = p.traj.pq_pz;
    if (d->stance_leg != 'L' && d->stance_leg != 'R' && d->stance_leg != 'l' && d->stance_leg != 'r')
        throw std::runtime_error("Invalid stance leg input");
    const int sleg = (d->stance_leg == 'L' || d->stance_leg == 'l') ? 0 : 1;
    ContactDynamicsConstraints& dc = p.dc;
    dc.model = &p.model;
    dc.fk.model = &p.model;
    dc.nv = nv;
    dc.nd = 24;
    dc.ni = 12;
    std::vector<int> seen(nv, 0);
    auto take = [&](int id) {
        if (id < 0 || id >= nv || seen[id]++) throw std::runtime_error("Can not find joint");
        return id;
    };
    for (int k = 0; k < 6; k++) dc.dep.push_back(take(d->base_ids[k]));  // DigitDynamicsConstraints.h:19-57 order
    for (int leg = 0; leg < 2; leg++)
        for (int k = 0; k < 9; k++) dc.dep.push_back(take(d->leg_ids[leg][DIGIT_LEG_DEPENDENT[k]]));
    for (int leg = 0; leg < 2; leg++)
        for (int k = 0; k < 6; k++) dc.ind.push_back(take(d->leg_ids[leg][DIGIT_LEG_INDEPENDENT[k]]));
    dc.fk_start = d->base_ids[5] + 1;
    dc.contact_joint_id = d->leg_ids[sleg][DG_TOE_ROLL] + 1;
    {
        double R[9], pp[3];
        digit_stance_foot_endT(sleg, R, pp);
        Mat3 Re;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Re(r, c) = R[3 * r + c];
        dc.stance_foot_endT = Transform(Re, Vec3(pp[0], pp[1], pp[2]));
    }
    Mat3 Rd;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Rd(r, c) = d->stance_foot_T_des_R[3 * r + c];
    dc.stance_foot_T_des = Transform(Rd, Vec3(d->stance_foot_T_des_p[0], d->stance_foot_T_des_p[1], d->stance_foot_T_des_p[2]));
    // closures in the row order of the compiled reference file (DigitDynamicsConstraints.cpp:579-596):
    // right toe-A, right toe-B, left toe-A, left toe-B, right knee, left knee
    const int order[6][2] = {{1, 0}, {1, 1}, {0, 0}, {0, 1}, {1, 2}, {0, 2}};
    for (int k = 0; k < 6; k++) {
        const int leg = order[k][0];
        const DigitLoop& L = DIGIT_LOOPS[order[k][1]];
        Closure cl;
        cl.start = d->leg_ids[leg][L.start] + 1;
        cl.endA = d->leg_ids[leg][L.endA] + 1;
        cl.endB = d->leg_ids[leg][L.endB] + 1;
        cl.endTA = Transform(Vec3(L.rod[0], L.rod[1], L.rod[2]));
        double an[3] = {L.anchor[leg][0], L.anchor[leg][1], L.anchor[leg][2]};
        if (order[k][1] == 2) digit_knee_anchor(leg, an);
        cl.endTB = Transform(Vec3(an[0], an[1], an[2]));
        dc.closures.push_back(cl);
    }
    for (int leg = 0; leg < 2; leg++) {
        for (int k = 0; k < 6; k++) {
            const DigitToeGuess& g = DIGIT_TOE_GUESS[leg][k];
            LoopGuess lg;
            lg.nin = 2;
            lg.in_joint[0] = d->leg_ids[leg][DG_TOE_A];
            lg.in_joint[1] = d->leg_ids[leg][DG_TOE_B];
            lg.out_joint = d->leg_ids[leg][DIGIT_LOOPS[0].depj[k]];
            lg.sign = g.sign; lg.c0 = g.c0;
            lg.lin.assign(g.lin, g.lin + 8);
            lg.pair.assign(g.pair, g.pair + DIGIT_TOE_NPAIR);
            dc.guesses.push_back(lg);
        }
        for (int k = 0; k < 3; k++) {
            const DigitKneeGuess& g = DIGIT_KNEE_GUESS[leg][k];
            LoopGuess lg;
            lg.nin = 1;
            lg.in_joint[0] = d->leg_ids[leg][DG_KNEE];
            lg.out_joint = d->leg_ids[leg][DIGIT_LOOPS[2].depj[k]];
            lg.sign = g.sign; lg.c0 = g.c0;
            lg.lin.assign(g.lin, g.lin + 4);
            dc.guesses.push_back(lg);
        }
    }
    p.cid.model = &p.model;
    p.cid.traj = &p.traj;
    p.cid.dc = &p.dc;
    VecX tl(d->torque_limits_lower, d->torque_limits_lower + 12), tu(d->torque_limits_upper, d->torque_limits_upper + 12);
    VecX jl(d->joint_limits_lower, d->joint_limits_lower + nv), ju(d->joint_limits_upper, d->joint_limits_upper + nv);
    p.cons.emplace_back(new ConstrainedTorqueLimits(&p.cid, tl, tu));                                   // DigitSingleStepOptimizer.cpp:74-79
    p.cons.emplace_back(new ConstrainedJointLimits(&p.cid, jl, ju));                                    // :82-87
    p.cons.emplace_back(new RectangleSurfaceContactConstraints(&p.cid, d->mu, d->gamma, d->Lx, d->Ly)); // :90-94
    if (d->gait.enabled) {  // :97-101
        double R[9], pp[3];
        digit_stance_foot_endT(1 - sleg, R, pp);  // the swing foot's end transform (DigitCustomizedConstraints.cpp:16-24)
        Mat3 Re;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Re(r, c) = R[3 * r + c];
        p.cons.emplace_back(new CustomizedConstraints(&p.cid, true, sleg == 0, d->leg_ids[1 - sleg][DG_TOE_ROLL] + 1,
                                                      Transform(Re, Vec3(pp[0], pp[1], pp[2])), d->gait));
    }
    if (d->period.enabled) {  // :102-106; JOINT_MAP of DigitSingleStepPeriodicityConstraints.h:19-32
        double R[9], pp[3];
        digit_stance_foot_endT(1 - sleg, R, pp);  // DigitDynamicsConstraints.cpp:81-95 after the swap
        Mat3 Re;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Re(r, c) = R[3 * r + c];
        std::vector<int> id1, id2;
        for (int leg = 0; leg < 2; leg++)
            for (int k = 0; k < 6; k++) {
                id1.push_back(d->leg_ids[leg][DIGIT_LEG_INDEPENDENT[k]]);
                id2.push_back(d->leg_ids[1 - leg][DIGIT_LEG_INDEPENDENT[k]]);
            }
        std::unique_ptr<Constraints> pc(new PeriodicityConstraints(&p.cid, d->mu, d->gamma, d->Lx, d->Ly, 1.0, id1, id2,
                                                                   d->leg_ids[1 - sleg][DG_TOE_ROLL] + 1,
                                                                   Transform(Re, Vec3(pp[0], pp[1], pp[2]))));
        if (d->period.enabled == RAPTOR_PERIOD_LINK_ONLY) p.link = std::move(pc); else p.cons.push_back(std::move(pc));
    }
    p.finish();
    *out = pr.release();
    ORACLE_CATCH
}

int oracle_constrained_id_eval_host(raptor_problem* pr, const double* Z, int B, double* q, double* v, double* a,
                                    double* tau, double* lambda, double* ptau_pz, double* plambda_pz) {
    ORACLE_TRY
    if (!pr || !Z) throw std::invalid_argument("null argument");
    Problem& p = pr->p;
    if (p.cid.dc == nullptr) throw std::runtime_error("dcPtr_ is not defined yet!");
    const int nv = p.model.nv, N = p.traj.N, n = p.n, ni = p.dc.ni, nd = p.dc.nd;
    const bool deriv = ptau_pz != nullptr || plambda_pz != nullptr;
    for (int b = 0; b < B; b++) {
        p.cid.compute(Z + (size_t)b * n, deriv);
        for (int i = 0; i < N; i++) {
            const size_t e = (size_t)b * N + i;
            for (int j = 0; j < nv; j++) {
                if (q) q[e * nv + j] = p.cid.q[i][j];
                if (v) v[e * nv + j] = p.cid.v[i][j];
                if (a) a[e * nv + j] = p.cid.a[i][j];
            }
            for (int j = 0; j < ni; j++) {
                if (tau) tau[e * ni + j] = p.cid.tau[i][j];
                if (ptau_pz)
                    for (int c = 0; c < n; c++) ptau_pz[(e * ni + j) * n + c] = p.cid.ptau_pz[i](j, c);
            }
            for (int j = 0; j < 6; j++) {  // lambda.tail(6): the stance-foot contact wrench
                if (lambda) lambda[e * 6 + j] = p.cid.lambda[i][nd - 6 + j];
                if (plambda_pz)
                    for (int c = 0; c < n; c++) plambda_pz[(e * 6 + j) * n + c] = p.cid.plambda_pz[i](nd - 6 + j, c);
            }
        }
    }
    ORACLE_CATCH
}

// lambda(i) / plambda_pz(i) with all n_dep rows (ConstrainedInverseDynamics.cpp:154, 169-171)
int oracle_constrained_id_lambda_host(raptor_problem* pr, const double* Z, int B, double* lambda, double* plambda_pz) {
    ORACLE_TRY
    if (!pr || !Z) throw std::invalid_argument("null argument");
    Problem& p = pr->p;
    if (p.cid.dc == nullptr) throw std::runtime_error("dcPtr_ is not defined yet!");
    const int N = p.traj.N, n = p.n, nd = p.dc.nd;
    for (int b = 0; b < B; b++) {
        p.cid.compute(Z + (size_t)b * n, plambda_pz != nullptr);
        for (int i = 0; i < N; i++) {
            const size_t e = (size_t)b * N + i;
            for (int j = 0; j < nd; j++) {
                if (lambda) lambda[e * nd + j] = p.cid.lambda[i][j];
                if (plambda_pz)
                    for (int c = 0; c < n; c++) plambda_pz[(e * nd + j) * n + c] = p.cid.plambda_pz[i](j, c);
            }
        }
    }
    ORACLE_CATCH
}

// Costs/src/{MinimizePower,MinimizeTorque,MinimizeInitialVelocity,MinimizeInitialAcceleration}.cpp
static void cost_compute(Problem& p, int kind, const double* z, bool deriv, double& f, std::vector<double>& grad) {
    const int N = p.traj.N, V = p.traj.varLength, Nact = p.traj.Nact;
    f = 0;
    grad.assign(V, 0.0);
    if (kind == RAPTOR_COST_MINIMIZE_POWER || kind == RAPTOR_COST_MINIMIZE_TORQUE) {
        if (p.cid.dc == nullptr) throw std::invalid_argument("costs: the problem has no ConstrainedInverseDynamics");
        p.cid.compute(z, deriv);
        for (int i = 0; i < N; i++) {
            const VecX& tau = p.cid.tau[i];
            if (kind == RAPTOR_COST_MINIMIZE_POWER) {  // MinimizePower.cpp:32-54
                const VecX& v = p.traj.q_d[i];
                VecX power(Nact);
                double power_squared = 0;
                for (int j = 0; j < Nact; j++) {
                    power[j] = tau[j] * v[j];
                    power_squared += power[j] * power[j];
                }
                if (power_squared > 1e-8) {  // Power::SQUARE_ROOT_THRESHOLD
                    const double power_norm = std::sqrt(power_squared);
                    f += power_norm;
                    if (deriv)
                        for (int c = 0; c < V; c++) {
                            double s = 0;
                            for (int j = 0; j < Nact; j++)
                                s += (tau[j] * p.traj.pq_d_pz[i](j, c) + v[j] * p.cid.ptau_pz[i](j, c)) * power[j];
                            grad[c] += s / power_norm;
                        }
                }
            } else {  // MinimizeTorque.cpp:32-50
                double tau_squared = 0;
                for (int j = 0; j < Nact; j++) tau_squared += tau[j] * tau[j];
                const double tau_norm = std::sqrt(tau_squared);
                f += tau_norm;
                if (tau_norm > 1e-8 && deriv)
                    for (int c = 0; c < V; c++) {
                        double s = 0;
                        for (int j = 0; j < Nact; j++) s += p.cid.ptau_pz[i](j, c) * tau[j];
                        grad[c] += s / tau_norm;
                    }
            }
        }
        f /= N;
        for (double& g : grad) g /= N;
    } else if (kind == RAPTOR_COST_MINIMIZE_INITIAL_VELOCITY || kind == RAPTOR_COST_MINIMIZE_INITIAL_ACCELERATION) {
        p.traj.compute(z, deriv);  // MinimizeInitialVelocity.cpp:18-30
        const bool vel = kind == RAPTOR_COST_MINIMIZE_INITIAL_VELOCITY;
        const VecX& v0 = vel ? p.traj.q_d[0] : p.traj.q_dd[0];
        double sq = 0;
        for (int j = 0; j < Nact; j++) sq += v0[j] * v0[j];
        const double nrm = std::sqrt(sq);
        f = nrm;
        if (nrm > 1e-8 && deriv) {
            const MatX& P = vel ? p.traj.pq_d_pz[0] : p.traj.pq_dd_pz[0];
            for (int c = 0; c < V; c++) {
                double s = 0;
                for (int j = 0; j < Nact; j++) s += P(j, c) * v0[j];
                grad[c] = s / nrm;
            }
        }
    } else if (kind == RAPTOR_COST_MINIMIZE_PATH_LENGTH || kind == RAPTOR_COST_MINIMIZE_JERK) {
        // MinimizePathLength.cpp:10-61 on q, MinimizeJerk.cpp:10-61 on q_dd: sum of 0.5 |x(i+1) - x(i)|^2 over the samples, / N
        p.traj.compute(z, deriv);
        const bool jerk = kind == RAPTOR_COST_MINIMIZE_JERK;
        for (int i = 0; i < N - 1; i++) {
            const VecX& x1 = jerk ? p.traj.q_dd[i] : p.traj.q[i];
            const VecX& x2 = jerk ? p.traj.q_dd[i + 1] : p.traj.q[i + 1];
            VecX diff(Nact);
            double dd = 0;
            for (int j = 0; j < Nact; j++) {
                diff[j] = x2[j] - x1[j];
                dd += diff[j] * diff[j];
            }
            f += 0.5 * dd;
            if (deriv) {
                const MatX& P1 = jerk ? p.traj.pq_dd_pz[i] : p.traj.pq_pz[i];
                const MatX& P2 = jerk ? p.traj.pq_dd_pz[i + 1] : p.traj.pq_pz[i + 1];
                for (int c = 0; c < V; c++) {
                    double s = 0;
                    for (int j = 0; j < Nact; j++) s += diff[j] * (P2(j, c) - P1(j, c));
                    grad[c] += s;
                }
            }
        }
        f /= N;
        for (double& g : grad) g /= N;
    } else {
        throw std::invalid_argument("unknown cost kind");
    }
}

int oracle_eval_costs_host(raptor_problem* pr, const double* Z, int B, int num_costs, const int* kinds, double* F, double* GRAD) {
    ORACLE_TRY
    if (!pr || (!Z && B > 0) || !kinds || !F) throw std::invalid_argument("null argument");
    if (B < 0 || num_costs < 0) throw std::invalid_argument("negative size");
    Problem& p = pr->p;
    std::vector<double> grad;
    for (int b = 0; b < B; b++)
        for (int k = 0; k < num_costs; k++) {
            double f;
            cost_compute(p, kinds[k], Z + (size_t)b * p.n, GRAD != nullptr, f, grad);
            F[(size_t)b * num_costs + k] = f;
            if (GRAD)
                for (int c = 0; c < p.n; c++) GRAD[((size_t)b * num_costs + k) * p.n + c] = grad[c];
        }
    ORACLE_CATCH
}
// Costs::hess_f of MinimizePathLength / MinimizeJerk (MinimizePathLength.cpp:43-52, 56-60; MinimizeJerk.cpp twin): the
// pq_pz_pz terms of :45-51 are zero matrices for every trajectory class here, so hess_f does not depend on z.
int oracle_cost_hessian_host(raptor_problem* pr, int kind, double* HESS) {
    ORACLE_TRY
    if (!pr || !HESS) throw std::invalid_argument("null argument");
    if (kind != RAPTOR_COST_MINIMIZE_PATH_LENGTH && kind != RAPTOR_COST_MINIMIZE_JERK)
        throw std::runtime_error("cost Hessian: MinimizePathLength / MinimizeJerk only");
    Problem& p = pr->p;
    const int N = p.traj.N, V = p.traj.varLength;
    std::vector<double> z(p.n, 0.0);
    p.traj.compute(z.data(), true);
    MatX hess(V, V);
    for (int i = 0; i < N - 1; i++) {
        const MatX& P1 = (kind == RAPTOR_COST_MINIMIZE_JERK) ? p.traj.pq_dd_pz[i] : p.traj.pq_pz[i];
        const MatX& P2 = (kind == RAPTOR_COST_MINIMIZE_JERK) ? p.traj.pq_dd_pz[i + 1] : p.traj.pq_pz[i + 1];
        const MatX d = P2 - P1;
        hess = hess + d.transpose() * d;  // :43
    }
    size_t it = 0;
    for (int a = 0; a < V; a++)
        for (int b = a; b < V; b++) HESS[it++] = hess(a, b) / N;  // :59
    ORACLE_CATCH
}

int oracle_eval_costs(raptor_problem* pr, const double* Z, int B, int num_costs, const int* kinds, double* F, double* GRAD, void*) {
    return oracle_eval_costs_host(pr, Z, B, num_costs, kinds, F, GRAD);
}

void oracle_problem_destroy(raptor_problem* p) { delete p; }

int oracle_problem_layout(const raptor_problem* pr, int* n, int* m, int* num_blocks, int* block_offsets) {
    ORACLE_TRY
    if (!pr) throw std::invalid_argument("null problem");
    if (n) *n = pr->p.n;
    if (m) *m = pr->p.m;
    if (num_blocks) *num_blocks = (int)pr->p.cons.size();
    if (block_offsets)
        for (size_t i = 0; i < pr->p.offsets.size(); i++) block_offsets[i] = pr->p.offsets[i];
    ORACLE_CATCH
}

int oracle_problem_bounds(const raptor_problem* pr_, double* g_lb, double* g_ub) {
    ORACLE_TRY
    raptor_problem* pr = const_cast<raptor_problem*>(pr_);
    if (!pr) throw std::invalid_argument("null problem");
    int it = 0;
    for (auto& c : pr->p.cons) {
        c->compute_bounds();
        for (int i = 0; i < c->m; i++, it++) {
            if (g_lb) g_lb[it] = c->g_lb[i];
            if (g_ub) g_ub[it] = c->g_ub[i];
        }
    }
    ORACLE_CATCH
}

int oracle_problem_tspan(const raptor_problem* pr, double* tspan) {
    ORACLE_TRY
    if (!pr || !tspan) throw std::invalid_argument("null argument");
    for (int i = 0; i < pr->p.traj.N; i++) tspan[i] = pr->p.traj.tspan[i];
    ORACLE_CATCH
}

// Optimizer::eval_g / eval_jac_g packing (Optimization/src/Optimizer.cpp:361-385, 467-486), looped over B.
int oracle_eval_batch_host(raptor_problem* pr, const double* Z, int B, unsigned flags, double* G, double* JAC) {
    ORACLE_TRY
    if (!pr || (!Z && B > 0)) throw std::invalid_argument("null argument");
    if (B < 0) throw std::invalid_argument("B < 0");
    Problem& p = pr->p;
    const bool want_jac = (flags & RAPTOR_EVAL_JAC) != 0;
    const bool want_g = (flags & RAPTOR_EVAL_G) != 0;
    if (want_jac && !JAC) throw std::invalid_argument("JAC buffer is null");
    if (want_g && !G) throw std::invalid_argument("G buffer is null");
    for (int b = 0; b < B; b++) {
        const double* z = Z + (size_t)b * p.n;
        // fresh result storage per z: the reference objects would have been constructed once and
        // carry state only through is_computed(); each z here is a distinct evaluation.
        size_t it = 0, itj = 0;
        for (auto& c : p.cons) {
            std::fill(c->g.begin(), c->g.end(), 0.0);
            c->pg_pz.setZero();
            c->compute(z, want_jac);
            if (want_g)
                for (int i = 0; i < c->m; i++) G[(size_t)b * p.m + it++] = c->g[i];
            if (want_jac)
                for (int i = 0; i < c->m; i++)
                    for (int j = 0; j < p.n; j++) JAC[((size_t)b * p.m) * p.n + itj++] = c->pg_pz(i, j);
        }
    }
    ORACLE_CATCH
}
int oracle_eval_batch(raptor_problem* pr, const double* Z, int B, unsigned flags, double* G, double* JAC, void*) {
    return oracle_eval_batch_host(pr, Z, B, flags, G, JAC);
}

// raptor_eval_batch_status: the same evaluation, z by z, with the report of the dependent-joint root finding
// (DigitDynamicsConstraints.cpp:344-367: at most 50 iterations, the result accepted whatever the solver's state).
int oracle_eval_batch_status_host(raptor_problem* pr, const double* Z, int B, unsigned flags, double* G, double* JAC, int* status,
                                  int* iterations) {
    ORACLE_TRY
    if (!pr || (B > 0 && (!Z || !status))) throw std::invalid_argument("null argument");
    if (B < 0) throw std::invalid_argument("B < 0");
    Problem& p = pr->p;
    for (int b = 0; b < B; b++) {
        p.dc.stat_flags = 0;
        p.dc.stat_max_iter = 0;
        const int rc = oracle_eval_batch_host(pr, Z + (size_t)b * p.n, 1, flags, G ? G + (size_t)b * p.m : nullptr,
                                              JAC ? JAC + (size_t)b * p.m * p.n : nullptr);
        if (rc == RAPTOR_ERR_RUNTIME && p.dc.stat_flags != 0) {
            // a loop Jacobian that is singular at the accepted point: the reference only asserts there
            // (DigitDynamicsConstraints.cpp:392-393); reported, the rows of this z are left as they are
            status[b] = p.dc.stat_flags | RAPTOR_STATUS_NONFINITE;
            if (iterations) iterations[b] = p.dc.stat_max_iter;
            continue;
        }
        if (rc != RAPTOR_OK) return rc;
        status[b] = p.dc.stat_flags;
        if (iterations) iterations[b] = p.dc.stat_max_iter;
    }
    ORACLE_CATCH
}
int oracle_eval_batch_status(raptor_problem* pr, const double* Z, int B, unsigned flags, double* G, double* JAC, int* status,
                             int* iterations, void*) {
    return oracle_eval_batch_status_host(pr, Z, B, flags, G, JAC, status, iterations);
}

// KinovaOptimizer::set_parameters, q_des_input / tplan_n_input (KinovaOptimizer.cpp:28-29, 39-40, 112-113)
int oracle_kinova_set_objective(raptor_problem* pr, const double* q_des, int tplan_n) {
    ORACLE_TRY
    if (!pr || !q_des) throw std::invalid_argument("null argument");
    Problem& p = pr->p;
    if (!p.is_kinova) throw std::invalid_argument("objective: not a KinovaOptimizer problem");
    if (tplan_n < 0 || tplan_n >= p.traj.N) throw std::invalid_argument("tplan_n must be in [0, N)");
    p.q_des.assign(q_des, q_des + p.traj.Nact);
    p.tplan_n = tplan_n;
    ORACLE_CATCH
}

// KinovaOptimizer::eval_f / eval_grad_f (KinovaOptimizer.cpp:152-218)
int oracle_kinova_eval_objective_host(raptor_problem* pr, const double* Z, int B, double* F, double* GRAD) {
    ORACLE_TRY
    if (!pr || (B > 0 && (!Z || !F))) throw std::invalid_argument("null argument");
    if (B < 0) throw std::invalid_argument("B < 0");
    Problem& p = pr->p;
    if (!p.is_kinova || p.tplan_n < 0) throw std::invalid_argument("objective: raptor_kinova_set_objective has not been called");
    const int n = p.n;
    for (int b = 0; b < B; b++) {
        p.traj.compute(Z + (size_t)b * n, GRAD != nullptr);
        const VecX& qplan = p.traj.q[p.tplan_n];
        double obj = 0;
        const int order[7] = {0, 2, 4, 6, 1, 3, 5};  // :169-175, continuous joints first
        for (int t = 0; t < 7; t++) {
            const int i = order[t];
            const double d = (i % 2 == 0) ? wrapToPi(qplan[i] - p.q_des[i]) : (qplan[i] - p.q_des[i]);
            obj += std::pow(d, 2);
        }
        F[b] = 0.5 * obj;  // :177
        if (GRAD) {
            const MatX& pqplan_pz = p.traj.pq_pz[p.tplan_n];
            for (int i = 0; i < n; i++) {  // :206-215
                const double qdiff = (i % 2 == 0) ? wrapToPi(qplan[i] - p.q_des[i]) : (qplan[i] - p.q_des[i]);
                GRAD[(size_t)b * n + i] = qdiff * pqplan_pz(i, i);
            }
        }
    }
    ORACLE_CATCH
}
int oracle_kinova_eval_objective(raptor_problem* pr, const double* Z, int B, double* F, double* GRAD, void*) {
    return oracle_kinova_eval_objective_host(pr, Z, B, F, GRAD);
}

// Optimizer::eval_h (Optimization/src/Optimizer.cpp:499-635) with KinovaOptimizer::eval_hess_f (KinovaOptimizer.cpp:223-250):
// H[b] = upper triangle, row by row (:531-538, :621-627), of obj_factor * hess_f + sum_r lambda_r * pg_pz_pz(r).
int oracle_eval_h_batch_host(raptor_problem* pr, const double* Z, int B, const double* obj_factor, const double* lambda, double* H) {
    ORACLE_TRY
    if (!pr || (B > 0 && (!Z || !lambda || !H))) throw std::invalid_argument("null argument");
    if (B < 0) throw std::invalid_argument("B < 0");
    Problem& p = pr->p;
    if (!p.is_kinova) throw std::runtime_error("eval_h: enable_hessian is false for this optimizer");  // :513-515 (only KinovaOptimizer sets it)
    if (obj_factor && p.tplan_n < 0) throw std::invalid_argument("eval_h: raptor_kinova_set_objective has not been called");
    const int n = p.n, N = p.traj.N, Nact = p.traj.Nact;
    auto* kc = dynamic_cast<KinovaCustomizedConstraints*>(p.cons[3].get());
    InverseDynamicsHessian idh;
    for (int b = 0; b < B; b++) {
        const double* z = Z + (size_t)b * n;
        const double* lam = lambda + (size_t)b * p.m;
        MatX hessian(n, n);
        const double of = obj_factor ? obj_factor[b] : 0.0;
        if (of != 0) {  // :552-563, KinovaOptimizer.cpp:243-247
            p.traj.compute(z, true);
            const MatX& pqplan_pz = p.traj.pq_pz[p.tplan_n];
            for (int i = 0; i < n; i++) hessian(i, i) += of * (pqplan_pz(i, i) * pqplan_pz(i, i));
        }
        // [0] JointLimits, [1] VelocityLimits: pg_pz_pz = pq_pz_pz / pq_d_pz_pz = zero matrices (JointLimits.cpp:40-44)
        // [2] TorqueLimits: pg_pz_pz(i * Nact + j) = ptau_pz_pz(j, i)  (TorqueLimits.cpp:36-40)
        bool any = false;
        for (int r = p.offsets[2]; r < p.offsets[3]; r++) any = any || (lam[r] != 0);  // :568-578
        if (any) {
            idh.compute(p.id, z);
            for (int i = 0; i < N; i++)
                for (int j = 0; j < Nact; j++) hessian = hessian + idh.ptau_pz_pz[i][j] * lam[p.offsets[2] + i * Nact + j];  // :616
        }
        // [3] KinovaCustomizedConstraints
        any = false;
        for (int r = p.offsets[3]; r < p.offsets[4]; r++) any = any || (lam[r] != 0);
        if (any && kc->box.numObstacles > 0) {  // KinovaCustomizedConstraints.cpp:80-82: pg_pz_pz keeps its zero initialisation
            p.traj.compute(z, true);
            for (int i = 0; i < N; i++)
                for (int j = 0; j < kc->num_spheres; j++) {
                    const double l = lam[p.offsets[3] + i * kc->num_spheres + j];
                    hessian = hessian + kinova_sphere_hessian(*kc, i, j) * l;
                }
        }
        size_t it = (size_t)b * (n * (n + 1) / 2);
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) H[it++] = hessian(i, j);  // :621-627
    }
    ORACLE_CATCH
}
int oracle_eval_h_batch(raptor_problem* pr, const double* Z, int B, const double* obj_factor, const double* lambda, double* H, void*) {
    return oracle_eval_h_batch_host(pr, Z, B, obj_factor, lambda, H);
}

// oracle-only component view: InverseDynamics::ptau_pz_pz (InverseDynamics.cpp:141-171), [B][N][nv][n][n] -- what the reference's
// Hessian test reads (Tests/KinematicsDynamics/TestInverseDynamics.cpp:48-79)
int oracle_inverse_dynamics_hessian_host(raptor_problem* pr, const double* Z, int B, double* ptau_pz_pz) {
    ORACLE_TRY
    if (!pr || !Z || !ptau_pz_pz) throw std::invalid_argument("null argument");
    Problem& p = pr->p;
    const int nv = p.model.nv, N = p.traj.N, n = p.n;
    InverseDynamicsHessian idh;
    for (int b = 0; b < B; b++) {
        idh.compute(p.id, Z + (size_t)b * n);
        for (int i = 0; i < N; i++)
            for (int j = 0; j < nv; j++)
                for (int x = 0; x < n; x++)
                    for (int y = 0; y < n; y++)
                        ptau_pz_pz[((((size_t)b * N + i) * nv + j) * n + x) * n + y] = idh.ptau_pz_pz[i][j](x, y);
    }
    ORACLE_CATCH
}

int oracle_trajectory_eval_host(raptor_problem* pr, const double* Z, int B, double* q, double* q_d, double* q_dd) {
    ORACLE_TRY
    if (!pr || !Z) throw std::invalid_argument("null argument");
    Trajectory& t = pr->p.traj;
    for (int b = 0; b < B; b++) {
        t.compute(Z + (size_t)b * pr->p.n, false);
        for (int i = 0; i < t.N; i++)
            for (int j = 0; j < t.Nact; j++) {
                const size_t o = ((size_t)b * t.N + i) * t.Nact + j;
                if (q) q[o] = t.q[i][j];
                if (q_d) q_d[o] = t.q_d[i][j];
                if (q_dd) q_dd[o] = t.q_dd[i][j];
            }
    }
    ORACLE_CATCH
}

int oracle_inverse_dynamics_eval_host(raptor_problem* pr, const double* Z, int B, double* tau, double* ptau_pz) {
    ORACLE_TRY
    if (!pr || !Z) throw std::invalid_argument("null argument");
    Problem& p = pr->p;
    const int nv = p.model.nv, N = p.traj.N, n = p.n;
    for (int b = 0; b < B; b++) {
        p.id.compute(Z + (size_t)b * n, ptau_pz != nullptr);
        for (int i = 0; i < N; i++)
            for (int j = 0; j < nv; j++) {
                if (tau) tau[((size_t)b * N + i) * nv + j] = p.id.tau[i][j];
                if (ptau_pz)
                    for (int c = 0; c < n; c++) ptau_pz[(((size_t)b * N + i) * nv + j) * n + c] = p.id.ptau_pz[i](j, c);
            }
    }
    ORACLE_CATCH
}

int oracle_capsule_distance_batch_host(const double* P, const double* dP, const double* R, int n_pairs, int F,
                                       double* dist, double* grad) {
    ORACLE_TRY
    if (!P || !R || !dist) throw std::invalid_argument("null argument");
    if (F < 0 || n_pairs < 0) throw std::invalid_argument("negative size");
    for (int i = 0; i < n_pairs; i++)
        dist[i] = capsule_distance(P + (size_t)i * 12, dP ? dP + (size_t)i * 12 * F : nullptr, R + (size_t)i * 4, F,
                                   (dP && grad) ? grad + (size_t)i * F : nullptr);
    ORACLE_CATCH
}
int oracle_capsule_distance_batch(const double* P, const double* dP, const double* R, int n_pairs, int F, double* dist,
                                  double* grad, void*) {
    return oracle_capsule_distance_batch_host(P, dP, R, n_pairs, F, dist, grad);
}

// *MultipleStepPeriodicityConstraints::compute for the link between step s (pr_curr, created with its reset-map block) and step
// s + 1 (pr_next): g [B][mp], pg_pz [B][mp][n_curr], and the rows of pg3_pz2 / pg4_pz2 [B][2 Nact][n_next]
int oracle_step_link_eval_batch_host(raptor_problem* pr_curr, raptor_problem* pr_next, const double* Zc, const double* Zn, int B,
                                     unsigned flags, double* G, double* JAC, double* JAC2) {
    ORACLE_TRY
    if (!pr_curr || !pr_next || (B > 0 && (!Zc || !Zn))) throw std::invalid_argument("null argument");
    Problem &pc = pr_curr->p, &pn = pr_next->p;
    PeriodicityConstraints* pk = pc.link ? dynamic_cast<PeriodicityConstraints*>(pc.link.get())
                                         : (pc.cons.empty() ? nullptr : dynamic_cast<PeriodicityConstraints*>(pc.cons.back().get()));
    if (!pk) throw std::invalid_argument("step link: the current step's problem must be created with its reset-map block");
    if (pc.traj.varLength != pn.traj.varLength || pc.traj.Nact != pn.traj.Nact)
        throw std::invalid_argument("currTrajPtr_ and nextTrajPtr_ should have the same varLength");
    const bool jac = (flags & RAPTOR_EVAL_JAC) != 0;
    if (jac && (!JAC || !JAC2)) throw std::invalid_argument("JAC buffer is null");
    if ((flags & RAPTOR_EVAL_G) && !G) throw std::invalid_argument("G buffer is null");
    const int mp = pk->m, n = pc.traj.varLength, n2 = pn.traj.varLength, ni = pc.traj.Nact;
    for (int b = 0; b < B; b++) {
        pk->next_traj = &pn.traj;
        pk->next_z = Zn + (size_t)b * n2;
        try {
            pk->compute(Zc + (size_t)b * n, jac);
        } catch (...) {
            pk->next_traj = nullptr;
            throw;
        }
        pk->next_traj = nullptr;
        if (flags & RAPTOR_EVAL_G)
            for (int r = 0; r < mp; r++) G[(size_t)b * mp + r] = pk->g[r];
        if (jac) {
            for (int r = 0; r < mp; r++)
                for (int c = 0; c < n; c++) JAC[((size_t)b * mp + r) * n + c] = pk->pg_pz(r, c);
            for (int r = 0; r < ni; r++)
                for (int c = 0; c < n2; c++) {
                    JAC2[((size_t)b * 2 * ni + r) * n2 + c] = pk->pg3_pz2(r, c);
                    JAC2[((size_t)b * 2 * ni + ni + r) * n2 + c] = pk->pg4_pz2(r, c);
                }
        }
    }
    ORACLE_CATCH
}
int oracle_step_link_eval_batch(raptor_problem* pr_curr, raptor_problem* pr_next, const double* Zc, const double* Zn, int B,
                                unsigned flags, double* G, double* JAC, double* JAC2, void*) {
    return oracle_step_link_eval_batch_host(pr_curr, pr_next, Zc, Zn, B, flags, G, JAC, JAC2);
}

// Arm-arm rows of DualKinovaOptimizer::eval_g / eval_jac_g
// (Examples/Kinova/CollisionAvoidanceTrajectory/src/DualKinovaOptimizer.cpp:269-305, 449-510)
int oracle_dual_kinova_arm_arm_batch_host(raptor_problem* pr1, raptor_problem* pr2, const double* Z, int B, unsigned flags,
                                          double* G, double* JAC);
int oracle_dual_kinova_arm_arm_batch(raptor_problem* pr1, raptor_problem* pr2, const double* Z, int B, unsigned flags, double* G,
                                     double* JAC, void*) {
    return oracle_dual_kinova_arm_arm_batch_host(pr1, pr2, Z, B, flags, G, JAC);
}
int oracle_dual_kinova_arm_arm_batch_host(raptor_problem* pr1, raptor_problem* pr2, const double* Z, int B, unsigned flags,
                                          double* G, double* JAC) {
    ORACLE_TRY
    if (!pr1 || !pr2 || (B > 0 && !Z)) throw std::invalid_argument("null argument");
    Problem &p1 = pr1->p, &p2 = pr2->p;
    KinovaCustomizedConstraints* k1 = p1.cons.empty() ? nullptr : dynamic_cast<KinovaCustomizedConstraints*>(p1.cons.back().get());
    KinovaCustomizedConstraints* k2 = p2.cons.empty() ? nullptr : dynamic_cast<KinovaCustomizedConstraints*>(p2.cons.back().get());
    if (!k1 || !k2) throw std::invalid_argument("dual Kinova: both problems must end with KinovaCustomizedConstraints");
    if (p1.traj.N != p2.traj.N) throw std::invalid_argument("dual Kinova: the two trajectories must have the same N");
    const bool jac = (flags & RAPTOR_EVAL_JAC) != 0;
    if (jac && !JAC) throw std::invalid_argument("JAC buffer is null");
    if ((flags & RAPTOR_EVAL_G) && !G) throw std::invalid_argument("G buffer is null");
    const int N = p1.traj.N, n1 = p1.traj.varLength, n2 = p2.traj.varLength, nvars = n1 + n2, rows = N * 4;
    for (int b = 0; b < B; b++) {
        const double* z = Z + (size_t)b * nvars;
        p1.traj.compute(z, jac);
        p2.traj.compute(z + n1, jac);
        for (int i = 0; i < N; i++) {
            int j = 0;
            for (int j1 = 1; j1 < 3; j1++)
                for (int j2 = 1; j2 < 3; j2++) {
                    const int ids[4] = {k1->tapered_capsules[j1].first, k1->tapered_capsules[j1].second,
                                        k2->tapered_capsules[j2].first, k2->tapered_capsules[j2].second};
                    double P[12], R[4];
                    std::vector<double> dP(jac ? (size_t)12 * nvars : 0, 0.0);  // dk_tc*_sphere_*: zero outside the own robot's columns
                    for (int e = 0; e < 4; e++) {
                        KinovaCustomizedConstraints* k = e < 2 ? k1 : k2;
                        Vec3 c;
                        MatX pc;
                        k->sphere_center(i, ids[e], jac, c, pc);
                        for (int x = 0; x < 3; x++) P[e * 3 + x] = c(x);
                        R[e] = k->sphere_radius[ids[e]];
                        if (jac)
                            for (int x = 0; x < 3; x++)
                                for (int cc = 0; cc < (e < 2 ? n1 : n2); cc++) dP[(size_t)(e * 3 + x) * nvars + (e < 2 ? 0 : n1) + cc] = pc(x, cc);
                    }
                    const size_t row = (size_t)b * rows + i * 4 + j;
                    const double dist = capsule_distance(P, jac ? dP.data() : nullptr, R, nvars, jac ? JAC + row * nvars : nullptr);
                    if (flags & RAPTOR_EVAL_G) G[row] = dist;
                    j++;
                }
        }
    }
    ORACLE_CATCH
}

int oracle_init(int) { return RAPTOR_OK; }
int oracle_profile_begin(raptor_problem*, int) { return RAPTOR_OK; }
int oracle_profile_end(raptor_problem*, int* n, double* a, double* b) {
    if (n) *n = 0;
    if (a) *a = 0;
    if (b) *b = 0;
    return RAPTOR_OK;
}

// ---------------- oracle-only probes used to pin the oracle itself (tests/test_oracle_*.py) ----------------
// pinocchio::rnea restatement: tau (armature included, no damping/friction) and accumulated link wrenches f[nv][6]
int oracle_rnea(const raptor_model* model, const double* q, const double* v, const double* a, double* tau, double* f) {
    ORACLE_TRY
    std::vector<Vec6> ff;
    featherstone_rnea(model->m, q, v, a, nullptr, nullptr, nullptr, tau, nullptr, &ff);
    if (f)
        for (int i = 0; i < model->m.nv; i++)
            for (int k = 0; k < 6; k++) f[i * 6 + k] = ff[i](k);
    ORACLE_CATCH
}
// computeRNEADerivatives restatement: dtau/dq, dtau/dv, M (row-major nv x nv)
int oracle_rnea_derivatives(const raptor_model* model, const double* q, const double* v, const double* a, double* tau,
                            double* dq, double* dv, double* da) {
    ORACLE_TRY
    MatX Q, V, A;
    rnea_derivatives(model->m, q, v, a, tau, Q, V, A);
    const int nv = model->m.nv;
    for (int i = 0; i < nv * nv; i++) {
        dq[i] = Q.a[i];
        dv[i] = V.a[i];
        da[i] = A.a[i];
    }
    ORACLE_CATCH
}
// ForwardKinematicsSolver::compute(start,end,q,nullptr,&Transform(offset),order<=1): T.R (9), T.p (3), translation Jacobian (3 x nv)
int oracle_fk(const raptor_model* model, int start, int end, const double* q, const double* offset, double* R, double* p, double* Jp) {
    ORACLE_TRY
    ForwardKinematicsSolver fk;
    fk.model = &model->m;
    Transform endT(offset ? Vec3(offset[0], offset[1], offset[2]) : Vec3());
    fk.compute(start, end, q, nullptr, &endT, Jp ? 1 : 0);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            if (R) R[i * 3 + j] = fk.T.R(i, j);
        if (p) p[i] = fk.T.p(i);
    }
    if (Jp) {
        MatX J = fk.getTranslationJacobian();
        for (size_t i = 0; i < J.a.size(); i++) Jp[i] = J.a[i];
    }
    ORACLE_CATCH
}
// BoxCollisionAvoidance::computeDistance(point, ppoint_pz): min distance, argmin, gradient row of the argmin
int oracle_box_distance(int O, const double* centers, const double* rpy, const double* size, const double* point,
                        const double* ppoint_pz, int W, double* min_dist, int* min_index, double* grad) {
    ORACLE_TRY
    BoxCollisionAvoidance box;
    box.init(O, centers, rpy, size);
    Vec3 pt(point[0], point[1], point[2]);
    if (ppoint_pz) {
        MatX pp(3, W);
        for (int i = 0; i < 3 * W; i++) pp.a[i] = ppoint_pz[i];
        box.computeDistance(pt, pp);
        if (grad)
            for (int c = 0; c < W; c++) grad[c] = box.pdistances_pz(box.minimumDistanceIndex, c);
    } else {
        box.computeDistance(pt);
    }
    *min_dist = box.minimumDistance;
    *min_index = box.minimumDistanceIndex;
    ORACLE_CATCH
}

}  // extern "C"
