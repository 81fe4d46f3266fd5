// gfx950 kernels of the hot path.  Included once by raptor_hip.hip.
//
//   K_id   (id_kernel)          one lane per (decision vector b, time sample i, column c of z):
//                               trajectory -> world-frame RNEA -> directional derivative along column c
//                               -> TorqueLimits rows (+ JointLimits / VelocityLimits rows).
//   K_kinova (kinova_kernel.h)  the whole Kinova evaluation fused, 8 lanes per sample-eval (analytic derivatives).
//   K_traj (traj_kernel)        q, q_d, q_dd only (component view for the parity tests).
//
// Why a column per lane.  d(tau)/dz is needed for every column of z.  For the trajectories on this path a
// column touches ONE joint (ArmourBezierCurves.cpp:147-151: diagonal; BezierCurves.cpp:165-169: block row),
// so the column's contribution is a single directional derivative of RNEA along
// (dq, dv, da) = (w0, w1, w2) * e_joint.  That replaces the reference's three dense nv x nv Jacobians and
// three GEMMs (InverseDynamics.cpp:106-139) by one forward-mode sweep per lane, with no intermediate
// matrices in memory; consecutive lanes hold consecutive columns, so the stores of one Jacobian row are
// contiguous.  fp64 throughout.
#ifndef RAPTOR_KERNELS_H
#define RAPTOR_KERNELS_H

#include "../../include/raptor_hip.h"
#include "device_types.h"
#include "spatial_dev.h"

#define RAPTOR_BLOCK 256

// Utils::sign (Utils/include/Utils.h:85-95)
RD double sign_eps(double v) { return v > 1e-8 ? 1.0 : (v < -1e-8 ? -1.0 : 0.0); }

// ---------------------------------------------------------------------------------------------
// trajectory: q, q_d, q_dd of joint j at sample i   (coefficients^T * B, BezierCurves.cpp:143-145)
// ---------------------------------------------------------------------------------------------
RD void traj_joint(const DevTraj* __restrict__ T, const double* __restrict__ basis, const double* __restrict__ z,
                   int i, int j, double& q, double& qd, double& qdd) {
    const int K = T->K;
    const double* B0 = basis + (size_t)(i * 3) * K;
    const double* B1 = B0 + K;
    const double* B2 = B1 + K;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (T->type == RAPTOR_TRAJ_ARMOUR_BEZIER) {
        const double qf = T->armour_q0[j] + z[j];  // ArmourBezierCurves.cpp:96-99
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double ck = (k < 3) ? T->armour_c[k][j] : qf;
            s0 += ck * B0[k];
            s1 += ck * B1[k];
            s2 += ck * B2[k];
        }
    } else if (T->type == RAPTOR_TRAJ_PIECEWISE_BEZIER) {
        // PiecewiseBezierCurves is linear in z and column c moves joint c % Nact only: dense per-sample column weights
        // W[i][order][c] (pq_pz / pq_d_pz / pq_dd_pz entries, PiecewiseBezierCurves.cpp:184-243) + the q0 / qT contribution
        const int V = T->varLength, Nact = T->Nact;
        const double* W0 = basis + (size_t)(i * 3) * V;
        const double* W1 = W0 + V;
        const double* W2 = W1 + V;
        for (int c = j; c < V; c += Nact) {
            const double zc = z[c];
            s0 += W0[c] * zc;
            s1 += W1[c] * zc;
            s2 += W2[c] * zc;
        }
        const double* fx = basis + (size_t)T->N * 3 * V + (size_t)(i * 3) * Nact + j;
        s0 += fx[0];
        s1 += fx[Nact];
        s2 += fx[2 * Nact];
    } else {
        const double* zj = z + j * K;  // column-major reshape (Utils.h:142-146)
        for (int k = 0; k < K; k++) {
            const double ck = zj[k];
            s0 += ck * B0[k];
            s1 += ck * B1[k];
            s2 += ck * B2[k];
        }
        if (T->nfixed > 0) {  // rows pinned by the constrained initial state (BezierCurves.cpp:84-96): table after the basis
            const double* fx = basis + (size_t)T->N * 3 * K + (size_t)(i * 3) * T->Nact + j;
            s0 += fx[0];
            s1 += fx[T->Nact];
            s2 += fx[2 * T->Nact];
        }
    }
    q = s0;
    qd = s1;
    qdd = s2;
}

// which joint column c of z drives, and with what weights (pq_pz, pq_d_pz, pq_dd_pz entries)
RD void column_direction(const DevTraj* __restrict__ T, const double* __restrict__ basis,
                         const double* __restrict__ wsum, int i, int c, int& jd, double& w0, double& w1, double& w2) {
    if (T->type == RAPTOR_TRAJ_ARMOUR_BEZIER) {
        jd = c;
        w0 = wsum[i * 3];
        w1 = wsum[i * 3 + 1];
        w2 = wsum[i * 3 + 2];
    } else if (T->type == RAPTOR_TRAJ_PIECEWISE_BEZIER) {
        const int V = T->varLength;
        jd = c % T->Nact;
        const double* W0 = basis + (size_t)(i * 3) * V;
        w0 = W0[c];
        w1 = W0[V + c];
        w2 = W0[2 * V + c];
    } else {
        const int K = T->K;
        jd = c / K;
        const int k = c - jd * K;
        const double* B0 = basis + (size_t)(i * 3) * K;
        w0 = B0[k];
        w1 = B0[K + k];
        w2 = B0[2 * K + k];
    }
}

// ---------------------------------------------------------------------------------------------
// K_id : serial-chain models (parent[i] == i-1).
//
// Lane = (sample-eval e, column c).  The lanes of one e are consecutive (flat index w = e*V + c), so a
// 256-lane block covers at most 256/V + 2 sample-evals.  Base quantities that are identical for all
// columns of e -- the world-frame joint axis s_l, the link wrench f_l and the joint-space extras
// (armature/damping/friction) -- are kept ONCE per e in LDS ("link-recursive sweeps staged through LDS"):
// every lane of e computes and stores the same values (no barrier needed: a lane only ever reads back
// what it, or a lane holding bit-identical values, has written), and reads them back as broadcasts.
// Per-lane state is only what depends on the column: the running directional velocity/acceleration and
// NV torque-derivative accumulators.  The link loop is rolled, so constants stream through SGPRs
// instead of being hoisted into (and spilled from) the register file.
//
//   forward, link l:   s_l, v_l, a_l, I_l, f_l             (base, world frame)
//                      Delta = s_k * w0 once l == k        (virtual displacement of the subtree of joint k)
//                      ds_l, dv_l, da_l, df_l              (directional derivative along column c)
//                      dtau_j += s_j . df_l   for j <= l   (immediate accumulation, replaces a df array)
//   backward, link l:  F_l = sum_{i>=l} f_i ;  tau_l = s_l . F_l + extras
//                      dtau_l += (Delta x s_l) . F_l  for l > k
// ---------------------------------------------------------------------------------------------
#define RAPTOR_ID_SLOT 13  // doubles per link in an LDS slot: s (6), f (6), extras (1)

RD SV lds_sv(const double* p) { return SV{v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5])}; }
RD void lds_put(double* p, SV x) {
    p[0] = x.a.x; p[1] = x.a.y; p[2] = x.a.z;
    p[3] = x.l.x; p[4] = x.l.y; p[5] = x.l.z;
}

template <int NV, bool DERIV>
__global__ __launch_bounds__(RAPTOR_BLOCK, 2) void id_kernel(const DevModel* __restrict__ M, const DevTraj* __restrict__ T,
                                                           const double* __restrict__ basis,
                                                           const double* __restrict__ wsum, const DevLayout L,
                                                           const double* __restrict__ Z, const long long total,
                                                           double* __restrict__ G, double* __restrict__ JAC) {
    extern __shared__ double lds[];
    const int V = L.n;
    const int lanes_per_e = DERIV ? V : 1;
    const long long wblock = (long long)blockIdx.x * blockDim.x;
    const long long w = wblock + threadIdx.x;
    if (w >= total) return;
    const int N = T->N;
    const long long e = w / lanes_per_e;
    const int c = DERIV ? (int)(w - e * lanes_per_e) : 0;
    const int i = (int)(e % N);
    const long long b = e / N;
    const double* z = Z + b * V;
    double* slot = lds + (size_t)(e - wblock / lanes_per_e) * (NV * RAPTOR_ID_SLOT);
    const size_t rowbase = (size_t)b * L.m;
    const bool lead = (c == 0) && (G != nullptr);

    int jd = -1;
    double w0 = 0.0, w1 = 0.0, w2 = 0.0;
    if (DERIV) column_direction(T, basis, wsum, i, c, jd, w0, w1, w2);

    M3 Rw;
#pragma unroll
    for (int k = 0; k < 9; k++) Rw.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
    V3 ow = v3(0, 0, 0);
    SV v = sv_zero();
    SV a = SV{v3(0, 0, 0), v3(M->neg_gravity[0], M->neg_gravity[1], M->neg_gravity[2])};  // -a_grav
    SV dv = sv_zero(), da = sv_zero(), Delta = sv_zero();
    double dtau[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) dtau[j] = 0.0;

#pragma unroll 1
    for (int l = 0; l < NV; l++) {
        double q, qd, qdd;
        traj_joint(T, basis, z, i, l, q, qd, qdd);
        const M3 A = mul(Rw, M->plR[l]);
        V3 o = ow + mul(Rw, v3(M->plp[l][0], M->plp[l][1], M->plp[l][2]));
        double sq, cq;
        sincos(q, &sq, &cq);
        M3 R;
        SV s;
        joint_world(M->jtype[l], q, cq, sq, A, R, o, s);

        const SV sqd = s * qd;
        const SV vi = v + sqd;
        const SV ai = a + s * qdd + crm(v, sqd);
        Inertia I;
        I.mass = M->mass[l];
        I.c = o + mul(R, v3(M->lever[l][0], M->lever[l][1], M->lever[l][2]));
        rotate_inertia(R, M->Ic[l], I);
        const SV h = apply(I, vi);
        const SV g = apply(I, ai);
        const SV f = g + crf(vi, h);
        double* sl = slot + l * RAPTOR_ID_SLOT;
        lds_put(sl, s);
        lds_put(sl + 6, f);
        // InverseDynamics.cpp:115-121 (damping, friction); armature is inside pinocchio::rnea (>= 3)
        sl[12] = M->armature[l] * qdd + M->damping[l] * qd + M->friction[l] * sign_eps(qd);

        // JointLimits / VelocityLimits rows (JointLimits.cpp:32-45, VelocityLimits.cpp:32-44)
        if (L.off_q >= 0) {
            const size_t row = rowbase + L.off_q + (size_t)i * NV + l;
            if (DERIV) JAC[row * V + c] = (l == jd) ? w0 : 0.0;
            if (lead) G[row] = q;
        }
        if (L.off_qd >= 0) {
            const size_t row = rowbase + L.off_qd + (size_t)i * NV + l;
            if (DERIV) JAC[row * V + c] = (l == jd) ? w1 : 0.0;
            if (lead) G[row] = qd;
        }

        if (DERIV) {
            const bool isk = (l == jd);
            if (isk) Delta = s * w0;
            const SV ds = crm(Delta, s);  // zero up to and including the driven joint
            const double dqd = isk ? w1 : 0.0, dqdd = isk ? w2 : 0.0;
            const SV dsqd = ds * qd + s * dqd;
            const SV dvi = dv + dsqd;
            const SV dai = da + ds * qdd + s * dqdd + crm(dv, sqd) + crm(v, dsqd);
            const SV dpv = dvi - crm(Delta, vi);  // derivative as seen in the moving body
            const SV dpa = dai - crm(Delta, ai);
            const SV df = crf(Delta, g) + apply(I, dpa) + crf(dvi, h) + crf(vi, crf(Delta, h) + apply(I, dpv));
            dv = dvi;
            da = dai;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                if (j <= l) {
                    dtau[j] += dot(lds_sv(slot + j * RAPTOR_ID_SLOT), df);
                    if (j == l && isk) dtau[j] += M->armature[l] * w2 + M->damping[l] * w1;
                }
            }
        }
        v = vi;
        a = ai;
        Rw = R;
        ow = o;
    }

    // ---- backward sweep: suffix sums of wrenches, no transforms needed in the world frame ----
    SV F = sv_zero();
#pragma unroll
    for (int l = NV - 1; l >= 0; l--) {
        const double* sl = slot + l * RAPTOR_ID_SLOT;
        const SV s = lds_sv(sl);
        F = F + lds_sv(sl + 6);
        const size_t row = rowbase + L.off_tau + (size_t)i * NV + l;
        if (DERIV) {
            double d = dtau[l];
            if (l > jd) d += dot(crm(Delta, s), F);
            JAC[row * V + c] = d;
        }
        if (lead) G[row] = dot(s, F) + sl[12];
    }
}

// ---------------------------------------------------------------------------------------------
// K_traj : Trajectories::compute values only
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RAPTOR_BLOCK) void traj_kernel(const DevTraj* __restrict__ T, const double* __restrict__ basis,
                                                             const double* __restrict__ Z, const long long total,
                                                             double* __restrict__ q, double* __restrict__ qd,
                                                             double* __restrict__ qdd) {
    const long long w = (long long)blockIdx.x * RAPTOR_BLOCK + threadIdx.x;
    if (w >= total) return;
    const int Nact = T->Nact;
    const int j = (int)(w % Nact);
    const long long e = w / Nact;
    const int i = (int)(e % T->N);
    const long long b = e / T->N;
    double a0, a1, a2;
    traj_joint(T, basis, Z + b * T->varLength, i, j, a0, a1, a2);
    if (q) q[w] = a0;
    if (qd) qd[w] = a1;
    if (qdd) qdd[w] = a2;
}

#endif
