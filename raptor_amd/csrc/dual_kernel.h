// Arm-arm collision rows of DualKinovaOptimizer (Examples/Kinova/CollisionAvoidanceTrajectory/src/DualKinovaOptimizer.cpp:
// 269-305 eval_g, 449-510 eval_jac_g): per sample four tapered-capsule pairs (capsules 1, 2 of robot 1 against capsules 1, 2 of
// robot 2), each capsule spanned by two collision spheres of KinovaCustomizedConstraints (sphere_centers_copy /
// sphere_centers_gradient_copy, KinovaCustomizedConstraints.cpp:109,114), gradient over the numVars1 + numVars2 columns.
//
// One lane per (z pair, sample, capsule pair): both arms' forward kinematics in registers, the four sphere centres and their
// 3 x 14 Jacobians (zero outside the own robot's columns, as the reference's dk_tc*_sphere_* temporaries) go to a staging area in
// the layout capsule_kernel reads; the distance / gradient arithmetic itself stays in capsule_kernel (capsule_kernel.h).
#ifndef RAPTOR_DUAL_KERNEL_H
#define RAPTOR_DUAL_KERNEL_H

#include "kernels.h"

struct DualCapsules {
    int ids[2][2][2];  // [robot][capsule 1, 2][first, second] sphere index
};

// centres of two spheres of one arm and their Jacobians w.r.t. that arm's V decision variables: column c moves joint c mod 7 with
// weight pq_pz(c mod 7, c) -- ArmourBezierCurves (V = 7, w0 * identity) and PiecewiseBezierCurves (the reference's
// DualKinovaOptimizer members are KinovaLongerHorizonOptimizers: DualKinovaOptimizer.cpp:41-42)
RD void kinova_two_spheres(const DevModel* __restrict__ M, const DevTraj* __restrict__ T, const double* __restrict__ basis,
                           const double* __restrict__ wsum, const DevSpheres* __restrict__ S, const double* __restrict__ z,
                           const int i, const int sa, const int sb, double* __restrict__ Pout, double* __restrict__ dPout,
                           const int F, const int col0, const int zero0, const int zeroV, double* __restrict__ Rout) {
    constexpr int NV = 7;
    const int ja = S->joint_id[sa], jb = S->joint_id[sb];  // Pinocchio joint ids (1-based)
    M3 Rw;
#pragma unroll
    for (int k = 0; k < 9; k++) Rw.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
    V3 ow = v3(0, 0, 0);
    SV sj[NV];
    V3 ca = v3(0, 0, 0), cb = v3(0, 0, 0);
#pragma unroll
    for (int l = 0; l < NV; l++) {
        double q, qd, qdd, sq, cq;
        traj_joint(T, basis, z, i, l, q, qd, qdd);
        sincos(q, &sq, &cq);
        const M3 A = mul(Rw, M->plR[l]);
        V3 o = ow + mul(Rw, v3(M->plp[l][0], M->plp[l][1], M->plp[l][2]));
        M3 R;
        joint_world(M->jtype[l], q, cq, sq, A, R, o, sj[l]);
        Rw = R;
        ow = o;
        if (l + 1 == ja) ca = ow + mul(Rw, v3(S->offset[sa][0], S->offset[sa][1], S->offset[sa][2]));
        if (l + 1 == jb) cb = ow + mul(Rw, v3(S->offset[sb][0], S->offset[sb][1], S->offset[sb][2]));
    }
    Pout[0] = ca.x; Pout[1] = ca.y; Pout[2] = ca.z;
    Pout[3] = cb.x; Pout[4] = cb.y; Pout[5] = cb.z;
    Rout[0] = S->radius[sa];
    Rout[1] = S->radius[sb];
    if (dPout == nullptr) return;
    const int V = T->varLength;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        // d centre / d q_j = omega_j x (centre - o_j) = s.a x centre + s.l  (world-origin joint axis), for joints up to the sphere's link
        const V3 dja = (j < ja) ? (cross(sj[j].a, ca) + sj[j].l) : v3(0, 0, 0);
        const V3 djb = (j < jb) ? (cross(sj[j].a, cb) + sj[j].l) : v3(0, 0, 0);
        for (int c = j; c < V; c += NV) {
            int jd;
            double w0, w1, w2;
            column_direction(T, basis, wsum, i, c, jd, w0, w1, w2);
            dPout[0 * F + col0 + c] = dja.x * w0; dPout[1 * F + col0 + c] = dja.y * w0; dPout[2 * F + col0 + c] = dja.z * w0;
            dPout[3 * F + col0 + c] = djb.x * w0; dPout[4 * F + col0 + c] = djb.y * w0; dPout[5 * F + col0 + c] = djb.z * w0;
        }
    }
    // the other robot's columns
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < zeroV; c++) dPout[r * F + zero0 + c] = 0.0;
}

__global__ __launch_bounds__(64) void dual_kinova_points_kernel(
    const DevModel* __restrict__ M1, const DevTraj* __restrict__ T1, const double* __restrict__ basis1,
    const double* __restrict__ wsum1, const DevSpheres* __restrict__ S1, const DevModel* __restrict__ M2,
    const DevTraj* __restrict__ T2, const double* __restrict__ basis2, const double* __restrict__ wsum2,
    const DevSpheres* __restrict__ S2, const DualCapsules caps, const double* __restrict__ Z, const long long pair0,
    const int npairs, double* __restrict__ P, double* __restrict__ dP, double* __restrict__ R) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= npairs) return;
    const long long gp = pair0 + idx;
    const int N = T1->N;
    const int j = (int)(gp & 3);
    const long long e = gp >> 2;
    const int i = (int)(e % N);
    const long long b = e / N;
    const int j1 = j >> 1, j2 = j & 1;  // j counts (j1, j2) with j2 fastest (DualKinovaOptimizer.cpp:283-285)
    const int n1 = T1->varLength, n2 = T2->varLength, F = n1 + n2;
    const double* z = Z + b * F;
    double* Pp = P + (size_t)idx * 12;
    double* dp = dP ? dP + (size_t)idx * 12 * F : nullptr;
    double* Rp = R + (size_t)idx * 4;
    kinova_two_spheres(M1, T1, basis1, wsum1, S1, z, i, caps.ids[0][j1][0], caps.ids[0][j1][1], Pp, dp, F, 0, n1, n2, Rp);
    kinova_two_spheres(M2, T2, basis2, wsum2, S2, z + n1, i, caps.ids[1][j2][0], caps.ids[1][j2][1], Pp + 6, dp ? dp + 6 * F : nullptr,
                       F, n1, 0, n1, Rp + 2);
}

#endif
