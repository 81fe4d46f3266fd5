// K_sphere_rows : KinovaCustomizedConstraints rows (sphere vs oriented boxes) for ANY trajectory whose columns move one joint
// each -- used by the KinovaLongerHorizonOptimizer path (PiecewiseBezierCurves: 21 * degree (+14) columns), where the fused
// Kinova kernel's "column = joint" layout does not apply.
//   KinovaCustomizedConstraints::compute   Examples/Kinova/CollisionAvoidanceTrajectory/src/KinovaCustomizedConstraints.cpp:73-158
//   BoxCollisionAvoidance::computeDistance Constraints/src/BoxCollisionAvoidance.cpp:92-204, 312-373
//
// 32 lanes per sample-eval, 8 sample-evals per workgroup.  Lanes 0..6 of a group evaluate the trajectory and sincos of their
// joint (LDS exchange inside the half-wave); lane sp < ns walks the chain to its sphere's joint, finds the arg-min box (the
// Jacobian kernel's code and tie rules) and leaves the distance's gradient per JOINT, d dist / d q_c = s_c . [p x n ; n], in
// LDS.  The reference's 3 x nv translation Jacobian times pq_pz (:112) is then one multiply per output entry,
//     pg_pz(row, c) = (d dist / d q_{joint(c)}) * pq_pz(joint(c), c),
// written by the whole workgroup as one flat, contiguous stream (the obstacle rows of consecutive samples are adjacent).
#ifndef RAPTOR_SPHERE_ROWS_KERNEL_H
#define RAPTOR_SPHERE_ROWS_KERNEL_H

#include "kinova_kernel.h"

#define SR_LPE 32   // lanes per sample-eval (>= RAPTOR_MAX_SPHERES)
#define SR_EPB 8    // sample-evals per workgroup
#define SR_NV 7

template <bool DERIV>
__global__ __launch_bounds__(RAPTOR_BLOCK) void sphere_rows_kernel(
    const DevModel* __restrict__ M, const DevTraj* __restrict__ T, const double* __restrict__ basis,
    const double* __restrict__ wsum, const DevSpheres* __restrict__ S, const DevBox* __restrict__ boxes, const DevLayout L,
    const double* __restrict__ Z, const long long E, double* __restrict__ G, double* __restrict__ JAC) {
    __shared__ double lx[SR_EPB][SR_NV][3];                       // cos q, sin q, q
    __shared__ double lg[SR_EPB][RAPTOR_MAX_SPHERES][SR_NV];      // d dist / d q_c per sphere
    const int tid = threadIdx.x;
    const int sp = tid & (SR_LPE - 1);
    const int el = tid >> 5;
    const long long e0 = (long long)blockIdx.x * SR_EPB;
    const long long e = e0 + el;
    const int N = T->N, V = L.n;
    const int ns = S->num_spheres, O = S->num_obstacles;
    const bool valid = e < E;
    const int i = valid ? (int)(e % N) : 0;
    const long long b = valid ? e / N : 0;
    const double* z = Z + b * V;

    if (valid && sp < SR_NV) {
        double q, qd, qdd, sq, cq;
        traj_joint(T, basis, z, i, sp, q, qd, qdd);
        sincos(q, &sq, &cq);
        lx[el][sp][0] = cq; lx[el][sp][1] = sq; lx[el][sp][2] = q;
    }
    wave_sync();

    if (valid && sp < ns) {
        const size_t row = (size_t)b * L.m + L.off_obs + (size_t)i * ns + sp;
        if (O == 0) {  // :80-82, g / pg_pz keep their zero initialisation
            if (G != nullptr) G[row] = 0.0;
#pragma unroll
            for (int c = 0; c < SR_NV; c++) lg[el][sp][c] = 0.0;
        } else {
            const int jid = S->joint_id[sp];
            M3 Rw;
#pragma unroll
            for (int k = 0; k < 9; k++) Rw.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
            V3 ow = v3(0, 0, 0);
            SV sv[SR_NV];
#pragma unroll
            for (int l = 0; l < SR_NV; l++) {
                sv[l] = sv_zero();
                if (l < jid) {
                    const M3 A = mul(Rw, M->plR[l]);
                    V3 o = ow + mul(Rw, v3(M->plp[l][0], M->plp[l][1], M->plp[l][2]));
                    M3 R;
                    joint_world(M->jtype[l], lx[el][l][2], lx[el][l][0], lx[el][l][1], A, R, o, sv[l]);
                    Rw = R;
                    ow = o;
                }
            }
            const V3 point = ow + mul(Rw, v3(S->offset[sp][0], S->offset[sp][1], S->offset[sp][2]));
            BoxBest best;
            best.key = 1e38; best.d2 = 0.0; best.sign = 1.0; best.dl = v3(0, 0, 0); best.idx = 0;
            for (int ob = 0; ob < O; ob++) box_candidate(boxes + ob, ob, point, best);
            if (G != nullptr) G[row] = box_best_distance(best) - S->radius[sp];  // :145
            if (DERIV) {
                const SV Wr = box_gradient_wrench(boxes + best.idx, best, point, 1.0);
#pragma unroll
                for (int c = 0; c < SR_NV; c++) lg[el][sp][c] = (c < jid) ? dot(sv[c], Wr) : 0.0;
            }
        }
    }
    if (!DERIV) return;
    __syncthreads();

    // ---- flat copy-out: rows (el, sp) x V columns; consecutive (el, sp) are consecutive rows unless b changes ----
    const long long eend = (e0 + SR_EPB < E) ? e0 + SR_EPB : E;
    const int nrows = (int)(eend - e0) * ns;
    const long long total = (long long)nrows * V;
    for (long long idx = tid; idx < total; idx += RAPTOR_BLOCK) {
        const int r = (int)(idx / V);
        const int c = (int)(idx - (long long)r * V);
        const int rel = r / ns;
        const int rsp = r - rel * ns;
        const long long re = e0 + rel;
        const int ri = (int)(re % N);
        const long long rb = re / N;
        int jd;
        double w0, w1, w2;
        column_direction(T, basis, wsum, ri, c, jd, w0, w1, w2);
        JAC[((size_t)rb * L.m + L.off_obs + (size_t)ri * ns + rsp) * V + c] = lg[rel][rsp][jd] * w0;
    }
}

// K_kinova_expand : pg_pz rows of the four Kinova blocks from the records of kinova_fused_kernel<true, RECORD> and the column
// weights of the trajectory (PiecewiseBezierCurves: W[i][order][c], column c moves joint c mod 7):
//     JointLimits row j:     [c mod 7 == j] W0[c]              VelocityLimits row j: [c mod 7 == j] W1[c]
//     TorqueLimits row j:    Dq[j][c mod 7] W0[c] + Dv[j][c mod 7] W1[c] + Da[j][c mod 7] W2[c]     (InverseDynamics.cpp:137-139)
//     obstacle row (sphere): gq[sphere][c mod 7] W0[c]                                                (KinovaCustomizedConstraints.cpp:112)
// One workgroup per `spb` consecutive samples of a decision vector: records + weights staged in LDS; each of the four row blocks
// of those samples is ONE contiguous piece of the output in Optimizer order (spb * rows * V doubles -- the longer the piece, the
// closer a streaming write gets to the HBM rate) and leaves as 16-byte stores (V even); no integer division in the
// loops.  The HBM-bound half of the longer-horizon path, like legged_expand_kernel.
__global__ __launch_bounds__(RAPTOR_BLOCK) void kinova_expand_kernel(const DevTraj* __restrict__ T, const double* __restrict__ basis,
                                                                      const DevLayout L, const int ns, const int spb,
                                                                      const double* __restrict__ REC, double* __restrict__ JAC) {
    extern __shared__ double le[];  // spb records (KREC_STRIDE each), then spb x (W0 | W1 | W2) (3 V each)
    const int tid = threadIdx.x;
    const int N = T->N, V = L.n;
    const int groups = (N + spb - 1) / spb;
    const long long b = blockIdx.x / groups;
    const int i0 = (int)(blockIdx.x - b * groups) * spb;
    const int cnt = min(spb, N - i0);
    const int rlen = KREC_G + 7 * ns;
    const size_t e0 = (size_t)b * N + i0;
    for (int s = 0; s < cnt; s++)
        for (int t = tid; t < rlen; t += RAPTOR_BLOCK) le[s * KREC_STRIDE + t] = REC[(e0 + s) * KREC_STRIDE + t];
    double* Wall = le + spb * KREC_STRIDE;
    for (int t = tid; t < cnt * 3 * V; t += RAPTOR_BLOCK) Wall[t] = basis[(size_t)(i0 * 3) * V + t];  // samples are adjacent in the table
    __syncthreads();
    const size_t rowbase = (size_t)b * L.m;
    auto value = [&](int region, int s, int r, int c, int jd) -> double {
        const double* W = Wall + s * 3 * V;
        const double* rec = le + s * KREC_STRIDE;
        if (region == 0) return (jd == r) ? W[c] : 0.0;
        if (region == 1) return (jd == r) ? W[V + c] : 0.0;
        if (region == 2) return rec[KREC_D + r * 7 + jd] * W[c] + rec[KREC_D + 49 + r * 7 + jd] * W[V + c] + rec[KREC_D + 98 + r * 7 + jd] * W[2 * V + c];
        return rec[KREC_G + r * 7 + jd] * W[c];
    };
    const bool wide = (V & 1) == 0 && (((size_t)JAC) & 15) == 0;
    const int cw = wide ? 2 : 1;              // columns per store
    const int per_row = V / cw;               // stores per row
    const float inv_per_row = 1.0f / (float)per_row;
#pragma unroll 1
    for (int region = 0; region < 4; region++) {
        const int rows = (region == 3) ? ns : 7;
        const int off = (region == 0) ? L.off_q : (region == 1) ? L.off_qd : (region == 2) ? L.off_tau : L.off_obs;
        double* dst = JAC + (rowbase + off + (size_t)i0 * rows) * V;
        const int len = cnt * rows * per_row;
        const float inv_rows = 1.0f / (float)rows;
        for (int t = tid; t < len; t += RAPTOR_BLOCK) {
            // exact for these small integers: the quotient's distance to the next integer is >= 0.5 / divisor
            const int ra = (int)(((float)t + 0.5f) * inv_per_row);  // row within the piece
            const int c = (t - ra * per_row) * cw;
            const int s = (int)(((float)ra + 0.5f) * inv_rows);
            const int r = ra - s * rows;
            const int q7 = (int)(((float)c + 0.5f) * (1.0f / 7.0f));
            const int jd = c - 7 * q7;
            if (wide) {
                const int jd1 = (jd == 6) ? 0 : jd + 1;
                reinterpret_cast<double2*>(dst)[t] = make_double2(value(region, s, r, c, jd), value(region, s, r, c + 1, jd1));
            } else {
                dst[t] = value(region, s, r, c, jd);
            }
        }
    }
}

// MinimizePathLength / MinimizeJerk (Costs/src/MinimizePathLength.cpp:10-61, MinimizeJerk.cpp:10-61) for any trajectory whose
// columns move one joint each:  f = (1/N) sum_i 0.5 |x(i+1) - x(i)|^2,  x = q or q_dd;
// grad_f(c) = (1/N) sum_i (x(i+1) - x(i))_{joint(c)} (w(i+1, c) - w(i, c)),  w = the column's pq_pz / pq_dd_pz entry.
// One workgroup per (decision vector, cost): the N x Nact values go to LDS once, lanes then own columns; fixed summation order.
__global__ __launch_bounds__(RAPTOR_BLOCK) void traj_cost_kernel(const DevTraj* __restrict__ T, const double* __restrict__ basis,
                                                                  const double* __restrict__ wsum, const double* __restrict__ Z,
                                                                  const int n, const int num_costs, const int* __restrict__ kinds,
                                                                  double* __restrict__ F, double* __restrict__ GRAD) {
    extern __shared__ double lx[];  // [N][Nact], then 256 partial sums
    const int tid = threadIdx.x;
    const long long b = blockIdx.x / num_costs;
    const int k = blockIdx.x - (int)(b * num_costs);
    const bool jerk = kinds[k] == RAPTOR_COST_MINIMIZE_JERK;
    // n >= V: the legged problems append variables the trajectory ignores (TalosSingleStepOptimizer.cpp:50-51)
    const int N = T->N, Nact = T->Nact;
    const int V = (T->type == RAPTOR_TRAJ_BEZIER || T->type == RAPTOR_TRAJ_POLYNOMIAL) ? T->K * Nact : T->varLength;
    const double* z = Z + b * n;
    for (int idx = tid; idx < N * Nact; idx += RAPTOR_BLOCK) {
        const int i = idx / Nact, j = idx - i * Nact;
        double q, qd, qdd;
        traj_joint(T, basis, z, i, j, q, qd, qdd);
        lx[idx] = jerk ? qdd : q;
    }
    __syncthreads();
    double* red = lx + N * Nact;
    double part = 0.0;
    for (int idx = tid; idx < (N - 1) * Nact; idx += RAPTOR_BLOCK) {
        const double d = lx[idx + Nact] - lx[idx];
        part += 0.5 * d * d;
    }
    red[tid] = part;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int t = 0; t < RAPTOR_BLOCK; t++) s += red[t];
        F[b * num_costs + k] = s / N;
    }
    if (GRAD == nullptr) return;
    for (int c = tid; c < n; c += RAPTOR_BLOCK) {
        if (c >= V) {
            GRAD[(b * num_costs + k) * n + c] = 0.0;
            continue;
        }
        int jd;
        double w0, w1, w2, p0, p1, p2;
        column_direction(T, basis, wsum, 0, c, jd, p0, p1, p2);
        double s = 0.0;
        for (int i = 0; i < N - 1; i++) {
            column_direction(T, basis, wsum, i + 1, c, jd, w0, w1, w2);
            s += (lx[(i + 1) * Nact + jd] - lx[i * Nact + jd]) * (jerk ? (w2 - p2) : (w0 - p0));
            p0 = w0;
            p2 = w2;
        }
        GRAD[(b * num_costs + k) * n + c] = s / N;
    }
}

#endif
