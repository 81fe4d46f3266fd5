// K_period : the reset-map ("periodicity") block of the single-step optimizers
// (Examples/Digit/src/DigitSingleStepPeriodicityConstraints.cpp:57-215, Examples/Talos/src/
// TalosSingleStepPeriodicityConstraints.cpp:57-215, Unitree-G1 twin):
//   g1 = H(q-) (v+ - v-) - J(q-)^T lambda      g2 = J(q-) v+      g3, g4 position / velocity reset      g5 contact on lambda
// with q-, v- the last sample of the step, J the constraint Jacobian after the stance swap, v+ and lambda taken from z.
//
// MI355X design -- NOT the reference's algorithm order.  The reference forms crba(q-), J, the tensors Jx_partial_dq and
// JTx_partial_dq, a gravity-free computeRNEADerivatives and multiplies by pq_pz(last), pv_pz(last) (nv x varLength).
// Only 12 directions matter: along actuator t the robot moves on the stance-phase manifold with joint rates
// Pfull[:, t] (identity rows + the P_dep already in the core kernel's record), so
//     pg1 = (A Pfull - H Vq) B_k(T) - (H Pfull) dB_k(T),     A = d(H dv - J^T lambda)/dq,  Vq = d v-/d q_t.
// A comes from per-joint 6-vectors in world-frame algebra (psi_k = a_parent x s_k, IS_k = I^C_k s_k, the contact wrench W of
// lambda.tail(6) and X_k = s_k x* W - d_k W, cq_k = s_k x* F_k + I^C_k psi_k + X_k), no tensors; Vq needs no J_dep^-1: the old
// stance foot stays planted, sum_base s_b Vq_b = -(sum over its path of (w_j x s_j) v-_j + s_j Vq_j), a 3 x 3 solve.
// One workgroup of four wavefronts per decision vector: wavefront 0 walks the model's own tree (lane = joint in DFS pre-order,
// composites by wave shuffles), wavefront 1 builds Pfull and Vq meanwhile, then all do the small products and stream the rows
// out densely.  numpy models: scripts/proto_period.py (directional and matrix forms, both agree with the oracle to 5e-14).
#ifndef RAPTOR_PERIOD_KERNEL_H
#define RAPTOR_PERIOD_KERNEL_H

#include "digit_kernel.h"
#include "legged_kernel.h"

#define PL_WAVES 4   // wavefronts per decision vector: 0 = tree phases, 1 = d v-/d q, all = the small products and the rows
#define PL_CUS 256            // compute units of the MI355X (staggered start of the first round)
#ifndef PER_STAGGER
#define PER_STAGGER 5         // Digit instance (three workgroups per CU)
#endif
#ifndef PER_STAGGER_SMALL
#define PER_STAGGER_SMALL 0   // Talos / G1 instance (four per CU)
#endif
#define PL_BLOCK (64 * PL_WAVES)
// LDS map of legged_period_kernel<NVM, NDM> (doubles): NVM / NDM = compiled bounds on nv / n_dep (18 / 6 for Talos and G1,
// 36 / 24 for Digit), so that the small robots keep several decision vectors per CU in flight
#define PL_HS (NVM + 1)                            // padded row stride of H and A (per-lane rows on distinct LDS banks)
#define PL_TS RAPTOR_LEG_MAX_IND                   // row stride of the [joint][direction] tables
#define PL_LOC 0                                   // [nv][16] local joint transform (phases 0-A); then [nv][12] psi, X per lane
#define PL_S (PL_LOC + NVM * 18)                   // [nv][6] world-frame joint axes, by lane
#define PL_D1 PL_LOC                               // [nv][ni] coefficient of B_k(T) in the g1 rows   } written after the tree
#define PL_D2 (PL_LOC + NVM * PL_TS)               // [nv][ni] coefficient of dB_k(T)                  } phases are over
#define PL_DV (PL_S + NVM * 6)                     // by lane: v+ - v-, v+, v-
#define PL_VP (PL_DV + NVM)
#define PL_VM (PL_VP + NVM)
#define PL_H (PL_VM + NVM)                         // [nv][HS] crba(q-) incl. armature, by model joint
#define PL_JA (PL_H + NVM * PL_HS)                 // [nd][nv] constraint Jacobian after the swap
#define PL_A (PL_JA + NDM * NVM)                   // [nv + 6][HS] d(H dv - J^T lambda)/dq, then the six rows d(J_foot v+)/dq
#define PL_ZEROED (NVM * PL_HS + NDM * NVM + (NVM + 6) * PL_HS)
#define PL_PF (PL_A + (NVM + 6) * PL_HS)           // [nv][TS] joint rates along the manifold directions (identity rows + P_dep)
#define PL_VQ (PL_PF + NVM * PL_TS)                // [nv][TS] d v- / d q_t at fixed actuator rates
#define PL_G2 (PL_VQ + NVM * PL_TS)                // [nd][ni] D_t (J v+)
#define PL_W (PL_G2 + NDM * PL_TS)                 // [4][16] B, dB of the first sample, B, dB of the last
#define PL_C5 (PL_W + 64)                          // [7][6] d g5 / d lambda.tail(6)
#define PL_E (PL_C5 + 42)                          // [3][3] E = d rpy / d omega at the contact frame; [3][3] A9_ij = m . T_ij;
#define PL_TMP (PL_E + 27)                         // [3][3] Bm = sum_i T_ij omega+_i   (T_ij = d_rpy_rates(R_c, e_i, e_j))
#define PL_PATH (PL_TMP + 27)                      // bytes: [nv][2][RAPTOR_LEG_MAX_DEPTH] lanes of this lane's path, their model joints
#define PL_IDT (PL_PATH + NVM * 2 * RAPTOR_LEG_MAX_DEPTH / 8)  // ints: [3][16] columns / joint that the reset rows g3, g4 touch
#define PL_TOTAL (PL_IDT + 24)
constexpr int period_lds_doubles(int NVM, int NDM) { return PL_TOTAL; }

// directional derivative of rpy_rates(R, w) with respect to R along dR = [ww]x R (w fixed): the second-order part of
// ForwardKinematicsSolver::getRPYHessian (ForwardKinematics.cpp:445-498)
RD V3 d_rpy_rates(const M3& R, V3 w, V3 ww) {
    const V3 c0 = col(R, 0), c1 = col(R, 1), c2 = col(R, 2);
    const V3 d0 = cross(w, c0), d1 = cross(w, c1), d2 = cross(w, c2);        // dR = [w]x R, by column
    const V3 v0 = cross(ww, c0), v1 = cross(ww, c1), v2 = cross(ww, c2);     // variation of R
    const V3 e0 = cross(w, v0), e1 = cross(w, v1), e2 = cross(w, v2);        // variation of dR
    // roll = atan2(-R12, R22): entries (1,2) = c2.y, (2,2) = c2.z
    const double D0 = c2.y * c2.y + c2.z * c2.z;
    const double N0 = c2.y * d2.z - c2.z * d2.y;
    const double vN0 = v2.y * d2.z + c2.y * e2.z - v2.z * d2.y - c2.z * e2.y;
    const double vD0 = 2.0 * (c2.y * v2.y + c2.z * v2.z);
    // yaw = atan2(-R01, R00): entries (0,0) = c0.x, (0,1) = c1.x
    const double D2 = c0.x * c0.x + c1.x * c1.x;
    const double N2 = c1.x * d0.x - c0.x * d1.x;
    const double vN2 = v1.x * d0.x + c1.x * e0.x - v0.x * d1.x - c0.x * e1.x;
    const double vD2 = 2.0 * (c0.x * v0.x + c1.x * v1.x);
    // pitch = asin(R02): entry (0,2) = c2.x
    const double x = c2.x;
    const double om = 1.0 - x * x;
    const double da = safe_dasin(x);
    const double dda = (x >= 1.0 || x <= -1.0) ? 0.0 : x / (om * sqrt(om));
    return v3((vN0 * D0 - N0 * vD0) / (D0 * D0), dda * v2.x * d2.x + da * e2.x, (vN2 * D2 - N2 * vD2) / (D2 * D2));
}

// Closed loops of Digit at q-: one lane per (decision vector, leg).  Writes, for its loop, the closure rows of J, J v+,
// J^T lambda and their derivatives along the loop's own drivers (the only directions that move the loop).
template <bool TOE>
RD void digit_period_loops_body(const DevTraj* __restrict__ T, const DevDigit* __restrict__ D, const DevPeriod* __restrict__ P,
                                const double* __restrict__ basis, const int n, const int nv, const int nd,
                                const double* __restrict__ Z, const long long total, const long long blk,
                                const double* __restrict__ loops, double* __restrict__ ploops) {
    constexpr int NY = TOE ? 6 : 3, NX = TOE ? 2 : 1, NCL = TOE ? 2 : 1, NA = TOE ? 3 : 2, NB = 2, NVL = NY + NX;
    const long long w = blk * 64 + threadIdx.x;
    if (w >= total) return;
    const int leg = (int)(w & 1);
    const long long b = w >> 1;
    const int N = T->N;
    const double* z = Z + b * n;
    const DevLoop* L = &D->loop[leg][TOE ? 1 : 0];
    const int l = leg * 2 + (TOE ? 1 : 0);
    const double* lrec = loops + ((size_t)(b * N + N - 1) * D->recs_per_sample + L->rec0) * DIGIT_REC;

    double th[NVL], vp[NVL], lam[NY], Y1[NX][NY];
    {
        double qd, qdd;
        traj_joint(T, basis, z, N - 1, L->drv_traj[0], th[NY], qd, qdd);
        if (TOE) traj_joint(T, basis, z, N - 1, L->drv_traj[1], th[NY + 1], qd, qdd);
    }
#pragma unroll
    for (int r = 0; r < NY; r++) {
        th[r] = lrec[r * DIGIT_REC];
        Y1[0][r] = lrec[r * DIGIT_REC + 1];
        if (TOE) Y1[NX - 1][r] = lrec[r * DIGIT_REC + 2];
        lam[r] = z[n - nd + L->row0 + r];
    }
#pragma unroll
    for (int k = 0; k < NVL; k++) vp[k] = z[n - nv - nd + L->var_joint[k]];

    LoopSolver<NY, NX, NCL, NA, NB> S;
    S.L = L;
    S.geometry(th);
    double Jm[NY][NVL];
    S.jacobian(Jm);
    double* out = ploops + (size_t)b * P->loop_stride + P->loop_off[l];
#pragma unroll
    for (int r = 0; r < NY; r++) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NVL; k++) {
            out[r * NVL + k] = Jm[r][k];
            s += Jm[r][k] * vp[k];
        }
        out[NY * NVL + r] = s;
    }
#pragma unroll
    for (int k = 0; k < NVL; k++) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < NY; r++) s += Jm[r][k] * lam[r];
        out[NY * NVL + NY + k] = s;
    }
    double* oDJv = out + NY * NVL + NY + NVL;
    double* oDJTl = oDJv + NY * NX;
#pragma unroll
    for (int x = 0; x < NX; x++) {
        double U[NVL];
#pragma unroll
        for (int r = 0; r < NY; r++) U[r] = Y1[x][r];
#pragma unroll
        for (int xx = 0; xx < NX; xx++) U[NY + xx] = (xx == x) ? 1.0 : 0.0;
        double d[NY];
#pragma unroll
        for (int r = 0; r < NY; r++) d[r] = 0.0;
        S.add_D2(vp, U, d);
#pragma unroll
        for (int r = 0; r < NY; r++) oDJv[r * NX + x] = d[r];
#pragma unroll 1
        for (int k = 0; k < NVL; k++) {
            double ek[NVL], dk[NY];
#pragma unroll
            for (int j = 0; j < NVL; j++) ek[j] = (j == k) ? 1.0 : 0.0;
#pragma unroll
            for (int r = 0; r < NY; r++) dk[r] = 0.0;
            S.add_D2(ek, U, dk);
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < NY; r++) s += lam[r] * dk[r];
            oDJTl[k * NX + x] = s;
        }
    }
}

// knee and toe loops of both legs in one launch: the first `toe_blocks` workgroups take the toe loops (the longer ones first)
__global__ __launch_bounds__(64) void digit_period_loops_kernel(const DevTraj* __restrict__ T, const DevDigit* __restrict__ D,
                                                                const DevPeriod* __restrict__ P,
                                                                const double* __restrict__ basis, const int n, const int nv,
                                                                const int nd, const double* __restrict__ Z,
                                                                const long long total, const int toe_blocks,
                                                                const double* __restrict__ loops, double* __restrict__ ploops) {
    if ((int)blockIdx.x < toe_blocks) digit_period_loops_body<true>(T, D, P, basis, n, nv, nd, Z, total, blockIdx.x, loops, ploops);
    else digit_period_loops_body<false>(T, D, P, basis, n, nv, nd, Z, total, (long long)blockIdx.x - toe_blocks, loops, ploops);
}

// Phase accounting: -DPER_STOP=k ends the kernel after phase k with a checksum of the whole LDS image (measurement builds only).
#ifdef PER_STOP
#define PER_PHASE_END(k) if (PER_STOP == (k)) { double c_ = 0.0; for (int k_ = tid; k_ < PL_TOTAL; k_ += PL_BLOCK) c_ += lds[k_]; \
                                               if (JAC != nullptr) JAC[rowbase * n + tid] = c_; return; }
#else
#define PER_PHASE_END(k)
#endif
template <int NVM, int NDM, bool DERIV>
__global__ __launch_bounds__(PL_BLOCK, (NVM <= 18 ? 4 : 3)) void legged_period_kernel(const DevModel* __restrict__ M, const DevTraj* __restrict__ T,
                                                                   const DevLegged* __restrict__ Q, const DevPeriod* __restrict__ P,
                                                                   const double* __restrict__ basis, const DevLayout L,
                                                                   const double* __restrict__ Z, const double* __restrict__ core,
                                                                   const double* __restrict__ loops,
                                                                   const double* __restrict__ ploops, const int mrows,
                                                                   const int roff, const DevTraj* __restrict__ T2,
                                                                   const double* __restrict__ basis2,
                                                                   const double* __restrict__ Z2, double* __restrict__ G,
                                                                   double* __restrict__ JAC, double* __restrict__ JAC2) {
    // Z2 == nullptr: the single-step block, rows roff.. of the problem's mrows rows.  Z2 != nullptr: the link between this
    // walking step and the next one (*MultipleStepPeriodicityConstraints.cpp:140-160): the reset rows compare the next step's
    // first sample (its own decision vector Z2, trajectory tables T2 / basis2) with q-, v+, and their derivative with respect to
    // Z2 goes to JAC2 [B][2 Nact][n2]
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    {   // Staggered start (legged_kernel.h has the long story): the workgroups of the first round would walk their trees together and
        // then write their rows together; spread by their slot on the CU (dispatch order: 256 CUs, then the next slot) the Digit instance
        // runs 4096 problems in 0.304 instead of 0.316 ms (units of ~3.5 us per slot: 2 / 5 / 8 -> 0.311 / 0.304 / 0.317).  Launches of
        // less than two rounds start together.
        constexpr int WPC = (NVM <= 18) ? 4 : 3;  // workgroups per CU (launch bounds / LDS)
        constexpr int UNITS = (NVM <= 18) ? PER_STAGGER_SMALL : PER_STAGGER;
        if (UNITS > 0 && gridDim.x >= 2u * PL_CUS * WPC && blockIdx.x < (unsigned)(PL_CUS * WPC)) {
            const int slot = ((int)blockIdx.x / PL_CUS) % WPC;
            for (int k = 0; k < slot * UNITS; k++) __builtin_amdgcn_s_sleep(127);
        }
    }
    const int g = tid & 63;
    const int wv = tid >> 6;
    const bool lead = wv == 0;     // wavefront 0 walks the tree (lane = joint in DFS order)
    const long long b = blockIdx.x;
    const int nj = P->nj, ni = Q->ni, nd = Q->nd, nv = Q->nv, K = Q->K, n = L.n, N = T->N;
    const int ncl = nd - 6;  // closure rows
    const double* z = Z + b * n;
    const double* rec0 = core + (size_t)(b * N) * Q->c_stride;
    const double* recN = core + (size_t)(b * N + N - 1) * Q->c_stride;
    const double* pl = (P->nloops > 0) ? ploops + (size_t)b * P->loop_stride : nullptr;
    const int cv = n - nv - nd, cl = n - nd;
    const size_t rowbase = (size_t)b * mrows + roff;
    const bool link = Z2 != nullptr;
    const bool on = g < nj;
    const int gj = on ? g : 0;
    const int jnt = P->lane_joint[gj];
    const int gend = on ? P->end[gj] : g;
    const int depth = (on && lead) ? P->depth[gj] : 0;
    const int nf = P->new_foot_lane, of = P->old_foot_lane;
    const double arm = on ? M->armature[jnt] : 0.0;
    const V3 lt = v3(z[cl + nd - 6], z[cl + nd - 5], z[cl + nd - 4]);
    const V3 lr = v3(z[cl + nd - 3], z[cl + nd - 2], z[cl + nd - 1]);

    // ---- phase 0: this joint's local transform at q-, the reset velocities; clear the sparse pieces ----
    for (int k = tid; k < PL_ZEROED; k += PL_BLOCK) lds[PL_H + k] = 0.0;  // H, JA and A are adjacent
    for (int k = tid; k < NDM * PL_TS; k += PL_BLOCK) lds[PL_G2 + k] = 0.0;
    // the walks below are chains of dependent LDS reads: keep their indices in LDS too (one byte each), not in global tables
    unsigned char* pathL = reinterpret_cast<unsigned char*>(lds + PL_PATH) + gj * 2 * RAPTOR_LEG_MAX_DEPTH;
    unsigned char* pathJ = pathL + RAPTOR_LEG_MAX_DEPTH;
    const unsigned char* pathOf = reinterpret_cast<unsigned char*>(lds + PL_PATH) + of * 2 * RAPTOR_LEG_MAX_DEPTH;
    double vpl = 0.0, dvj = 0.0;
    if (on && lead) {
        {  // one batch of independent 8-byte loads (the int tables would be a chain of dependent loads per level)
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(pathL);
#pragma unroll
            for (int k = 0; k < RAPTOR_LEG_MAX_DEPTH / 4; k++) dst[k] = P->path8[gj][k];
        }
        const double qm = recN[Q->c_q + jnt];
        const double vm = recN[Q->c_v + jnt];
        vpl = z[cv + jnt];
        dvj = vpl - vm;
        double sq, cq;
        sincos(qm, &sq, &cq);
        M3 Pm, X;
#pragma unroll
        for (int k = 0; k < 9; k++) Pm.m[k] = M->plR[jnt][k];
        V3 o = v3(M->plp[jnt][0], M->plp[jnt][1], M->plp[jnt][2]);
        SV sj;
        joint_world(M->jtype[jnt], qm, cq, sq, Pm, X, o, sj);
        const int jt = M->jtype[jnt];
        const bool prism = jt > 3 || jt < -3;
        const V3 ax = prism ? sj.l : sj.a;
        double* x = lds + PL_LOC + g * 16;
#pragma unroll
        for (int k = 0; k < 9; k++) x[k] = X.m[k];
        x[9] = o.x; x[10] = o.y; x[11] = o.z;
        x[12] = ax.x; x[13] = ax.y; x[14] = ax.z;
        x[15] = prism ? 1.0 : 0.0;
        lds[PL_DV + g] = dvj;
        lds[PL_VP + g] = vpl;
        lds[PL_VM + g] = vm;
    }
    if (DERIV && wv == 1 && on) {
        // wavefront 1, meanwhile: joint rates along every manifold direction (identity rows + P_dep of the last sample) and
        // d v-/d q_t of the closed-loop joints (Y2 contracted with the drivers' rates)
        const int ti = Q->joint_ind[jnt], dd = Q->joint_dep[jnt];
        double* pf = lds + PL_PF + jnt * PL_TS;
        double* vq = lds + PL_VQ + jnt * PL_TS;
        if (ti >= 0) {
            for (int t = 0; t < ni; t++) { pf[t] = (ti == t) ? 1.0 : 0.0; vq[t] = 0.0; }
        } else if (dd < 6) {
            for (int t = 0; t < ni; t++) { pf[t] = recN[Q->c_P + dd * ni + t]; vq[t] = 0.0; }
        } else {
            const int c0 = Q->dep_col[dd][0], c1 = Q->dep_col[dd][1];
            const double* p2 = recN + Q->c_Pl + (dd - 6) * 2;
            const double* y = loops + ((size_t)(b * N + N - 1) * Q->nrec + P->joint_rec[jnt]) * DIGIT_REC;
            const double va = recN[Q->c_v + Q->ind_joint[c0]];
            const double vb = (c1 >= 0) ? recN[Q->c_v + Q->ind_joint[c1]] : 0.0;
            for (int t = 0; t < ni; t++) {
                pf[t] = (t == c0) ? p2[0] : ((t == c1) ? p2[1] : 0.0);
                vq[t] = (t == c0) ? y[3] * va + y[4] * vb : ((t == c1) ? y[4] * va + y[5] * vb : 0.0);
            }
        }
    }
    __syncthreads();
    PER_PHASE_END(1)

    // ---- phase A (wavefront 0): world-frame pose and joint axis (walk from the world down to this joint) ----
    M3 Rw;
#pragma unroll
    for (int k = 0; k < 9; k++) Rw.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
    V3 ow = v3(0, 0, 0);
    SV s = sv_zero();
#pragma unroll 1
    for (int d = 0; d < depth; d++) {
        const double* x = lds + PL_LOC + pathL[d] * 16;
        const M3 R = mul(Rw, x);
        const V3 o = ow + mul(Rw, v3(x[9], x[10], x[11]));
        const V3 axw = mul(Rw, v3(x[12], x[13], x[14]));
        s = (x[15] != 0.0) ? SV{v3(0, 0, 0), axw} : SV{axw, cross(o, axw)};
        Rw = R;
        ow = o;
    }
    if (on && lead) lds_put(lds + PL_S + g * 6, s);
    __syncthreads();
    PER_PHASE_END(2)

    if (DERIV && wv == 1 && g < ni) {
        // ---- wavefront 1, lane = direction t: d v-/d q_t of the floating base.  The old stance foot stays planted along the
        // direction: sum_base s_b Vq_b = -(sum over its path of (w_j x s_j) v-_j + s_j Vq_j), w_j the twist per unit q_t ----
        const int t = g;
        const int dof = P->depth[of];
        SV w = sv_zero(), r = sv_zero();
#pragma unroll 1
        for (int d = 0; d < dof; d++) {
            const int k = pathOf[d];
            const int jk = pathOf[RAPTOR_LEG_MAX_DEPTH + d];
            const SV sj = lds_sv(lds + PL_S + k * 6);
            w = w + sj * lds[PL_PF + jk * PL_TS + t];
            r = r + crm(w, sj) * lds[PL_VM + k] + sj * lds[PL_VQ + jk * PL_TS + t];
        }
        // the floating base (validated at set-up): three prismatic joints along orthonormal axes, then three revolute joints
        SV sb[6];
#pragma unroll
        for (int q = 0; q < 6; q++) sb[q] = lds_sv(lds + PL_S + pathOf[q] * 6);
        const V3 ra = -r.a;
        const V3 c45 = cross(sb[4].a, sb[5].a);
        const double det = dot(sb[3].a, c45);
        const double x3 = dot(ra, c45) / det;
        const double x4 = dot(sb[3].a, cross(ra, sb[5].a)) / det;
        const double x5 = dot(sb[3].a, cross(sb[4].a, ra)) / det;
        const V3 rl = -r.l - sb[3].l * x3 - sb[4].l * x4 - sb[5].l * x5;
        const double xs[6] = {dot(sb[0].l, rl), dot(sb[1].l, rl), dot(sb[2].l, rl), x3, x4, x5};
#pragma unroll
        for (int q = 0; q < 6; q++) lds[PL_VQ + pathOf[RAPTOR_LEG_MAX_DEPTH + q] * PL_TS + t] = xs[q];
    }

    if (lead) {
        // ---- phase B (wavefront 0): values.  a = sum s_j (v+ - v-)_j, V+ = sum s_j v+_j along the path ----
        InertiaO I;
        {
            Inertia Ic;
            const V3 c = ow + mul(Rw, v3(M->lever[jnt][0], M->lever[jnt][1], M->lever[jnt][2]));
            rotate_inertia(Rw, M->Ic[jnt], Ic);
            const double m = on ? M->mass[jnt] : 0.0;
            const double keep = on ? 1.0 : 0.0;
            I.m = m;
            I.hv = c * m;
            const double cc = dot(c, c);
            I.xx = keep * Ic.xx + m * (cc - c.x * c.x);
            I.yy = keep * Ic.yy + m * (cc - c.y * c.y);
            I.zz = keep * Ic.zz + m * (cc - c.z * c.z);
            I.xy = keep * Ic.xy - m * c.x * c.y;
            I.xz = keep * Ic.xz - m * c.x * c.z;
            I.yz = keep * Ic.yz - m * c.y * c.z;
        }
        SV aD = sv_zero(), Vp = sv_zero();
#pragma unroll 1
        for (int d = 0; d < depth; d++) {
            const int j = pathL[d];
            const SV sj = lds_sv(lds + PL_S + j * 6);
            aD = aD + sj * lds[PL_DV + j];
            Vp = Vp + sj * lds[PL_VP + j];
        }
        M3 Rc;
        V3 pc;
        {
            const M3 Rmine = mul(Rw, P->endR);
            const V3 pmine = ow + mul(Rw, v3(P->endp[0], P->endp[1], P->endp[2]));
#pragma unroll
            for (int k = 0; k < 9; k++) Rc.m[k] = __shfl(Rmine.m[k], nf, 64);
            pc = v3(__shfl(pmine.x, nf, 64), __shfl(pmine.y, nf, 64), __shfl(pmine.z, nf, 64));
        }
        SV Vpn;  // V+ of the new stance foot
        Vpn.a = v3(__shfl(Vp.a.x, nf, 64), __shfl(Vp.a.y, nf, 64), __shfl(Vp.a.z, nf, 64));
        Vpn.l = v3(__shfl(Vp.l.x, nf, 64), __shfl(Vp.l.y, nf, 64), __shfl(Vp.l.z, nf, 64));
        // E = d rpy / d omega at the contact frame, and the bilinear form T_ij = d_rpy_rates(R_c, e_i, e_j) contracted with
        // the two fixed vectors it meets (m = lambda's moment part, omega+): nine lanes evaluate one (i, j) pair each, once
        if (g < 9) {
            const int i = g / 3, j = g - 3 * i;
            const V3 ei = v3(i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0);
            const V3 ej = v3(j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0);
            const V3 Tij = d_rpy_rates(Rc, ei, ej);
            lds[PL_E + 9 + g] = dot(lr, Tij);  // A9[i][j]
            const double wi = (i == 0) ? Vpn.a.x : ((i == 1) ? Vpn.a.y : Vpn.a.z);
            lds[PL_TMP + g * 3] = Tij.x * wi; lds[PL_TMP + g * 3 + 1] = Tij.y * wi; lds[PL_TMP + g * 3 + 2] = Tij.z * wi;
            if (j == 0) {  // E[:, i] = rpy_rates(R_c, e_i)
                const V3 Ei = rpy_rates(Rc, ei);
                lds[PL_E + i] = Ei.x; lds[PL_E + 3 + i] = Ei.y; lds[PL_E + 6 + i] = Ei.z;
            }
        }
        wave_sync();
        double bm = 0.0;
        if (g < 9) {  // Bm[r][j] = sum_i T_ij[r] omega+_i
            const int r = g / 3, j = g - 3 * r;
            bm = lds[PL_TMP + (0 * 3 + j) * 3 + r] + lds[PL_TMP + (1 * 3 + j) * 3 + r] + lds[PL_TMP + (2 * 3 + j) * 3 + r];
        }
        wave_sync();
        if (g < 9) lds[PL_E + 18 + g] = bm;
        wave_sync();
        const double* Em = lds + PL_E;
#define M3MUL(m_, q_) v3((m_)[0] * (q_).x + (m_)[1] * (q_).y + (m_)[2] * (q_).z, (m_)[3] * (q_).x + (m_)[4] * (q_).y + (m_)[5] * (q_).z, (m_)[6] * (q_).x + (m_)[7] * (q_).y + (m_)[8] * (q_).z)
        // - J_foot^T lambda.tail(6) is a wrench at the contact point: W = [p_c x f + E^T m ; f]
        const V3 Etl = v3(Em[0] * lr.x + Em[3] * lr.y + Em[6] * lr.z, Em[1] * lr.x + Em[4] * lr.y + Em[7] * lr.z,
                          Em[2] * lr.x + Em[5] * lr.y + Em[8] * lr.z);
        const SV Wc = SV{cross(pc, lt) + Etl, lt};
        SV F = applyO(I, aD);
        if (g == nf) F = F - Wc;
        InertiaO IC = I;
        const int src = gend & 63;
#define SCAN(x) { const double y_ = __shfl_down(x, d, 64); if (take) x += y_; }
#define CUT(x) { const double y_ = __shfl(x, src, 64); x -= y_; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const bool take = (g + d) < 64;
            SCAN(F.a.x) SCAN(F.a.y) SCAN(F.a.z) SCAN(F.l.x) SCAN(F.l.y) SCAN(F.l.z)
            SCAN(IC.m) SCAN(IC.hv.x) SCAN(IC.hv.y) SCAN(IC.hv.z)
            SCAN(IC.xx) SCAN(IC.xy) SCAN(IC.xz) SCAN(IC.yy) SCAN(IC.yz) SCAN(IC.zz)
        }
        CUT(F.a.x) CUT(F.a.y) CUT(F.a.z) CUT(F.l.x) CUT(F.l.y) CUT(F.l.z)
        CUT(IC.m) CUT(IC.hv.x) CUT(IC.hv.y) CUT(IC.hv.z)
        CUT(IC.xx) CUT(IC.xy) CUT(IC.xz) CUT(IC.yy) CUT(IC.yz) CUT(IC.zz)
#undef SCAN
#undef CUT
        const int myloop = on ? P->joint_loop[jnt] : -1;
        const int myvar = on ? P->joint_var[jnt] : 0;
        const double* mypl = (myloop >= 0) ? pl + P->loop_off[myloop] : nullptr;
        const int mynvl = (myloop >= 0) ? P->loop_nvl[myloop] : 0, mynr = (myloop >= 0) ? P->loop_nr[myloop] : 0;
        const bool onnf = on && g <= nf && nf < gend;  // the new stance foot hangs below this joint
        const SV IS = applyO(IC, s);
        if (on) {
            // (1) H (v+ - v-) - J^T lambda   (:88-98)
            double val = dot(s, F) + arm * dvj;
            if (myloop >= 0) val -= mypl[mynr * mynvl + mynr + myvar];
            if (G != nullptr) G[rowbase + jnt] = val;
            // crba(q-): H_kc = s_c . I^C_k s_k for c on the path of k   (:89-96; armature on the diagonal as Pinocchio >= 3 has it)
#pragma unroll 1
            for (int d = 0; d < depth; d++) {
                const int c = pathL[d];
                const int jc = pathJ[d];
                const double h = dot(lds_sv(lds + PL_S + c * 6), IS) + ((c == g) ? arm : 0.0);
                lds[PL_H + jnt * PL_HS + jc] = h;
                lds[PL_H + jc * PL_HS + jnt] = h;
            }
            // stance-foot rows of J after the swap (TalosDynamicsConstraints.cpp:264-279): translation and RPY Jacobian columns
            if (onnf) {
                const V3 Jt = s.l + cross(s.a, pc);
                const V3 Jr = M3MUL(Em, s.a);
                double* ja = lds + PL_JA + ncl * nv + jnt;
                ja[0] = Jt.x; ja[nv] = Jt.y; ja[2 * nv] = Jt.z;
                ja[3 * nv] = Jr.x; ja[4 * nv] = Jr.y; ja[5 * nv] = Jr.z;
            }
            // closure rows of J (Digit): columns of the loop joints only
            if (myloop >= 0) {
                const int row0 = P->loop_row0[myloop];
                for (int r = 0; r < mynr; r++) lds[PL_JA + (row0 + r) * nv + jnt] = mypl[r * mynvl + myvar];
            }
        }
        if (G != nullptr) {
            // (2) J v+   (:119-120)
            if (g == nf) {
                const V3 lin = Vp.l + cross(Vp.a, pc);
                const V3 rot = M3MUL(Em, Vp.a);
                double* o = G + rowbase + nv + ncl;
                o[0] = lin.x; o[1] = lin.y; o[2] = lin.z; o[3] = rot.x; o[4] = rot.y; o[5] = rot.z;
            }
            if (g < ncl) {
                const int l = P->row_loop[g];
                G[rowbase + nv + g] = pl[P->loop_off[l] + P->loop_nr[l] * P->loop_nvl[l] + P->row_local[g]];
            }
            // (3) position reset, (4) velocity reset   (:130-144)
            if (g < ni) {
                if (link) {
                    double qn, vn, an;
                    traj_joint(T2, basis2, Z2 + b * T2->varLength, 0, g, qn, vn, an);
                    const int j = Q->ind_joint[g];
                    G[rowbase + nv + nd + g] = qn - recN[Q->c_q + j];
                    G[rowbase + nv + nd + ni + g] = vn - z[cv + j];
                } else {
                    const int a = P->id1[g], c = P->id2[g];
                    G[rowbase + nv + nd + g] = rec0[Q->c_q + a] + P->sign * recN[Q->c_q + c];
                    G[rowbase + nv + nd + ni + g] = rec0[Q->c_v + a] + P->sign * z[cv + c];
                }
            }
            // (5) contact constraints on the impact impulse   (:152-179)
            if (g == 0) {
                double* o = G + rowbase + nv + nd + 2 * ni;
                const double fz = lt.z;
                o[0] = fz;
                o[1] = lt.x * lt.x + lt.y * lt.y - (Q->mu * fz) * (Q->mu * fz);
                o[2] = lr.z * lr.z - (Q->gamma * fz) * (Q->gamma * fz);
                o[3] = lr.x - Q->Lx * fz;
                o[4] = -Q->Lx * fz - lr.x;
                o[5] = lr.y - Q->Ly * fz;
                o[6] = -Q->Ly * fz - lr.y;
            }
        }
        if (DERIV) {
            // ---- phase C1 (wavefront 0): per-joint vectors of A = d(H dv - J^T lambda)/dq at fixed dv, lambda:
            //   psi_k = a_parent(k) x s_k,   X_k = s_k x* W - d_k W (k above the new stance foot),   cq_k = s_k x* F_k + I^C_k psi_k + X_k
            //   A[j][k] = IS_j . psi_k + [foot below j] s_j . X_k  (k on the path of j),   A[j][k] = s_j . cq_k  (j strictly above k)
            // and the columns of d(J_foot v+)/dq.  (scripts/proto_period.py, matrix_form) ----
            const SV psi = crm(aD - s * dvj, s);
            SV X = sv_zero();
            if (onnf) {
                const double* A9 = lds + PL_E + 9;
                const double* Bm = lds + PL_E + 18;
                const V3 spc = s.l + cross(s.a, pc);  // velocity of the contact point per unit rate of this joint
                X = crf(s, Wc) - SV{cross(spc, lt) + M3MUL(A9, s.a), v3(0, 0, 0)};
                const SV Cp = crm(s, Vpn - (Vp - s * vpl));
                const V3 lin = Cp.l + cross(Cp.a, pc) + cross(Vpn.a, spc);
                const V3 rot = M3MUL(Bm, s.a) + M3MUL(Em, Cp.a);
                double* bk = lds + PL_A + nv * PL_HS + jnt;
                bk[0] = lin.x; bk[PL_HS] = lin.y; bk[2 * PL_HS] = lin.z;
                bk[3 * PL_HS] = rot.x; bk[4 * PL_HS] = rot.y; bk[5 * PL_HS] = rot.z;
            }
            const SV cq = crf(s, F) + applyO(IC, psi) + X;
            wave_sync();  // the local transforms of phase A are dead: their area takes psi, X
            if (on) {
                lds_put(lds + PL_LOC + g * 12, psi);
                lds_put(lds + PL_LOC + g * 12 + 6, X);
            }
            wave_sync();
            // ---- phase C2: rows and columns of A over this joint's path ----
            if (on) {
#pragma unroll 1
                for (int d = 0; d < depth; d++) {
                    const int k = pathL[d];
                    const int jk = pathJ[d];
                    const double* vk = lds + PL_LOC + k * 12;
                    double a = dot(IS, lds_sv(vk));
                    if (onnf) a += dot(s, lds_sv(vk + 6));
                    lds[PL_A + jnt * PL_HS + jk] = a;
                    if (k != g) lds[PL_A + jk * PL_HS + jnt] = dot(lds_sv(lds + PL_S + k * 6), cq);
                }
            }
        }
#undef M3MUL
    }
    if (!DERIV) return;
    __syncthreads();
    PER_PHASE_END(3)

    // ---- phase C3 (all wavefronts): D1 = A Pfull - H Vq, D2 = -H Pfull, D_t(J_foot v+) = B Pfull.  Lane = row (model joint, then
    // the six foot rows), a wavefront takes every PL_WAVES-th direction; Pfull / Vq reads are LDS broadcasts ----
    {
        const bool rowA = g < nv, rowB = g >= nv && g < nv + 6;
        const double* Arow = lds + PL_A + ((g < nv + 6) ? g : 0) * PL_HS;
        const double* Hrow = lds + PL_H + (rowA ? g : 0) * PL_HS;
        const double hs = rowA ? 1.0 : 0.0;
        constexpr int TPW = (RAPTOR_LEG_MAX_IND + PL_WAVES - 1) / PL_WAVES;
        double d1[TPW], d2[TPW];
        int tt[TPW];
#pragma unroll
        for (int i = 0; i < TPW; i++) {
            d1[i] = d2[i] = 0.0;
            tt[i] = (wv + i * PL_WAVES < ni) ? wv + i * PL_WAVES : 0;
        }
#pragma unroll 2
        for (int c = 0; c < nv; c++) {
            const double a = Arow[c], h = Hrow[c] * hs;
            const double* pf = lds + PL_PF + c * PL_TS;
            const double* vq = lds + PL_VQ + c * PL_TS;
#pragma unroll
            for (int i = 0; i < TPW; i++) {
                const double p = pf[tt[i]], q = vq[tt[i]];
                d1[i] += a * p - h * q;
                d2[i] -= h * p;
            }
        }
        __syncthreads();  // every read of the tree-phase areas (S, psi, X) is over: D1 / D2 take their place
        const int rl = rowA ? P->joint_loop[g] : -1;
#pragma unroll
        for (int i = 0; i < TPW; i++) {
            const int t = wv + i * PL_WAVES;
            if (t >= ni) continue;
            if (rowA) {
                double v1 = d1[i];
                if (rl >= 0) {  // closure part of d(J^T lambda)/dq along the loop's own drivers (digit_period_loops_kernel)
                    const int nvl = P->loop_nvl[rl], nr = P->loop_nr[rl], nx = P->loop_nx[rl];
                    const double* dj = pl + P->loop_off[rl] + nr * nvl + nr + nvl + nr * nx + P->joint_var[g] * nx;
                    if (t == P->loop_col[rl][0]) v1 -= dj[0];
                    else if (nx > 1 && t == P->loop_col[rl][1]) v1 -= dj[1];
                }
                lds[PL_D1 + g * ni + t] = v1;
                lds[PL_D2 + g * ni + t] = d2[i];
            } else if (rowB) {
                lds[PL_G2 + (ncl + g - nv) * ni + t] = d1[i];
            }
        }
        if (lead && g < ncl) {  // D_t (J_closure v+): only the loop's own drivers move it
            const int l = P->row_loop[g];
            const int nx = P->loop_nx[l];
            const double* dj = pl + P->loop_off[l] + P->loop_nr[l] * P->loop_nvl[l] + P->loop_nr[l] + P->loop_nvl[l] + P->row_local[g] * nx;
            lds[PL_G2 + g * ni + P->loop_col[l][0]] = dj[0];
            if (nx > 1) lds[PL_G2 + g * ni + P->loop_col[l][1]] = dj[1];
        }
    }

    PER_PHASE_END(4)
    // ---- phase D: Bernstein weights of the first and last sample, d g5, then the dense rows ----
    if (lead && g < K) {
        const double* b0 = link ? basis2 : basis;                 // the first sample is the next step's in a link
        lds[PL_W + g] = b0[g];                                    // B_k(t_0)
        lds[PL_W + 16 + g] = b0[K + g];                           // dB_k(t_0)
        lds[PL_W + 32 + g] = basis[(size_t)((N - 1) * 3) * K + g];      // B_k(t_N-1)
        lds[PL_W + 48 + g] = basis[(size_t)((N - 1) * 3 + 1) * K + g];  // dB_k(t_N-1)
    }
    if (wv == 1 && g < ni) {  // which columns the reset rows touch: looked up once here, not per row in the store loops
        int* idt = reinterpret_cast<int*>(lds + PL_IDT);
        idt[g] = link ? -1 : Q->joint_ind[P->id1[g]];
        idt[16 + g] = link ? g : Q->joint_ind[P->id2[g]];
        idt[32 + g] = link ? Q->ind_joint[g] : P->id2[g];
    }
    if (lead && g < 42) {  // d g5 / d lambda.tail(6)   (:181-204)
        const int r = g / 6, c = g - r * 6;
        const double fz = lt.z, mu = Q->mu, gam = Q->gamma;
        double val = 0.0;
        switch (r) {
            case 0: val = (c == 2) ? 1.0 : 0.0; break;
            case 1: val = (c == 0) ? 2.0 * lt.x : ((c == 1) ? 2.0 * lt.y : ((c == 2) ? -2.0 * mu * mu * fz : 0.0)); break;
            case 2: val = (c == 5) ? 2.0 * lr.z : ((c == 2) ? -2.0 * gam * gam * fz : 0.0); break;
            case 3: val = (c == 3) ? 1.0 : ((c == 2) ? -Q->Lx : 0.0); break;
            case 4: val = (c == 3) ? -1.0 : ((c == 2) ? -Q->Lx : 0.0); break;
            case 5: val = (c == 4) ? 1.0 : ((c == 2) ? -Q->Ly : 0.0); break;
            default: val = (c == 4) ? -1.0 : ((c == 2) ? -Q->Ly : 0.0); break;
        }
        lds[PL_C5 + g] = val;
    }
    __syncthreads();
    PER_PHASE_END(5)
    // a thread owns a pair of adjacent columns (classified once) and every ngr-th row below it: 16-byte stores, a row's
    // pairs contiguous across the lanes, no per-element index arithmetic.  With n = 132 the 66 pairs alone would leave
    // three of the four wavefronts idle through 91 dependent iterations, hence the row groups.
    const int ndyn = ni * K;
    double* out = JAC + rowbase * n;
    const bool pairok = ((n & 1) == 0) && ((reinterpret_cast<size_t>(JAC) & 15) == 0) && ((reinterpret_cast<size_t>(JAC2) & 15) == 0);
    const int wdt = pairok ? 2 : 1;
    const int ncp = (n + wdt - 1) / wdt;
    const int ngr = (ncp < PL_BLOCK) ? PL_BLOCK / ncp : 1;
    const double* D1s = lds + PL_D1;
    const double* D2s = lds + PL_D2;
    const double* G2s = lds + PL_G2;
    const double* Hs = lds + PL_H;
    const double* JAs = lds + PL_JA;
    const double sgn = P->sign;
    const int* idt = reinterpret_cast<const int*>(lds + PL_IDT);
    for (int slot = tid; slot < ncp * ngr; slot += PL_BLOCK) {
        const int g = slot / ncp, cp = slot - g * ncp;
        const int c0 = cp * wdt;
        int kind[2], t[2], a[2];
        double wn0[2], wn1[2], w00[2], w01[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c = c0 + ((h < wdt) ? h : 0);
            kind[h] = (c < ndyn) ? 0 : ((c < cl) ? 1 : 2);
            t[h] = (c < ndyn) ? c / K : 0;
            const int k = (c < ndyn) ? c - t[h] * K : 0;
            a[h] = (c < ndyn) ? 0 : ((c < cl) ? c - cv : c - cl);
            w00[h] = lds[PL_W + k]; w01[h] = lds[PL_W + 16 + k]; wn0[h] = lds[PL_W + 32 + k]; wn1[h] = lds[PL_W + 48 + k];
        }
        double* o = out + c0;
#define PUT(row, v0, v1) { double* o_ = o + (size_t)(row) * n; if (pairok) store2_stream(o_, v0, v1); else o_[0] = v0; }
        for (int r = g; r < nv; r += ngr) {  // pg1_pz (:114-116)
            double v[2];
#pragma unroll
            for (int h = 0; h < 2; h++)
                v[h] = (kind[h] == 0) ? D1s[r * ni + t[h]] * wn0[h] + D2s[r * ni + t[h]] * wn1[h]
                                      : ((kind[h] == 1) ? Hs[r * PL_HS + a[h]] : -JAs[a[h] * nv + r]);
            PUT(r, v[0], v[1])
        }
        o += (size_t)nv * n;
        for (int d = g; d < nd; d += ngr) {  // pg2_pz (:126-127)
            double v[2];
#pragma unroll
            for (int h = 0; h < 2; h++)
                v[h] = (kind[h] == 0) ? G2s[d * ni + t[h]] * wn0[h] : ((kind[h] == 1) ? JAs[d * nv + a[h]] : 0.0);
            PUT(d, v[0], v[1])
        }
        o += (size_t)nd * n;
        for (int i = g; i < ni; i += ngr) {  // pg3_pz (:137; link: -pq_pz(last) rows of the independent joints)
            const int t1 = idt[i], t2 = idt[16 + i];
            const double s2 = link ? -1.0 : sgn;
            double v[2];
#pragma unroll
            for (int h = 0; h < 2; h++)
                v[h] = (kind[h] == 0) ? ((t[h] == t1) ? w00[h] : 0.0) + ((t[h] == t2) ? s2 * wn0[h] : 0.0) : 0.0;
            PUT(i, v[0], v[1])
        }
        o += (size_t)ni * n;
        for (int i = g; i < ni; i += ngr) {  // pg4_pz (:148; link: -pv_plus_pz rows of the independent joints)
            const int t1 = idt[i], j2 = idt[32 + i];
            const double s2 = link ? -1.0 : sgn;
            double v[2];
#pragma unroll
            for (int h = 0; h < 2; h++)
                v[h] = (kind[h] == 0) ? ((t[h] == t1) ? w01[h] : 0.0) : ((kind[h] == 1 && a[h] == j2) ? s2 : 0.0);
            PUT(i, v[0], v[1])
        }
        o += (size_t)ni * n;
        for (int i = g; i < 7; i += ngr) {  // pg5_pz (:181-204)
            double v[2];
#pragma unroll
            for (int h = 0; h < 2; h++) v[h] = (kind[h] == 2 && a[h] >= nd - 6) ? lds[PL_C5 + i * 6 + a[h] - (nd - 6)] : 0.0;
            PUT(i, v[0], v[1])
        }
#undef PUT
        if (link) {  // pg3_pz2, pg4_pz2: pq_pz(0), pq_d_pz(0) of the next step (same column layout as this step's z)
            double* o2 = JAC2 + (size_t)b * 2 * ni * n + c0;
            for (int i = g; i < 2 * ni; i += ngr) {
                const int ti = (i < ni) ? i : i - ni;
                double v[2];
#pragma unroll
                for (int h = 0; h < 2; h++) v[h] = (kind[h] == 0 && t[h] == ti) ? ((i < ni) ? w00[h] : w01[h]) : 0.0;
                double* o_ = o2 + (size_t)i * n;
                if (pairok) store2_stream(o_, v[0], v[1]); else o_[0] = v[0];
            }
        }
    }
}

#endif
