// K_loops : Digit's closed kinematic loops (DigitDynamicsConstraints::setupJointPosition and the loop rows of
// get_c / get_J / get_Jx_partial_dq / get_Jxy_partial_dq, Examples/Digit/src/DigitDynamicsConstraints.cpp:242-400,
// 443-3019; readable twin DigitDynamicsConstraints_clean.cpp:458-1040).
//
// MI355X design.  The reference solves all 24 dependent coordinates at once (GSL hybridsj on a 24 x 24 Jacobian built
// from 3000 lines of generated trigonometric polynomials) and then forms dense 24 x 36 derivative blocks.  The system
// is block triangular: each leg has a knee loop (3 unknowns driven by the knee) and a toe loop (6 unknowns driven by
// toe_A, toe_B) that do not depend on anything else, and the six floating-base coordinates follow from the stance
// foot (legged_kernel.h).  So: one lane per (sample-eval, leg, loop), everything in registers,
//   Newton on c(y; x) = 0 from the reference's fitted initial guess, LU with partial pivoting (3 x 3 / 6 x 6),
//   Y1 = dy/dx, Y2 = d2y/dx2, Y3 = d3y/dx3 by implicit differentiation, with the serial-chain identities
//       dp/dth_i = w_i x r_i,  d2p/dth_i dth_j = w_i x (w_j x r_j) (i <= j),  d3p = w_i x (w_j x (w_k x r_k)) (i <= j <= k)
//   (numpy model: scripts/proto_digit.py, agrees with the oracle's literal restatement to 1e-14).
// The per-joint records (10 doubles: y, Y1[2], Y2[3], Y3[4]) feed the dynamics kernel, which needs nothing else from
// the loops: theta' = Y1 x', theta'' = Y1 x'' + x'^T Y2 x', and the chain-rule weights of d tau / dz.
#ifndef RAPTOR_DIGIT_KERNEL_H
#define RAPTOR_DIGIT_KERNEL_H

#include "kernels.h"
#include "legged_kernel.h"

#define DIGIT_REC 10  // doubles per dependent loop joint: y, Y1a, Y1b, Y2aa, Y2ab, Y2bb, Y3aaa, Y3aab, Y3abb, Y3bbb

struct DevChainJoint {
    double plR[9], plp[3];
    int jtype;
    int var;  // index into the loop's local variable vector: 0..ny-1 unknowns, ny.. drivers
};
struct DevClosure {
    int nA, nB;
    DevChainJoint A[3], B[2];  // rod chain, anchor chain (both start in the loop's base link frame)
    double endA[3], endB[3];
};
struct DevLoop {
    int ny, nx, ncl;
    DevClosure cl[2];
    int drv_traj[2];   // trajectory index of the driving joints
    int rec0;          // first dependent-joint record of this loop in the per-sample loop buffer
    int var_joint[8];  // model joint of each local variable
    int row0;          // first row of the loop's closures in the reference's constraint order (DigitDynamicsConstraints.cpp:579-596)
    double g_sign[6], g_c0[6], g_lin[6][8], g_quad[6][36];  // initial guess (digit_tables.h); g_quad: upper triangle f_a f_b, a <= b
};
struct DevDigit {
    DevLoop loop[2][2];  // [leg][0: knee, 1: toe]
    int recs_per_sample;  // 18
};

// Local variable of every chain joint, fixed by the structure of Digit's loops (unknowns first in the order of
// DIGIT_LOOPS[].depj, then the drivers); the host checks its model-derived tables against these, so that every index into a
// per-lane array below is a compile-time constant (registers, no scratch).
//   toe loop : unknowns toe_A_rod 0, A2 1, toe_B_rod 2, B2 3, toe_pitch 4, toe_roll 5; drivers toe_A 6, toe_B 7
//              closure 0: rod chain toe_A, toe_A_rod, A2 | anchor chain toe_pitch, toe_roll;  closure 1: toe_B, toe_B_rod, B2 | same
//   knee loop: unknowns achilles_rod 0, ach2 1, tarsus 2; driver knee 3;  rod chain achilles_rod, ach2 | anchor chain knee, tarsus
__host__ __device__ constexpr int digit_chain_var(bool toe, int closure, bool rod_side, int i) {
    return toe ? (rod_side ? (closure == 0 ? (i == 0 ? 6 : (i == 1 ? 0 : 1)) : (i == 0 ? 7 : (i == 1 ? 2 : 3))) : (i == 0 ? 4 : 5))
               : (rod_side ? (i == 0 ? 0 : 1) : (i == 0 ? 3 : 2));
}

template <int N>
struct LU {
    double a[N][N];
    int piv[N];
};
// LU with partial pivoting, fully unrolled so that the matrix stays in registers (row swaps are predicated selects)
template <int N>
RD void lu_factor(LU<N>& m) {
#pragma unroll
    for (int c = 0; c < N; c++) {
        int p = c;
        double best = fabs(m.a[c][c]);
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const double v = fabs(m.a[r][c]);
            if (v > best) { best = v; p = r; }
        }
        m.piv[c] = p;
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const bool sw = (r == p);
#pragma unroll
            for (int k = 0; k < N; k++) {
                const double x = m.a[c][k], y = m.a[r][k];
                m.a[c][k] = sw ? y : x;
                m.a[r][k] = sw ? x : y;
            }
        }
        const double inv = 1.0 / m.a[c][c];
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const double f = m.a[r][c] * inv;
            m.a[r][c] = f;
#pragma unroll
            for (int k = c + 1; k < N; k++) m.a[r][k] -= f * m.a[c][k];
        }
    }
}
template <int N>
RD void lu_solve(const LU<N>& m, double* b) {
    // the factorisation swapped whole rows (multipliers included): P A = L U, so permute b completely first
#pragma unroll
    for (int c = 0; c < N; c++) {
        const int p = m.piv[c];
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const bool sw = (r == p);
            const double x = b[c], y = b[r];
            b[c] = sw ? y : x;
            b[r] = sw ? x : y;
        }
    }
#pragma unroll
    for (int c = 0; c < N; c++) {
#pragma unroll
        for (int r = c + 1; r < N; r++) b[r] -= m.a[r][c] * b[c];
    }
#pragma unroll
    for (int r = N - 1; r >= 0; r--) {
        double s = b[r];
#pragma unroll
        for (int k = r + 1; k < N; k++) s -= m.a[r][k] * b[k];
        b[r] = s / m.a[r][r];
    }
}

// geometry of one serial chain in the loop's base frame: joint axes w, joint origins o, end point p
template <int NJ>
struct ChainGeo {
    V3 w[NJ], o[NJ], p;
};
template <int NJ, bool TOE, int CLOSURE, bool ROD>
RD void chain_geometry(const DevChainJoint* __restrict__ J, const double* __restrict__ end, const double* th, ChainGeo<NJ>& g,
                       M3* Rend = nullptr, V3* oend = nullptr) {
    M3 Rw;
#pragma unroll
    for (int k = 0; k < 9; k++) Rw.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
    V3 ow = v3(0, 0, 0);
#pragma unroll
    for (int i = 0; i < NJ; i++) {
        const M3 A = mul(Rw, J[i].plR);
        V3 o = ow + mul(Rw, v3(J[i].plp[0], J[i].plp[1], J[i].plp[2]));
        double sq, cq;
        const double q = th[digit_chain_var(TOE, CLOSURE, ROD, i)];
        sincos_joint(q, &sq, &cq);
        M3 R;
        SV s;
        joint_world(J[i].jtype, q, cq, sq, A, R, o, s);
        g.w[i] = s.a;
        g.o[i] = o;
        Rw = R;
        ow = o;
    }
    g.p = ow + mul(Rw, v3(end[0], end[1], end[2]));
    if (Rend != nullptr) { *Rend = Rw; *oend = ow; }
}
// D^2 p [u, v] and D^3 p [u, v, w] for per-joint direction components (chain order)
template <int NJ>
RD V3 chain_D2(const ChainGeo<NJ>& g, const double* u, const double* v) {
    V3 out = v3(0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const V3 t = cross(g.w[j], g.p - g.o[j]);
#pragma unroll
        for (int i = 0; i <= j; i++) {
            const double c = (i == j) ? u[i] * v[i] : u[i] * v[j] + u[j] * v[i];
            out = axpy(out, c, cross(g.w[i], t));
        }
    }
    return out;
}
template <int NJ>
RD V3 chain_D3(const ChainGeo<NJ>& g, const double* u, const double* v, const double* w) {
    V3 out = v3(0, 0, 0);
#pragma unroll
    for (int k = 0; k < NJ; k++) {
        const V3 t = cross(g.w[k], g.p - g.o[k]);
#pragma unroll
        for (int j = 0; j <= k; j++) {
            const V3 t2 = cross(g.w[j], t);
#pragma unroll
            for (int i = 0; i <= j; i++) {
                // sum over the distinct arrangements of the multiset {i, j, k} onto (u, v, w)
                double c;
                if (i == j && j == k) c = u[i] * v[i] * w[i];
                else if (i == j) c = u[i] * v[i] * w[k] + u[i] * v[k] * w[i] + u[k] * v[i] * w[i];
                else if (j == k) c = u[i] * v[k] * w[k] + u[k] * v[i] * w[k] + u[k] * v[k] * w[i];
                else c = u[i] * v[j] * w[k] + u[i] * v[k] * w[j] + u[j] * v[i] * w[k] + u[j] * v[k] * w[i] + u[k] * v[i] * w[j] + u[k] * v[j] * w[i];
                out = axpy(out, c, cross(g.w[i], t2));
            }
        }
    }
    return out;
}

// NY unknowns, NX drivers, NCL closures (3 rows each, NY == 3 * NCL), chain lengths NA (rod side), NB (anchor side)
template <int NY, int NX, int NCL, int NA, int NB>
struct LoopSolver {
    ChainGeo<NA> ga[NCL];
    ChainGeo<NB> gb[NCL];
    const DevLoop* L;

    static constexpr bool TOE = NCL == 2;
    RD void geometry(const double* th) {
        chain_geometry<NA, TOE, 0, true>(L->cl[0].A, L->cl[0].endA, th, ga[0]);
        M3 Rb;
        V3 ob;
        chain_geometry<NB, TOE, 0, false>(L->cl[0].B, L->cl[0].endB, th, gb[0], &Rb, &ob);
        if (NCL > 1) {
            chain_geometry<NA, TOE, 1, true>(L->cl[NCL - 1].A, L->cl[NCL - 1].endA, th, ga[NCL - 1]);
            // the toe loop's two closures hang on the SAME anchor chain (toe_pitch, toe_roll; checked at set-up): only the end point differs
            gb[NCL - 1] = gb[0];
            const double* e1 = L->cl[NCL - 1].endB;
            gb[NCL - 1].p = ob + mul(Rb, v3(e1[0], e1[1], e1[2]));
        }
    }
    RD void residual(double* c) const {
#pragma unroll
        for (int k = 0; k < NCL; k++) {
            const V3 d = ga[k].p - gb[k].p;
            c[3 * k] = d.x; c[3 * k + 1] = d.y; c[3 * k + 2] = d.z;
        }
    }
    // dense Jacobian wrt all NY + NX local variables
    RD void jacobian(double (*Jm)[NY + NX]) const {
#pragma unroll
        for (int r = 0; r < 3 * NCL; r++)
#pragma unroll
            for (int k = 0; k < NY + NX; k++) Jm[r][k] = 0.0;
#pragma unroll
        for (int c = 0; c < NCL; c++) {
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const V3 t = cross(ga[c].w[i], ga[c].p - ga[c].o[i]);
                const int var = digit_chain_var(TOE, c, true, i);
                Jm[3 * c][var] += t.x; Jm[3 * c + 1][var] += t.y; Jm[3 * c + 2][var] += t.z;
            }
#pragma unroll
            for (int i = 0; i < NB; i++) {
                const V3 t = cross(gb[c].w[i], gb[c].p - gb[c].o[i]);
                const int var = digit_chain_var(TOE, c, false, i);
                Jm[3 * c][var] -= t.x; Jm[3 * c + 1][var] -= t.y; Jm[3 * c + 2][var] -= t.z;
            }
        }
    }
    RD void gather(int c, const double* U, double* ua, double* ub) const {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            ua[i] = U[digit_chain_var(TOE, c, true, i)];
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            ub[i] = U[digit_chain_var(TOE, c, false, i)];
        }
    }
    // out += D^2 c [U, V]
    RD void add_D2(const double* U, const double* V, double* out) const {
#pragma unroll
        for (int c = 0; c < NCL; c++) {
            double ua[NA], ub[NB], va[NA], vb[NB];
            gather(c, U, ua, ub);
            gather(c, V, va, vb);
            const V3 d = chain_D2<NA>(ga[c], ua, va) - chain_D2<NB>(gb[c], ub, vb);
            out[3 * c] += d.x; out[3 * c + 1] += d.y; out[3 * c + 2] += d.z;
        }
    }
    RD void add_D3(const double* U, const double* V, const double* W, double* out) const {
#pragma unroll
        for (int c = 0; c < NCL; c++) {
            double ua[NA], ub[NB], va[NA], vb[NB], wa[NA], wb[NB];
            gather(c, U, ua, ub);
            gather(c, V, va, vb);
            gather(c, W, wa, wb);
            const V3 d = chain_D3<NA>(ga[c], ua, va, wa) - chain_D3<NB>(gb[c], ub, vb, wb);
            out[3 * c] += d.x; out[3 * c + 1] += d.y; out[3 * c + 2] += d.z;
        }
    }
};

// One lane per (sample-eval, leg); `total` = sample-evals, grid = 2 x ceil(total / 64).  TOE = false: knee loops (NY 3, NX 1), TOE = true: toe loops (NY 6, NX 2).
#ifndef DIGIT_LOOPS_OCC
#define DIGIT_LOOPS_OCC 1
#endif
template <bool TOE>
RD void digit_loops_body(const DevTraj* __restrict__ T, const DevDigit* __restrict__ D, const double* __restrict__ basis, const int n,
                         const double* __restrict__ Z, const long long total, const int blk, double* __restrict__ loops,
                         int* __restrict__ status, int* __restrict__ iterations) {
    constexpr int NY = TOE ? 6 : 3, NX = TOE ? 2 : 1, NCL = TOE ? 2 : 1, NA = TOE ? 3 : 2, NB = 2, NS = NX * (NX + 1) / 2,
                  NT = NX * (NX + 1) * (NX + 2) / 6;
    // a wavefront works on ONE leg (blockIdx parity), so the loop's constants -- placements, end points, initial-guess tables -- sit
    // behind a uniform pointer and come through scalar loads into SGPR operands; with the leg taken from the lane index they were
    // ~180 vector loads (and 41 waits) in every Newton step
    const int leg = blk & 1;
    const long long e = (long long)(blk >> 1) * 64 + threadIdx.x;
    if (e >= total) return;
    const int N = T->N;
    const int i = (int)(e % N);
    const long long b = e / N;
    const double* z = Z + b * n;
    const DevLoop* L = &D->loop[leg][TOE ? 1 : 0];

    double th[NY + NX];
    double f[8];
    {
        double qd, qdd;
        traj_joint(T, basis, z, i, L->drv_traj[0], th[NY], qd, qdd);
        if (TOE) traj_joint(T, basis, z, i, L->drv_traj[1], th[NY + 1], qd, qdd);
        // initial guess: +-asin of a trigonometric polynomial of the drivers (DigitDynamicsConstraints.cpp:255-314)
        if (TOE) {
            const double q1 = th[NY], q2 = th[NY + 1];
            f[0] = cos(q1); f[1] = cos(q2); f[2] = sin(q1); f[3] = sin(q2);
            f[4] = cos(2.0 * q1); f[5] = cos(2.0 * q2); f[6] = sin(2.0 * q1); f[7] = sin(2.0 * q2);
        } else {
            const double q1 = th[NY];
            f[0] = cos(q1); f[1] = sin(q1); f[2] = cos(2.0 * q1); f[3] = sin(2.0 * q1);
            f[4] = f[5] = f[6] = f[7] = 0.0;
        }
#pragma unroll
        for (int r = 0; r < NY; r++) {
            double s = L->g_c0[r];
#pragma unroll
            for (int k = 0; k < (TOE ? 8 : 4); k++) s += L->g_lin[r][k] * f[k];
            if (TOE) {  // the pair terms as a dense upper-triangular quadratic form: constant indices, uniform coefficient loads
                int k = 0;
#pragma unroll
                for (int a = 0; a < 8; a++)
#pragma unroll
                    for (int bb = a; bb < 8; bb++) s += L->g_quad[r][k++] * (f[a] * f[bb]);
            }
            th[r] = L->g_sign[r] * safe_asin(s);
        }
    }

    LoopSolver<NY, NX, NCL, NA, NB> S;
    S.L = L;
    double Jm[NY][NY + NX];
    LU<NY> lu;
    // Newton (stands in for gsl_multiroot_fdfsolver_hybridsj, :344-367: same root; like the reference at least one step,
    // residual test sum|c| < 1e-14 after a step, at most 50 steps).  Geometry, Jacobian and LU of the accepted point are
    // reused by the derivative phase below.
    int iter = 0;
    double res_last = 0.0;
#pragma unroll 1
    while (true) {
        double c[NY];
        S.geometry(th);
        S.residual(c);
        S.jacobian(Jm);
#pragma unroll
        for (int r = 0; r < NY; r++)
#pragma unroll
            for (int k = 0; k < NY; k++) lu.a[r][k] = Jm[r][k];
        lu_factor<NY>(lu);
        double res = 0.0;
#pragma unroll
        for (int r = 0; r < NY; r++) res += fabs(c[r]);
        res_last = res;
        if ((iter > 0 && res < 1e-14) || iter >= 50 || !(res == res)) break;  // NaN: singular loop Jacobian, flows to the outputs
        lu_solve<NY>(lu, c);
#pragma unroll
        for (int r = 0; r < NY; r++) th[r] -= c[r];
        iter++;
    }

    if (status != nullptr) {
        // what the reference silently accepts (DigitDynamicsConstraints.cpp:350-367: the GSL loop just ends), made visible per z
        bool finite = (res_last == res_last);
#pragma unroll
        for (int r = 0; r < NY; r++) finite = finite && (fabs(th[r]) <= 1.79e308);
        const int flag = !finite ? RAPTOR_STATUS_NONFINITE : (!(res_last < 1e-14) ? RAPTOR_STATUS_NOT_CONVERGED : 0);
        if (flag) atomicOr(status + b, flag);
        if (iterations != nullptr) atomicMax(iterations + b, iter);
    }

#if defined(DIGIT_LOOPS_STOP) && DIGIT_LOOPS_STOP == 1  // measurement build: Newton only
    {
        double* out = loops + ((size_t)e * D->recs_per_sample + L->rec0) * DIGIT_REC;
        for (int r = 0; r < NY; r++) out[r * DIGIT_REC] = th[r] + lu.a[r][r];
        return;
    }
#endif
    double Y1[NX][NY], U[NX][NY + NX];
#pragma unroll
    for (int a = 0; a < NX; a++) {
#pragma unroll
        for (int r = 0; r < NY; r++) Y1[a][r] = -Jm[r][NY + a];
        lu_solve<NY>(lu, Y1[a]);
#pragma unroll
        for (int r = 0; r < NY; r++) U[a][r] = Y1[a][r];
#pragma unroll
        for (int x = 0; x < NX; x++) U[a][NY + x] = (x == a) ? 1.0 : 0.0;
    }
    // Y2[(a,b)], a <= b, packed aa, ab, bb
    double Y2[NS][NY], U2[NS][NY + NX];
    {
        int s = 0;
#pragma unroll
        for (int a = 0; a < NX; a++)
#pragma unroll
            for (int bb = a; bb < NX; bb++) {
                double rhs[NY];
#pragma unroll
                for (int r = 0; r < NY; r++) rhs[r] = 0.0;
                S.add_D2(U[a], U[bb], rhs);
#pragma unroll
                for (int r = 0; r < NY; r++) rhs[r] = -rhs[r];
                lu_solve<NY>(lu, rhs);
#pragma unroll
                for (int r = 0; r < NY; r++) { Y2[s][r] = rhs[r]; U2[s][r] = rhs[r]; }
#pragma unroll
                for (int x = 0; x < NX; x++) U2[s][NY + x] = 0.0;
                s++;
            }
    }
    // Y3[(a,b,d)], a <= b <= d, packed aaa, aab, abb, bbb
    double Y3[NT][NY];
#if defined(DIGIT_LOOPS_STOP) && DIGIT_LOOPS_STOP == 2  // measurement build: no third-order derivatives
    for (int t = 0; t < NT; t++)
        for (int r = 0; r < NY; r++) Y3[t][r] = 0.0;
    if (false)
#endif
    {
        int t = 0;
#pragma unroll
        for (int a = 0; a < NX; a++)
#pragma unroll
            for (int bb = a; bb < NX; bb++)
#pragma unroll
                for (int d = bb; d < NX; d++) {
                    // pair index of (x, y), x <= y, in the packed order: NX = 1 -> 0 ; NX = 2 -> aa 0, ab 1, bb 2
                    const int sab = a + bb, sad = a + d, sbd = bb + d;
                    double rhs[NY];
#pragma unroll
                    for (int r = 0; r < NY; r++) rhs[r] = 0.0;
                    S.add_D3(U[a], U[bb], U[d], rhs);
                    S.add_D2(U2[sab], U[d], rhs);
                    S.add_D2(U2[sad], U[bb], rhs);
                    S.add_D2(U2[sbd], U[a], rhs);
#pragma unroll
                    for (int r = 0; r < NY; r++) rhs[r] = -rhs[r];
                    lu_solve<NY>(lu, rhs);
#pragma unroll
                    for (int r = 0; r < NY; r++) Y3[t][r] = rhs[r];
                    t++;
                }
    }
    double* out = loops + ((size_t)e * D->recs_per_sample + L->rec0) * DIGIT_REC;
#pragma unroll
    for (int r = 0; r < NY; r++) {
        double* o = out + r * DIGIT_REC;
        o[0] = th[r];
        o[1] = Y1[0][r];
        o[2] = TOE ? Y1[NX - 1][r] : 0.0;
        o[3] = Y2[0][r];
        o[4] = TOE ? Y2[NS > 1 ? 1 : 0][r] : 0.0;
        o[5] = TOE ? Y2[NS - 1][r] : 0.0;
        o[6] = Y3[0][r];
        o[7] = TOE ? Y3[NT > 1 ? 1 : 0][r] : 0.0;
        o[8] = TOE ? Y3[NT > 2 ? 2 : 0][r] : 0.0;
        o[9] = TOE ? Y3[NT - 1][r] : 0.0;
    }
}


template <bool TOE>
__global__ __launch_bounds__(64, DIGIT_LOOPS_OCC) void digit_loops_kernel(const DevTraj* __restrict__ T, const DevDigit* __restrict__ D,
                                                         const double* __restrict__ basis, const int n,
                                                         const double* __restrict__ Z, const long long total,
                                                         double* __restrict__ loops, int* __restrict__ status,
                                                         int* __restrict__ iterations) {
    digit_loops_body<TOE>(T, D, basis, n, Z, total, (int)blockIdx.x, loops, status, iterations);
}
// Knee and toe loops in ONE launch (they are independent): the first `toe_blocks` workgroups take the toe loops, the rest the knee
// loops.  For a single problem the two kinds then run side by side instead of one launch after the other.
__global__ __launch_bounds__(64, DIGIT_LOOPS_OCC) void digit_loops_both_kernel(const DevTraj* __restrict__ T, const DevDigit* __restrict__ D,
                                                              const double* __restrict__ basis, const int n,
                                                              const double* __restrict__ Z, const long long total, const int toe_blocks,
                                                              double* __restrict__ loops, int* __restrict__ status,
                                                              int* __restrict__ iterations) {
    if ((int)blockIdx.x < toe_blocks) digit_loops_body<true>(T, D, basis, n, Z, total, (int)blockIdx.x, loops, status, iterations);
    else digit_loops_body<false>(T, D, basis, n, Z, total, (int)blockIdx.x - toe_blocks, loops, status, iterations);
}

// The multipliers of the closure rows, ConstrainedInverseDynamics::lambda(i) / plambda_pz(i) rows 0 .. 17
// (KinematicsDynamics/src/ConstrainedInverseDynamics.cpp:154, 169-171).  The reference solves J_dep^T lambda = tau_dep with the
// 24 x 24 QR of J_dep; J_dep is block triangular (a closure row touches its own loop's joints only, the contact rows touch the base
// and the stance leg), so with the contact multipliers known (the root wrench, legged_kernel.h) the rows of loop l are
//     C_y^T lambda_l = tau_tree,y        C_y = d c_l / d y,   tau_tree = joint torques of the re-rooted tree,
//     d lambda_l = C_y^-T ( d tau_tree,y  -  (d C_y^T) lambda_l ),   d C_y / d x_a = D^2 c [., U_a]  along the loop's own drivers only.
// One lane per (sample-eval, leg); geometry at the root the loops kernel found (its records), tau_tree,y and its derivatives per
// unit (q, v, a) of every actuator from the core kernel's `lamrec`.  Component view only: no constraint of the reference reads
// these rows, so the hot path never runs this.
//   lam  [E][nd]            rows row0 .. row0 + NY - 1 of this loop
//   dlam [E][ni][3][nd]     d lambda / d (q, v, a) of actuator it
template <bool TOE>
__global__ __launch_bounds__(64) void digit_lambda_kernel(const DevTraj* __restrict__ T, const DevDigit* __restrict__ D,
                                                          const double* __restrict__ basis, const int n,
                                                          const double* __restrict__ Z, const long long total,
                                                          const double* __restrict__ loops, const double* __restrict__ lamrec,
                                                          const int ni, const int nd, double* __restrict__ lam,
                                                          double* __restrict__ dlam) {
    constexpr int NY = TOE ? 6 : 3, NX = TOE ? 2 : 1, NCL = TOE ? 2 : 1, NA = TOE ? 3 : 2, NB = 2, NVL = NY + NX;
    const long long w = (long long)blockIdx.x * 64 + threadIdx.x;
    if (w >= total) return;
    const int leg = (int)(w & 1);
    const long long e = w >> 1;
    const int N = T->N;
    const int i = (int)(e % N);
    const long long b = e / N;
    const double* z = Z + b * n;
    const DevLoop* L = &D->loop[leg][TOE ? 1 : 0];
    const int nrec = D->recs_per_sample;
    const double* lrec = loops + ((size_t)e * nrec + L->rec0) * DIGIT_REC;
    const double* tr = lamrec + (size_t)e * (nrec * (1 + 3 * ni));

    double th[NVL], Y1[NX][NY];
    {
        double qd, qdd;
        traj_joint(T, basis, z, i, L->drv_traj[0], th[NY], qd, qdd);
        if (TOE) traj_joint(T, basis, z, i, L->drv_traj[1], th[NY + 1], qd, qdd);
    }
#pragma unroll
    for (int r = 0; r < NY; r++) {
        th[r] = lrec[r * DIGIT_REC];
        Y1[0][r] = lrec[r * DIGIT_REC + 1];
        if (TOE) Y1[NX - 1][r] = lrec[r * DIGIT_REC + 2];
    }
    LoopSolver<NY, NX, NCL, NA, NB> S;
    S.L = L;
    S.geometry(th);
    double Jm[NY][NVL];
    S.jacobian(Jm);
    LU<NY> lu;  // of C_y^T
#pragma unroll
    for (int r = 0; r < NY; r++)
#pragma unroll
        for (int k = 0; k < NY; k++) lu.a[k][r] = Jm[r][k];
    lu_factor<NY>(lu);
    double lm[NY];
#pragma unroll
    for (int k = 0; k < NY; k++) lm[k] = tr[L->rec0 + k];
    lu_solve<NY>(lu, lm);
#pragma unroll
    for (int r = 0; r < NY; r++) lam[(size_t)e * nd + L->row0 + r] = lm[r];
    if (dlam == nullptr) return;
    // (d C_y^T / d x_a) lambda for the loop's own drivers: component k = lambda . D^2 c [e_k, U_a]
    double corr[NX][NY];
#pragma unroll
    for (int a = 0; a < NX; a++) {
        double U[NVL];
#pragma unroll
        for (int r = 0; r < NY; r++) U[r] = Y1[a][r];
#pragma unroll
        for (int x = 0; x < NX; x++) U[NY + x] = (x == a) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < NY; k++) {
            double ek[NVL], dk[NY];
#pragma unroll
            for (int j = 0; j < NVL; j++) ek[j] = (j == k) ? 1.0 : 0.0;
#pragma unroll
            for (int r = 0; r < NY; r++) dk[r] = 0.0;
            S.add_D2(ek, U, dk);
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < NY; r++) s += lm[r] * dk[r];
            corr[a][k] = s;
        }
    }
    const int c0 = L->drv_traj[0], c1 = TOE ? L->drv_traj[NX - 1] : -1;
#pragma unroll 1
    for (int it = 0; it < ni; it++) {
#pragma unroll 1
        for (int x = 0; x < 3; x++) {
            double rhs[NY];
            const double* tsrc = tr + nrec + (it * 3 + x) * nrec + L->rec0;
#pragma unroll
            for (int k = 0; k < NY; k++) {
                double v = tsrc[k];
                if (x == 0 && it == c0) v -= corr[0][k];
                if (TOE && x == 0 && it == c1) v -= corr[NX - 1][k];
                rhs[k] = v;
            }
            lu_solve<NY>(lu, rhs);
            double* o = dlam + (((size_t)e * ni + it) * 3 + x) * nd + L->row0;
#pragma unroll
            for (int r = 0; r < NY; r++) o[r] = rhs[r];
        }
    }
}

// ConstrainedInverseDynamics::plambda_pz(i), all nd rows, [E][nd][n]: closure rows from `dlam` (digit_lambda_kernel), the last six
// (contact wrench) from the core kernel's records L[x][6][it]
__global__ __launch_bounds__(RAPTOR_BLOCK) void legged_lambda_expand_kernel(const DevTraj* __restrict__ T,
                                                                            const DevLegged* __restrict__ Q,
                                                                            const double* __restrict__ basis, const int n,
                                                                            const double* __restrict__ core,
                                                                            const double* __restrict__ dlam, const long long total,
                                                                            double* __restrict__ out) {
    const long long w = (long long)blockIdx.x * RAPTOR_BLOCK + threadIdx.x;
    if (w >= total) return;
    const int nd = Q->nd, ni = Q->ni, K = Q->K;
    const int c = (int)(w % n);
    const long long er = w / n;
    const int lr = (int)(er % nd);
    const long long e = er / nd;
    double val = 0.0;
    if (c < ni * K) {
        const int t = c / K, k = c - t * K;
        const int i = (int)(e % T->N);
        const double* Bs = basis + (size_t)(i * 3) * K + k;
        if (lr >= nd - 6) {
            const double* Lr = core + (size_t)e * Q->c_stride + Q->c_L + (lr - (nd - 6)) * ni + t;
            val = Lr[0] * Bs[0] + Lr[6 * ni] * Bs[K] + Lr[12 * ni] * Bs[2 * K];
        } else {
            const double* d = dlam + (((size_t)e * ni + t) * 3) * nd + lr;
            val = d[0] * Bs[0] + d[nd] * Bs[K] + d[2 * nd] * Bs[2 * K];
        }
    }
    out[w] = val;
}

#endif
