// K_kinova_hess_torque : the torque-row part of Optimizer::eval_h for KinovaOptimizer (Optimization/src/Optimizer.cpp:499-635,
// KinematicsDynamics/src/InverseDynamics.cpp:141-171), every derivative chain of a sample walked ONCE and no mixed chain at all.
//
// phi(z) = mu^T tau = sum_l W_l . f_l (virtual power, kinova_hess_kernel.h) and
//     d2 phi / dz_a dz_b = sum_l  W^ab.f + W^a.f^b + W^b.f^a + W.f^ab ,     f = I A + V x* I V.
// The round-2 kernel walked the chain once per pair in hyper-dual numbers: the value part 28 times, every first-order part 7 times.
// Here:
//   * 8 adjacent lanes own one time sample: lane 7 walks the VALUE chain (V, A, W per link, link-local frames), lane a < 7 the
//     FIRST-ORDER chain of z_a.  Past its seeded link a first-order chain obeys the same linear recursion as the value chain
//     (X_up applied to V^a, A^a, W^a plus the velocity coupling), so the eight lanes run one instruction stream; at link a the
//     chain is seeded from the value lane's state:  d(X_up u)/dq = (X_up u) x S_a  (S_a = the joint axis), times pq_pz, plus the
//     joint's own rates (pq_d_pz, pq_dd_pz: ArmourBezierCurves.cpp:147-151 is diagonal);
//   * the MIXED parts need no chain of their own.  Spatial cross products commute with Pluecker transforms, so with J_b = the axis
//     of joint b carried to the link being walked and Jd_b its rate (Jd_b = X_up Jd_b + (J_b x S_l) q_d_l, zero at link b) -- two
//     more 6-vectors on lane b --
//         V^ab = w0 V^a x J_b,   W^ab = w0 W^a x J_b,   A^ab = (w0 A^a + w1 V^a) x J_b + w0 V^a x Jd_b      (a <= b <= l),
//     a = b included.  Moving the cross products onto the forces they are multiplied with, a pair's term at a link is five
//     6-vector products of a covector of chain a with a vector of chain b:
//         J_b.G^a + Jd_b.G2^a + f^b.W^a + W^b.f^a - V^b.Q^a ,
//         G^a  = -[w0 W^a x* f + (w0 A^a + w1 V^a) x* (I W) + w0 V^a x* c],   G2^a = -w0 V^a x* (I W),
//         Q^a  = W x* (I V^a) - I (W x V^a),   c = the covector of  W.(u x* I V + V x* I u) = c.u ;
//   * per link every chain lane leaves (G, G2, J, Jd), then (W, f, Q, V) in its LDS record (two stages over the same 24 doubles),
//     and the 28 pairs, dealt round-robin in order of b over the eight lanes, read the two records they need: at link l the
//     (l+1)(l+2)/2 live pairs take ceil(./8) rounds, 14 rounds over the 7 links.  The records are exchanged between lanes of ONE
//     wavefront only: LDS operations of a wavefront execute in order, no barrier.
// scripts/proto_hess_shared.py is the numpy statement of this arithmetic (hess_stateless), equal to the hyper-dual form to 1e-15.
//
// Frames: every link frame is re-labelled cyclically so that its joint axis is the third one (a proper rotation: cross products are
// unchanged); the re-labelling is applied to the model constants as they are read (wave-uniform indices, scalar loads), so one
// instruction stream serves the three axis types.
//
// Launch: one workgroup (4 wavefronts) per decision vector, a wavefront takes 8 samples per pass; the sums over samples stay in the
// pair's lane, then three shuffles, then a fixed-order sum over the four wavefronts: deterministic, no atomics.  The result is ADDED
// to what kinova_hessian_kernel<., false> (obstacle rows) left in Hout / in the split form's partial slot, on the same stream.
#ifndef RAPTOR_KINOVA_HESS_TORQUE_KERNEL_H
#define RAPTOR_KINOVA_HESS_TORQUE_KERNEL_H

#include "kinova_hess_kernel.h"

#define HT_REC 26  // doubles per lane record: 24 used; 52 dwords: 16 consecutive lanes' 128-bit accesses tile the 64 banks
#define HT_WAVE (64 * HT_REC)
#define HT_TAB (H_WAVES * HT_WAVE)   // per wavefront: [8 samples][7 joints][q_d, q_dd, cos, sin, mu] + [8 samples][w0, w1, w2, -]
#define HT_TAB_S 40                  // doubles per sample (35 + 3, padded)
#define HT_TAB_WAVE (8 * HT_TAB_S)
#define HT_ACC (HT_TAB + H_WAVES * HT_TAB_WAVE)
#define HT_TOTAL (HT_ACC + H_WAVES * 32)
#ifndef HT_OCC
#define HT_OCC 2
#endif

RD V3 xez(V3 x, double s) { return V3{x.y * s, -x.x * s, 0.0}; }  // x cross (s e_z)
RD V3 mulE(const double (&E)[9], V3 x) {
    return V3{E[0] * x.x + E[1] * x.y + E[2] * x.z, E[3] * x.x + E[4] * x.y + E[5] * x.z, E[6] * x.x + E[7] * x.y + E[8] * x.z};
}
RD SV xup(const double (&E)[9], V3 p, SV x) {  // plux(R^T, p) then the joint rotation, as one matrix
    return SV{mulE(E, x.a), mulE(E, x.l + cross(x.a, p))};
}
RD SV ld_sv(const double* p) {  // 16-byte aligned
    const v2d_t a = *reinterpret_cast<const v2d_t*>(p), b = *reinterpret_cast<const v2d_t*>(p + 2), c = *reinterpret_cast<const v2d_t*>(p + 4);
    return SV{V3{a.x, a.y, b.x}, V3{b.y, c.x, c.y}};
}
RD void st_sv(double* p, SV x) {
    *reinterpret_cast<v2d_t*>(p) = v2d_t{x.a.x, x.a.y};
    *reinterpret_cast<v2d_t*>(p + 2) = v2d_t{x.a.z, x.l.x};
    *reinterpret_cast<v2d_t*>(p + 4) = v2d_t{x.l.y, x.l.z};
}
RD int sym6(int i, int j) {  // index of (i, j) in (xx, xy, xz, yy, yz, zz)
    const int a = i < j ? i : j, b = i < j ? j : i;
    return (a * (5 - a)) / 2 + b;
}

template <bool SPLIT>
__global__ __launch_bounds__(RAPTOR_BLOCK, HT_OCC) void kinova_hess_torque_kernel(
    const DevModel* __restrict__ M, const DevTraj* __restrict__ T, const double* __restrict__ basis, const double* __restrict__ wsum,
    const DevLayout L, const double* __restrict__ Z, const double* __restrict__ lambda, double* __restrict__ Hout) {
    extern __shared__ __align__(16) double lds[];
    constexpr int NV = 7;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 7, grp = lane >> 3;
    const bool isval = (c == 7);
    const long long b = blockIdx.x;
    const int N = T->N;
    const double* z = Z + b * NV;
    const double* lam = lambda + b * (long long)L.m;
    double* rec_base = lds + wave * HT_WAVE + (lane & ~7) * HT_REC;
    double* my = rec_base + c * HT_REC;
    const double* vrec = rec_base + 7 * HT_REC;
    double* tab = lds + HT_TAB + wave * HT_TAB_WAVE + grp * HT_TAB_S;

    // this lane's (up to) four pairs, dealt in order of b: (0,0) (0,1) (1,1) (0,2) ...; packed 4 bits each, b = 15: none
    unsigned pairs = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = 8 * r + c;
        int pb = 0;
        while ((pb + 1) * (pb + 2) / 2 <= i) pb++;
        const unsigned pa = (i < H_PAIRS) ? i - pb * (pb + 1) / 2 : 0, pbb = (i < H_PAIRS) ? pb : 15;
        pairs |= (pa | (pbb << 4)) << (8 * r);
    }
    asm volatile("" : "+v"(pairs));
    double acc[4] = {0.0, 0.0, 0.0, 0.0};

    const int NG = (N + 7) >> 3;  // groups of 8 samples
#pragma unroll 1
    for (int g = (SPLIT ? (int)blockIdx.y * H_WAVES : 0) + wave; g < NG; g += SPLIT ? 7 * H_WAVES : H_WAVES) {
        {   // trajectory of joint c at this lane's sample -> the wavefront's table
            const int i = g * 8 + grp;
            const int ii = min(i, N - 1);
            if (c < NV) {
                double q, qd, qdd, sq, cq;
                traj_joint(T, basis, z, ii, c, q, qd, qdd);
                sincos_joint(q, &sq, &cq);
                double* t = tab + c * 5;
                t[0] = qd; t[1] = qdd; t[2] = cq; t[3] = sq;
                t[4] = (i < N) ? lam[L.off_tau + ii * NV + c] : 0.0;  // a sample past N has no multipliers: W = 0 on every chain, it adds nothing
            } else {
                tab[35] = wsum[ii * 3]; tab[36] = wsum[ii * 3 + 1]; tab[37] = wsum[ii * 3 + 2];
            }
        }
        __builtin_amdgcn_wave_barrier();
        SV V = sv_zero(), W = sv_zero(), J = sv_zero(), Jd = sv_zero();
        SV A = SV{V3{0.0, 0.0, 0.0}, isval ? v3(M->neg_gravity[0], M->neg_gravity[1], M->neg_gravity[2]) : V3{0.0, 0.0, 0.0}};
        int pc0 = 0, pc1 = 1, pc2 = 2;  // the parent frame's re-labelling
#pragma unroll 1
        for (int l = 0; l < NV; l++) {
            const int jt = M->jtype[l];
            const double sg = jt < 0 ? -1.0 : 1.0;
            const int Kx = (jt < 0 ? -jt : jt) - 1, Ix = Kx == 2 ? 0 : Kx + 1, Jx = Ix == 2 ? 0 : Ix + 1;
            const double* t = tab + l * 5;
            const double vjs = sg * t[0];
            const double w0 = tab[35];
            {
                double E[9];
                {
                    const double C = t[2], Sn = sg * t[3];
                    const double* R = M->plR[l];
                    const double ri0 = R[pc0 * 3 + Ix], ri1 = R[pc1 * 3 + Ix], ri2 = R[pc2 * 3 + Ix];
                    const double rj0 = R[pc0 * 3 + Jx], rj1 = R[pc1 * 3 + Jx], rj2 = R[pc2 * 3 + Jx];
                    E[0] = C * ri0 + Sn * rj0; E[1] = C * ri1 + Sn * rj1; E[2] = C * ri2 + Sn * rj2;
                    E[3] = C * rj0 - Sn * ri0; E[4] = C * rj1 - Sn * ri1; E[5] = C * rj2 - Sn * ri2;
                    E[6] = R[pc0 * 3 + Kx]; E[7] = R[pc1 * 3 + Kx]; E[8] = R[pc2 * 3 + Kx];
                }
                const V3 p = v3(M->plp[l][pc0], M->plp[l][pc1], M->plp[l][pc2]);
                // ---- this lane's chain: X_up, then the joint's own motion on the value lane (CustomizedInverseDynamics.cpp:155-157) ----
                V = xup(E, p, V); A = xup(E, p, A); W = xup(E, p, W); J = xup(E, p, J); Jd = xup(E, p, Jd);
            }
            V.a.z += isval ? vjs : 0.0;
            A.a = A.a + xez(V.a, vjs);
            A.l = A.l + xez(V.l, vjs);
            Jd.a = Jd.a + xez(J.a, vjs);
            Jd.l = Jd.l + xez(J.l, vjs);
            A.a.z += isval ? sg * t[1] : 0.0;
            W.a.z += isval ? sg * t[4] : 0.0;
            if (isval) {
                st_sv(my, V);
                st_sv(my + 6, A);
                st_sv(my + 12, W);
            }
            __builtin_amdgcn_wave_barrier();
            const SV Vv = ld_sv(vrec), Wv = ld_sv(vrec + 12);
            Inertia Il;
            {
                Il.mass = M->mass[l];
                Il.c = v3(M->lever[l][Ix], M->lever[l][Jx], M->lever[l][Kx]);
                const double* Ic = M->Ic[l];
                Il.xx = Ic[sym6(Ix, Ix)]; Il.xy = Ic[sym6(Ix, Jx)]; Il.xz = Ic[sym6(Ix, Kx)];
                Il.yy = Ic[sym6(Jx, Jx)]; Il.yz = Ic[sym6(Jx, Kx)]; Il.zz = Ic[sym6(Kx, Kx)];
            }
            SV fv;
            {
                const SV Av = ld_sv(vrec + 6);
                if (c == l) {  // seeded here: pq_pz d(X_up u)/dq + the joint's own rates; the axis itself starts its journey
                    const double w1 = tab[36], w2 = tab[37];
                    V.a = xez(Vv.a, sg * w0);
                    V.l = xez(Vv.l, sg * w0);
                    V.a.z += sg * w1;
                    A.a = xez(Av.a - xez(Vv.a, vjs), sg * w0) + xez(V.a, vjs) + xez(Vv.a, sg * w1);
                    A.l = xez(Av.l - xez(Vv.l, vjs), sg * w0) + xez(V.l, vjs) + xez(Vv.l, sg * w1);
                    A.a.z += sg * w2;
                    W.a = xez(Wv.a, sg * w0);
                    W.l = xez(Wv.l, sg * w0);
                    J = sv_zero();
                    J.a.z = sg;
                    Jd = sv_zero();
                }
                fv = apply(Il, Av);
            }
            // ---- value-level force quantities (every lane: one instruction stream) ----
            const SV Hv = apply(Il, Vv);
            fv = fv + crf(Vv, Hv);
            // ---- stage 1: (G, G2 | J, Jd), signs folded: the pair term subtracts ----
            {
                const SV IW = apply(Il, Wv);
                const SV cv = SV{cross(Hv.a, Wv.a) + cross(Hv.l, Wv.l), cross(Hv.l, Wv.a)} - apply(Il, crm(Vv, Wv));
                const SV X1 = crf(V, IW);
                const SV G = (crf(W, fv) + crf(A, IW) + crf(V, cv)) * w0 + X1 * tab[36];
                if (!isval) {
                    st_sv(my, G);
                    st_sv(my + 6, X1 * w0);
                    st_sv(my + 12, J);
                    st_sv(my + 18, Jd);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pa = (pairs >> (8 * r)) & 15, pb = (pairs >> (8 * r + 4)) & 15;
                if (pb <= l) {
                    const double* ra = rec_base + pa * HT_REC;
                    const double* rb = rec_base + pb * HT_REC;
                    acc[r] -= dot(ld_sv(rb + 12), ld_sv(ra)) + dot(ld_sv(rb + 18), ld_sv(ra + 6));
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- stage 2: (W, f, Q | V) ----
            {
                const SV Hc = apply(Il, V);
                const SV fc = apply(Il, A) + crf(V, Hv) + crf(Vv, Hc);  // f^a = I A^a + V^a x* I V + V x* I V^a
                const SV Q = crf(Wv, Hc) - apply(Il, crm(Wv, V));
                if (!isval) {
                    st_sv(my, W);
                    st_sv(my + 6, fc);
                    st_sv(my + 12, Q);
                    st_sv(my + 18, V);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pa = (pairs >> (8 * r)) & 15, pb = (pairs >> (8 * r + 4)) & 15;
                if (pb <= l) {
                    const double* ra = rec_base + pa * HT_REC;
                    const double* rb = rec_base + pb * HT_REC;
                    acc[r] += (dot(ld_sv(rb + 6), ld_sv(ra)) + dot(ld_sv(rb), ld_sv(ra + 6))) - dot(ld_sv(rb + 18), ld_sv(ra + 12));
                }
            }
            __builtin_amdgcn_wave_barrier();  // the next link's records overwrite these
            pc0 = Ix; pc1 = Jx; pc2 = Kx;
        }
    }

    // ---- sums over the 8 sample groups of the wavefront, then over the wavefronts in a fixed order ----
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double x = acc[r];
        x += __shfl_xor(x, 8, 64);
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        if (lane < 8) lds[HT_ACC + wave * 32 + 8 * r + lane] = x;
    }
    __syncthreads();
    if (tid < H_PAIRS) {
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < H_WAVES; w++) x += lds[HT_ACC + w * 32 + tid];
        int pb = 0;
        while ((pb + 1) * (pb + 2) / 2 <= tid) pb++;
        const int pa = tid - pb * (pb + 1) / 2;
        double* out = SPLIT ? Hout + (b * 7 + blockIdx.y) * H_PAIRS : Hout + b * H_PAIRS;
        out[pair_index(pa, pb)] += x;
    }
}

#endif
