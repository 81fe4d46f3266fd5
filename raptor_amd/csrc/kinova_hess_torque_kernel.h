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
//     a = b included.  Moving the cross products onto the forces they are multiplied with, a pair's term at a link is six
//     6-vector products of a covector of chain a with a vector of chain b:
//         J_b.G^a + Jd_b.G2^a + f^b.W^a + W^b.f^a - U^a.H^b - U^b.H^a ,
//         G^a  = -[w0 W^a x* f + (w0 A^a + w1 V^a) x* (I W) + w0 V^a x* c],   G2^a = -w0 V^a x* (I W),
//         U^a  = V^a x W,  H^a = I V^a,   c = the covector of  W.(u x* I V + V x* I u) = c.u ;
//   * per link every chain lane leaves (G, G2, J, Jd), then (W, f, U, H) in its LDS record (two stages over the same 24 doubles),
//     and the 28 pairs, dealt round-robin in order of b over the eight lanes, read the two records they need: at link l the
//     (l+1)(l+2)/2 live pairs take ceil(./8) rounds, 14 rounds over the 7 links.  The records are exchanged between lanes of ONE
//     wavefront only: LDS operations of a wavefront execute in order, no barrier.
// scripts/proto_hess_shared.py is the numpy statement of this arithmetic (hess_stateless), equal to the hyper-dual form to 1e-15.
//
// Frames: every link frame is re-labelled cyclically so that its joint axis is the third one (a proper rotation: cross products are
// unchanged); the host lays the model constants out in those frames once per problem (DevHessLinks), so one instruction stream
// serves the three axis types and a link's constants are 23 consecutive doubles of scalar loads.
//
// Launch: one workgroup (4 wavefronts) per decision vector, a wavefront takes 8 samples per pass; the sums over samples stay in the
// pair's lane, then three shuffles, then a fixed-order sum over the four wavefronts: deterministic, no atomics.  The result is ADDED
// to what kinova_hessian_kernel (obstacle rows) left in Hout / in the split form's partial slot, on the same stream.
#ifndef RAPTOR_KINOVA_HESS_TORQUE_KERNEL_H
#define RAPTOR_KINOVA_HESS_TORQUE_KERNEL_H

#include "kinova_hess_kernel.h"

#ifndef HT_REC
#define HT_REC 26  // doubles per lane record: 24 used
#endif
#ifndef HT_GPAD
#define HT_GPAD 8  // doubles between the record blocks of two samples: block stride = 432 dwords = 48 mod 64 banks (0 / 4 / 16 doubles: 1.42 / 1.43 / 1.68 ms, 8: 1.385)
#endif
#define HT_GROUP (8 * HT_REC + HT_GPAD)
#define HT_WAVE (8 * HT_GROUP)
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
    return V3{__builtin_fma(E[0], x.x, __builtin_fma(E[1], x.y, E[2] * x.z)), __builtin_fma(E[3], x.x, __builtin_fma(E[4], x.y, E[5] * x.z)),
              __builtin_fma(E[6], x.x, __builtin_fma(E[7], x.y, E[8] * x.z))};
}
RD SV xup(const double (&E)[9], V3 p, SV x) {  // plux(R^T, p) then the joint rotation, as one matrix
    const V3 t = V3{__builtin_fma(x.a.y, p.z, __builtin_fma(-x.a.z, p.y, x.l.x)), __builtin_fma(x.a.z, p.x, __builtin_fma(-x.a.x, p.z, x.l.y)),
                    __builtin_fma(x.a.x, p.y, __builtin_fma(-x.a.y, p.x, x.l.z))};
    return SV{mulE(E, x.a), mulE(E, t)};
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
// the same products as spatial_dev.h's crf / crm / apply, written as chains of fused multiply-adds (the kernel is bound by VALU issue)
#define HT_FMA __builtin_fma
RD V3 cross_acc(V3 acc, V3 a, V3 b) {  // acc + a x b
    return V3{HT_FMA(a.y, b.z, HT_FMA(-a.z, b.y, acc.x)), HT_FMA(a.z, b.x, HT_FMA(-a.x, b.z, acc.y)), HT_FMA(a.x, b.y, HT_FMA(-a.y, b.x, acc.z))};
}
RD V3 cross_sub(V3 acc, V3 a, V3 b) {  // acc - a x b
    return V3{HT_FMA(-a.y, b.z, HT_FMA(a.z, b.y, acc.x)), HT_FMA(-a.z, b.x, HT_FMA(a.x, b.z, acc.y)), HT_FMA(-a.x, b.y, HT_FMA(a.y, b.x, acc.z))};
}
RD SV crf_acc(SV acc, SV v, SV f) {  // acc + v x* f
    return SV{cross_acc(cross_acc(acc.a, v.a, f.a), v.l, f.l), cross_acc(acc.l, v.a, f.l)};
}
RD SV crf_sub(SV acc, SV v, SV f) {  // acc - v x* f
    return SV{cross_sub(cross_sub(acc.a, v.a, f.a), v.l, f.l), cross_sub(acc.l, v.a, f.l)};
}
RD SV crf0(SV v, SV f) { return SV{cross_acc(cross(v.a, f.a), v.l, f.l), cross(v.a, f.l)}; }
RD SV crm0(SV v, SV m) { return SV{cross(v.a, m.a), cross_acc(cross(v.a, m.l), v.l, m.a)}; }
RD SV apply_f(const Inertia& I, SV x) {
    const V3 l = cross_acc(x.l, x.a, I.c) * I.mass;
    const V3 n = V3{HT_FMA(I.xx, x.a.x, HT_FMA(I.xy, x.a.y, I.xz * x.a.z)), HT_FMA(I.xy, x.a.x, HT_FMA(I.yy, x.a.y, I.yz * x.a.z)),
                    HT_FMA(I.xz, x.a.x, HT_FMA(I.yz, x.a.y, I.zz * x.a.z))};
    return SV{cross_acc(n, I.c, l), l};
}
RD SV sv_fma(SV x, double s, SV y) {  // x * s + y
    return SV{V3{HT_FMA(x.a.x, s, y.a.x), HT_FMA(x.a.y, s, y.a.y), HT_FMA(x.a.z, s, y.a.z)},
              V3{HT_FMA(x.l.x, s, y.l.x), HT_FMA(x.l.y, s, y.l.y), HT_FMA(x.l.z, s, y.l.z)}};
}

template <bool SPLIT>
__global__ __launch_bounds__(RAPTOR_BLOCK, HT_OCC) void kinova_hess_torque_kernel(
    const DevHessLinks* __restrict__ HL, const DevModel* __restrict__ M, const DevTraj* __restrict__ T, const double* __restrict__ basis, const double* __restrict__ wsum,
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
    double* rec_base = lds + wave * HT_WAVE + grp * HT_GROUP;
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
#pragma unroll 1
        for (int l = 0; l < NV; l++) {
            const double* K_ = HL->c[l];  // the link's constants, axis-last frames (device_types.h DevHessLinks)
            const double sg = K_[22];
            const double* t = tab + l * 5;
            const double vjs = sg * t[0];
            const double w0 = tab[35];
            {
                double E[9];
                {
                    const double C = t[2], Sn = sg * t[3];
                    E[0] = C * K_[0] + Sn * K_[3]; E[1] = C * K_[1] + Sn * K_[4]; E[2] = C * K_[2] + Sn * K_[5];
                    E[3] = C * K_[3] - Sn * K_[0]; E[4] = C * K_[4] - Sn * K_[1]; E[5] = C * K_[5] - Sn * K_[2];
                    E[6] = K_[6]; E[7] = K_[7]; E[8] = K_[8];
                }
                const V3 p = v3(K_[9], K_[10], K_[11]);
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
            Il.mass = K_[12];
            Il.c = v3(K_[13], K_[14], K_[15]);
            Il.xx = K_[16]; Il.xy = K_[17]; Il.xz = K_[18]; Il.yy = K_[19]; Il.yz = K_[20]; Il.zz = K_[21];
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
                fv = apply_f(Il, Av);
            }
            // ---- value-level force quantities (every lane: one instruction stream) ----
            const SV Hv = apply_f(Il, Vv);
            fv = crf_acc(fv, Vv, Hv);
            // ---- stage 1: (G, G2 | J, Jd), signs folded: the pair term subtracts ----
            {
                const SV IW = apply_f(Il, Wv);
                const SV cn = crf_acc(apply_f(Il, crm0(Vv, Wv)), Wv, Hv);  // minus the covector c
                const SV X1 = crf0(V, IW);
                const SV G = sv_fma(crf_sub(crf_acc(crf0(W, fv), A, IW), V, cn), w0, X1 * tab[36]);
                if (!isval) {
                    st_sv(my, G);
                    st_sv(my + 6, X1 * w0);
                    st_sv(my + 12, J);
                    st_sv(my + 18, Jd);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // (lanes whose pair is not live yet sit the round out; making the rounds wave-uniform with a select at the end measured 1 % slower)
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
            // ---- stage 2: (W, f | U, H):  U^a.H^b + U^b.H^a,  U = V x W(value),  H = I V ----
            {
                const SV Hc = apply_f(Il, V);
                const SV fc = crf_acc(crf_acc(apply_f(Il, A), V, Hv), Vv, Hc);  // f^a = I A^a + V^a x* I V + V x* I V^a
                if (!isval) {
                    st_sv(my, W);
                    st_sv(my + 6, fc);
                    st_sv(my + 12, crm0(V, Wv));
                    st_sv(my + 18, Hc);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pa = (pairs >> (8 * r)) & 15, pb = (pairs >> (8 * r + 4)) & 15;
                if (pb <= l) {
                    const double* ra = rec_base + pa * HT_REC;
                    const double* rb = rec_base + pb * HT_REC;
                    acc[r] += (dot(ld_sv(rb + 6), ld_sv(ra)) + dot(ld_sv(rb), ld_sv(ra + 6))) - (dot(ld_sv(rb + 18), ld_sv(ra + 12)) + dot(ld_sv(rb + 12), ld_sv(ra + 18)));
                }
            }
            __builtin_amdgcn_wave_barrier();  // the next link's records overwrite these
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
