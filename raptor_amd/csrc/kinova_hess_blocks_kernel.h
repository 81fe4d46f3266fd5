// eval_h for KinovaLongerHorizonOptimizer (PiecewiseBezierCurves): the Lagrangian Hessian over V = 21 degree (+14) variables.
//     Optimizer::eval_h                                Optimization/src/Optimizer.cpp:499-635
//     InverseDynamics::compute, compute_hessian        KinematicsDynamics/src/InverseDynamics.cpp:141-171
//     KinovaCustomizedConstraints, compute_hessian     KinovaCustomizedConstraints.cpp:117-155
//
// The trajectory is linear in z and column c moves joint c mod 7 with per-sample weights (w0, w1, w2)(i, c) = the entries of
// pq_pz / pq_d_pz / pq_dd_pz.  Unlike ArmourBezierCurves the columns of one joint carry DIFFERENT weight triples, so the
// Hessian is assembled from per-sample second-order blocks of phi = mu^T tau (+ the obstacle distances) in JOINT space,
//     H(c1, c2) = sum_i  w0 w0' (Pqq + Pobs)(j, j') + w1 w1' Pvv(j, j') + w0 w1' Pqv(j, j') + w1 w0' Pqv(j', j)
//                      + w0 w2' Pqa(j, j') + w2 w0' Pqa(j', j),            j = c1 mod 7, j' = c2 mod 7
// (tau is linear in q_dd and its q_dd coefficient does not depend on q_d: no vv-a, a-a blocks).
//   kernel 1 (kinova_hess_blocks_kernel): the 154 block entries per sample -- qq by the hyper-dual forward recursion of
//     kinova_hess_kernel.h with UNIT seeds, vv / qa / qv by the reduced passes there (tau is quadratic in q_d, linear in q_dd) -- an item is (joint j1 <= j2, kind of j1's seed, kind of j2's seed), a wavefront
//     owns an item, lanes = samples -- plus the 28 obstacle entries (a wavefront per sphere); Phi[b][182][N], sample-minor;
//   kernel 2 (kinova_hess_contract_kernel): one workgroup per z, a thread per (c1 <= c2), fixed summation order over samples.
#ifndef RAPTOR_KINOVA_HESS_BLOCKS_KERNEL_H
#define RAPTOR_KINOVA_HESS_BLOCKS_KERNEL_H

#include "kinova_hess_kernel.h"

#define HB_QQ 0      // Phi block offsets: qq (28, upper triangle), vv (28), qv (49: [jq][jv]), qa (49: [jq][ja]), obstacles (28)
#define HB_VV 28
#define HB_QV 56
#define HB_QA 105
#define HB_OBS 154
#define HB_NI 182
#define HB_MAX_ITEMS 48  // per wavefront (154 items dealt to four)
#define HB_SPLIT 12      // workgroups per (z, pass) in the small-batch form: 11 share the items, one owns the obstacle entries

// item code: j1 | j2 << 3 | k1 << 6 | k2 << 8 | out << 10   (j1 <= j2 joints, k = 0: q, 1: q_d, 2: q_dd seed, out = Phi row)
struct DevHessItems {
    int count[H_WAVES];
    int code[H_WAVES][HB_MAX_ITEMS];
};

// LDS: the sample tables HL_X / HL_MU / HL_WS, then ONE region that holds the wavefronts' hyper-dual virtual-velocity state during the
// item phase (4 x 24 fields, kinova_hess_kernel.h) and the world-frame tables HL_W / HL_RO of the obstacle phase afterwards; the
// obstacle sums of the four wavefronts meet in the HL_X area, which nothing reads after the forward kinematics: 72 KB, two
// workgroups per CU; registers capped at 256 for two wavefronts per SIMD.
#define HB_SUM HL_X  // [28][64] obstacle sums (after the forward kinematics)
static_assert(H_PAIRS <= 35, "the obstacle sums fit the HL_X area");
__global__ __launch_bounds__(RAPTOR_BLOCK, 2) void kinova_hess_blocks_kernel(
    const DevModel* __restrict__ M, const DevTraj* __restrict__ T, const double* __restrict__ basis,
    const double* __restrict__ wsum, const DevSpheres* __restrict__ S, const DevBox* __restrict__ boxes, const DevLayout L,
    const DevHessItems* __restrict__ items, const double* __restrict__ Z, const double* __restrict__ lambda,
    double* __restrict__ Phi) {
    __shared__ double lds[HL_ACC];
    constexpr int NV = 7;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long b = blockIdx.x;
    const int N = T->N;
    const int ns = S->num_spheres;
    const int O = S->num_obstacles;
    const double* z = Z + b * (long long)L.n;
    const double* lam = lambda + b * (long long)L.m;
    double* phi = Phi + b * (long long)HB_NI * N;
    // small batches are spread over more workgroups (launch grid y, z): blockIdx.y picks the item table, the LAST y also owns the
    // obstacle entries; blockIdx.z strides the 64-sample passes.  Every Phi entry has one writer and one fixed summation order.
    const DevHessItems* my = items + blockIdx.y;
    const bool do_obs = blockIdx.y == gridDim.y - 1;

#pragma unroll 1
    for (int i0 = blockIdx.z * H_CH; i0 < N; i0 += gridDim.z * H_CH) {
        for (int idx = tid; idx < H_CH * NV; idx += RAPTOR_BLOCK) {
            const int s = idx & (H_CH - 1), j = idx >> 6;
            const int i = min(i0 + s, N - 1);
            double q, qd, qdd, sq, cq;
            traj_joint(T, basis, z, i, j, q, qd, qdd);
            sincos(q, &sq, &cq);
            double* x = lds + HL_X + (j * 5) * H_CH + s;
            x[0] = qd; x[H_CH] = qdd; x[2 * H_CH] = cq; x[3 * H_CH] = sq; x[4 * H_CH] = q;
            lds[HL_MU + j * H_CH + s] = (i0 + s < N) ? lam[L.off_tau + i * NV + j] : 0.0;
        }
        __syncthreads();
        const bool live = (i0 + lane) < N;

        // ---- torque rows: this wavefront's items ----
        const int cnt = my->count[wave];
#pragma unroll 1
        for (int it = 0; it < cnt; it++) {
            const int code = my->code[wave][it];
            const int pa = code & 7, pb = (code >> 3) & 7, k1 = (code >> 6) & 3, k2 = (code >> 8) & 3, out = code >> 10;
#define HX(l, f) lds[HL_X + ((l) * 5 + (f)) * H_CH + lane]
#define HSG(l) (M->jtype[l] < 0 ? -1.0 : 1.0)
            if (k1 == 1 && k2 == 1) {  // two q_d seeds: the bilinear pass in plain doubles (kinova_hess_kernel.h)
                KinV kv;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    kv.xa.c[k] = 0.0; kv.xl.c[k] = 0.0; kv.ya.c[k] = 0.0; kv.yl.c[k] = 0.0;
                    kv.ca.c[k] = 0.0; kv.cl.c[k] = 0.0; kv.wa.c[k] = 0.0; kv.wl.c[k] = 0.0;
                }
                double acc = 0.0;
#pragma unroll 1
                for (int l = 0; l < NV; l++) {
                    vv_link(M, l, HX(l, 2), HSG(l) * HX(l, 3), (l == pa) ? 1.0 : 0.0, (l == pb) ? 1.0 : 0.0, lds[HL_MU + l * H_CH + lane], kv);
                    if (l >= pb) acc += vv_power(M, l, kv);
                }
                if (live) phi[(long long)out * N + i0 + lane] = acc;
                continue;
            }
            if (k1 + k2 == 1) {  // a q seed and a q_d seed: the written-out q_d variation, first-order duals in q
                const int jq = (k1 == 0) ? pa : pb, jb = (k1 == 0) ? pb : pa;
                KinQ<double> k0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    k0.va.c[k] = 0.0; k0.vl.c[k] = 0.0; k0.da.c[k] = 0.0; k0.dl.c[k] = 0.0;
                    k0.ea.c[k] = 0.0; k0.el.c[k] = 0.0; k0.wa.c[k] = 0.0; k0.wl.c[k] = 0.0;
                }
                int l = 0;
#pragma unroll 1
                for (; l < jq; l++)
                    qv_link(M, l, HX(l, 2), HSG(l) * HX(l, 3), HX(l, 0), (l == jb) ? 1.0 : 0.0, lds[HL_MU + l * H_CH + lane], k0);
                KinQ<D1> kq;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    kq.va.c[k] = D1{k0.va.c[k], 0.0}; kq.vl.c[k] = D1{k0.vl.c[k], 0.0}; kq.da.c[k] = D1{k0.da.c[k], 0.0};
                    kq.dl.c[k] = D1{k0.dl.c[k], 0.0}; kq.ea.c[k] = D1{k0.ea.c[k], 0.0}; kq.el.c[k] = D1{k0.el.c[k], 0.0};
                    kq.wa.c[k] = D1{k0.wa.c[k], 0.0}; kq.wl.c[k] = D1{k0.wl.c[k], 0.0};
                }
                double acc = 0.0;
                {
                    const double cq = HX(l, 2), sq = HX(l, 3);
                    qv_link(M, l, D1{cq, -sq}, HSG(l) * D1{sq, cq}, HX(l, 0), (l == jb) ? 1.0 : 0.0, lds[HL_MU + l * H_CH + lane], kq);
                    if (l >= pb) acc += qv_power_a(M, l, kq);
                    l++;
                }
#pragma unroll 1
                for (; l < NV; l++) {
                    qv_link(M, l, HX(l, 2), HSG(l) * HX(l, 3), HX(l, 0), (l == jb) ? 1.0 : 0.0, lds[HL_MU + l * H_CH + lane], kq);
                    if (l >= pb) acc += qv_power_a(M, l, kq);
                }
                if (live) phi[(long long)out * N + i0 + lane] = acc;
                continue;
            }
            if ((k1 == 2) != (k2 == 2)) {  // a q seed and a q_dd seed: the reduced first-order pass (kinova_hess_kernel.h)
                const int jq = (k1 == 0) ? pa : pb, ja = (k1 == 0) ? pb : pa;
                KinB<double> k0;
#pragma unroll
                for (int k = 0; k < 3; k++) { k0.ba.c[k] = 0.0; k0.bl.c[k] = 0.0; k0.wa.c[k] = 0.0; k0.wl.c[k] = 0.0; }
                int l = 0;
#pragma unroll 1
                for (; l < jq; l++)  // plain values up to the q seed
                    qa_link(M, l, HX(l, 2), HSG(l) * HX(l, 3), (l == ja) ? 1.0 : 0.0, lds[HL_MU + l * H_CH + lane], k0);
                KinB<D1> kq;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    kq.ba.c[k] = D1{k0.ba.c[k], 0.0}; kq.bl.c[k] = D1{k0.bl.c[k], 0.0};
                    kq.wa.c[k] = D1{k0.wa.c[k], 0.0}; kq.wl.c[k] = D1{k0.wl.c[k], 0.0};
                }
                double acc = 0.0;
                {   // l == jq: the seeded rotation
                    const double cq = HX(l, 2), sq = HX(l, 3);
                    qa_link(M, l, D1{cq, -sq}, HSG(l) * D1{sq, cq}, (l == ja) ? 1.0 : 0.0, lds[HL_MU + l * H_CH + lane], kq);
                    if (l >= pb) acc += qa_power_a(M, l, kq);
                    l++;
                }
#pragma unroll 1
                for (; l < NV; l++) {
                    qa_link(M, l, HX(l, 2), HSG(l) * HX(l, 3), (l == ja) ? 1.0 : 0.0, lds[HL_MU + l * H_CH + lane], kq);
                    if (l >= pb) acc += qa_power_a(M, l, kq);
                }
                if (live) phi[(long long)out * N + i0 + lane] = acc;
                continue;
            }
            const double s0 = (k1 == 0) ? 1.0 : 0.0, s1 = (k1 == 1) ? 1.0 : 0.0, s2 = (k1 == 2) ? 1.0 : 0.0;
            const double t0 = (k2 == 0) ? 1.0 : 0.0, t1 = (k2 == 1) ? 1.0 : 0.0, t2 = (k2 == 2) ? 1.0 : 0.0;
            Kin<double> k0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                k0.va.c[k] = 0.0; k0.vl.c[k] = 0.0; k0.aa.c[k] = 0.0; k0.wa.c[k] = 0.0; k0.wl.c[k] = 0.0;
                k0.al.c[k] = M->neg_gravity[k];
            }
            int l = 0;
#pragma unroll 1
            for (; l < pa; l++) hess_link(M, l, HX(l, 2), HSG(l) * HX(l, 3), HX(l, 0), HX(l, 1), lds[HL_MU + l * H_CH + lane], k0);
            KinVA<HD> kh;
            double* ws = lds + HL_WST + wave * 24 * H_CH + lane;  // this lane's column of the wavefront's w state
#define HPROM(src) \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; i_++) { \
        kh.va.c[i_] = promote(src.va.c[i_], (HD*)nullptr); kh.vl.c[i_] = promote(src.vl.c[i_], (HD*)nullptr); \
        kh.aa.c[i_] = promote(src.aa.c[i_], (HD*)nullptr); kh.al.c[i_] = promote(src.al.c[i_], (HD*)nullptr); \
    } \
    { T3<HD> wa_, wl_; \
      _Pragma("unroll") for (int i_ = 0; i_ < 3; i_++) { wa_.c[i_] = promote(src.wa.c[i_], (HD*)nullptr); wl_.c[i_] = promote(src.wl.c[i_], (HD*)nullptr); } \
      w_store(ws, 0, wa_); w_store(ws, 1, wl_); }
            if (pa < pb) {
                Kin<D1> k1s = promote<D1>(k0);
                {   // l == pa: first seed
                    const double cq = HX(l, 2), sq = HX(l, 3), sg = HSG(l);
                    hess_link(M, l, D1{cq, -sq * s0}, sg * D1{sq, cq * s0}, D1{HX(l, 0), s1}, D1{HX(l, 1), s2},
                              lds[HL_MU + l * H_CH + lane], k1s);
                    l++;
                }
#pragma unroll 1
                for (; l < pb; l++) hess_link(M, l, HX(l, 2), HSG(l) * HX(l, 3), HX(l, 0), HX(l, 1), lds[HL_MU + l * H_CH + lane], k1s);
                HPROM(k1s)
            } else {
                HPROM(k0)
            }
#undef HPROM
            double phi_ab;
            {   // l == pb: second seed (and the first one too when both sit on this joint)
                const double cq = HX(l, 2), sq = HX(l, 3), sg = HSG(l);
                const double da = (pa == pb) ? s0 : 0.0, ea = (pa == pb) ? s1 : 0.0, fa = (pa == pb) ? s2 : 0.0;
                hess_link_hd(M, l, HD{cq, -sq * da, -sq * t0, -cq * da * t0}, sg * HD{sq, cq * da, cq * t0, -sq * da * t0},
                             HD{HX(l, 0), ea, t1, 0.0}, HD{HX(l, 1), fa, t2, 0.0}, lds[HL_MU + l * H_CH + lane], kh, ws);
                phi_ab = hess_power_ab_lds(M, l, kh, ws);
                l++;
            }
#pragma unroll 1
            for (; l < NV; l++) {
                hess_link_hd(M, l, HX(l, 2), HSG(l) * HX(l, 3), HX(l, 0), HX(l, 1), lds[HL_MU + l * H_CH + lane], kh, ws);
                phi_ab += hess_power_ab_lds(M, l, kh, ws);
            }
#undef HX
#undef HSG
            if (live) phi[(long long)out * N + i0 + lane] = phi_ab;
        }
        __syncthreads();  // every wavefront is done with its w state: the world-frame tables take the same LDS

        if (wave == 0 && O > 0 && do_obs) {  // world-frame joint axes / origins and sphere-link poses, lane = sample
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

        __syncthreads();  // world-frame tables complete; nothing reads the sample tables of this pass any more

        // ---- obstacle rows: a wavefront per sphere, lane = sample; the four partial sums meet in LDS ----
        {
            double acc[H_PAIRS];
#pragma unroll
            for (int k = 0; k < H_PAIRS; k++) acc[k] = 0.0;
            if (O > 0 && do_obs) {
#pragma unroll 1
                for (int sp = wave; sp < ns; sp += H_WAVES) {
                    const double wgt = live ? lam[L.off_obs + (i0 + lane) * ns + sp] : 0.0;
                    hess_sphere_terms(lds, lane, S, boxes, O, sp, wgt, acc);
                }
            }
            // the wavefronts add their sums one after the other (a fixed order: deterministic), 14 KB instead of four copies
#pragma unroll 1
            for (int w = 0; w < H_WAVES; w++) {
                if (wave == w) {
#pragma unroll
                    for (int k = 0; k < H_PAIRS; k++) {
                        double* c = lds + HB_SUM + k * H_CH + lane;
                        *c = (w == 0) ? acc[k] : *c + acc[k];
                    }
                }
                __syncthreads();
            }
        }
        for (int idx = tid; idx < H_PAIRS * H_CH; idx += RAPTOR_BLOCK) {
            const int s = idx & (H_CH - 1), k = idx >> 6;
            if (do_obs && i0 + s < N) phi[(long long)(HB_OBS + k) * N + i0 + s] = lds[HB_SUM + idx];
        }
        __syncthreads();  // the sample tables are rewritten by the next pass
    }
}

// H[b][c1 <= c2] from the per-sample blocks and the column weights (PiecewiseBezierCurves layout: W[i][order][c]).
// `pairs[p]` = (c1 | c2 << 16, first | end << 16): the p-th entry of the upper triangle (row by row) and the sample range over which
// BOTH columns have a non-zero weight (a control point shapes two pieces only: most pairs interact over no sample at all).
// One workgroup per z walks the samples in chunks of CH: the chunk of Phi ([182][CH + 1], padded against bank conflicts) and of W are
// staged in LDS once, coalesced, then thread t owns the pairs p = t, t + 256, ... and adds the chunk's part of its sums -- one
// running sum per pair in ascending sample order (carried through H between chunks by the same thread), so the result does not
// depend on the chunking.  Phi is read from HBM exactly once.
__global__ __launch_bounds__(RAPTOR_BLOCK) void kinova_hess_contract_kernel(const DevTraj* __restrict__ T, const double* __restrict__ basis,
                                                                             const int2* __restrict__ pairs, const int P, const int CH,
                                                                             const double* __restrict__ Phi, double* __restrict__ Hout) {
    extern __shared__ double lc[];
    const int tid = threadIdx.x;
    const long long b = blockIdx.x;
    const int N = T->N, V = T->varLength, NV = T->Nact;
    const int PS = CH + 1;
    double* lphi = lc;                   // [HB_NI][CH + 1]
    double* lw = lc + HB_NI * PS;        // [CH][3][V]
    const double* phi = Phi + b * (long long)HB_NI * N;
    double* H = Hout + b * (long long)P;
    const int sh = 31 - __clz(CH);       // CH is a power of two
#pragma unroll 1
    for (int i0 = 0; i0 < N; i0 += CH) {
        const int cnt = min(CH, N - i0);
        if (i0 > 0) __syncthreads();
        for (int idx = tid; idx < HB_NI * CH; idx += RAPTOR_BLOCK) {
            const int row = idx >> sh, s = idx & (CH - 1);
            lphi[row * PS + s] = (s < cnt) ? phi[(long long)row * N + i0 + s] : 0.0;
        }
        for (int idx = tid; idx < cnt * 3 * V; idx += RAPTOR_BLOCK) lw[idx] = basis[(size_t)(i0 * 3) * V + idx];
        __syncthreads();
#pragma unroll 1
        for (int p = blockIdx.y * RAPTOR_BLOCK + tid; p < P; p += gridDim.y * RAPTOR_BLOCK) {  // grid y > 1: small batches
            const int2 pr = pairs[p];
            const int c1 = pr.x & 0xffff, c2 = pr.x >> 16;
            const int lo = max(pr.y & 0xffff, i0) - i0, hi = min(pr.y >> 16, i0 + cnt) - i0;
            if (lo >= hi) {
                if (i0 == 0) H[p] = 0.0;
                continue;
            }
            const int j1 = c1 % NV, j2 = c2 % NV;
            const int sym = pair_index(min(j1, j2), max(j1, j2));
            const double* Pqq = lphi + (HB_QQ + sym) * PS;
            const double* Pob = lphi + (HB_OBS + sym) * PS;
            const double* Pvv = lphi + (HB_VV + sym) * PS;
            const double* Pqv12 = lphi + (HB_QV + j1 * NV + j2) * PS;
            const double* Pqv21 = lphi + (HB_QV + j2 * NV + j1) * PS;
            const double* Pqa12 = lphi + (HB_QA + j1 * NV + j2) * PS;
            const double* Pqa21 = lphi + (HB_QA + j2 * NV + j1) * PS;
            double s = (i0 == 0) ? 0.0 : H[p];
            for (int i = lo; i < hi; i++) {
                const double* W = lw + i * 3 * V;
                const double a0 = W[c1], a1 = W[V + c1], a2 = W[2 * V + c1];
                const double b0 = W[c2], b1 = W[V + c2], b2 = W[2 * V + c2];
                s += a0 * b0 * (Pqq[i] + Pob[i]) + a1 * b1 * Pvv[i] + a0 * b1 * Pqv12[i] + a1 * b0 * Pqv21[i] + a0 * b2 * Pqa12[i] +
                     a2 * b0 * Pqa21[i];
            }
            H[p] = s;
        }
    }
}

#endif
