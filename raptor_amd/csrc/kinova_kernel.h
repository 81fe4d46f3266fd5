// K_kinova : the whole Kinova constraint evaluation for one sample-eval in ONE 8-lane group
// (eval_g / eval_jac_g of KinovaOptimizer: joint, velocity, torque limits + sphere/box obstacle rows).
//
// MI355X design (not a translation of the reference's per-class loops):
//   * a 256-lane workgroup owns 32 sample-evals; each sample-eval (b, i) is worked on by 8 adjacent
//     lanes of one wavefront, so the lanes of a group exchange data through LDS / wave shuffles
//     without any workgroup barrier;
//   * lane g owns joint g:  trajectory value + sincos (phase 0), its link's world-frame kinematics
//     (phase A: the link recursion runs with a per-lane trip count, lane g stops at link g),
//     its link's spatial inertia, momentum, wrench and Coriolis term (phase B);
//   * composite (subtree) quantities are an 8-lane suffix scan with wave shuffles (phase C);
//   * d(tau)/dz comes from the analytic world-frame identities below, NOT from nv-column forward
//     mode: with s_k the joint axis, sd_k = v_k x s_k, psi_k = a_k x s_k + v_k x sd_k, composite
//     inertia I^C_j, momentum H^C_j, wrench F_j and Q^C_j = sum_{l in subtree(j)} (v_l x*) I_l + transpose,
//         column k, row j >= k :  IS_j . gamma_k + BtS_j . beta_k        (IS = I^C s, BtS = Q^C s - s x* H^C)
//         column k, row j <  k :  s_j . (w2 IS_k + w1 (2 I^C sd + Q^C s + s x* H^C) + w0 (I^C psi + Q^C sd + sd x* H^C + s x* F))_k
//         gamma_k = w2 s + 2 w1 sd + w0 psi,  beta_k = w1 s + w0 sd,  (w0,w1,w2) = (pq_pz, pq_d_pz, pq_dd_pz)(k,k)
//     (derivation and numpy check: scripts/proto_world_derivs.py, scripts/proto_reduced_form.py;
//      this equals InverseDynamics.cpp:137-139 prnea_pq*pq_pz + prnea_pv*pv_pz + prnea_pa*pa_pz);
//   * the 17 (+3) collision spheres are dealt round-robin to the 8 lanes; the sphere Jacobian needs no
//     extra kinematics because d(centre)/dq_c . diff = s_c . [centre x diff ; diff];
//   * every Jacobian row is staged in LDS and leaves the workgroup as 16-byte-per-lane contiguous
//     stores (the per-e rows of one constraint block are contiguous in Optimizer order).
#ifndef RAPTOR_KINOVA_KERNEL_H
#define RAPTOR_KINOVA_KERNEL_H

#include "kernels.h"

#define KG 8       // lanes per sample-eval
#ifndef KIN_BLOCK
#define KIN_BLOCK 64   // lanes per workgroup: one wavefront (measured 64 / 128 / 256 / 512 lanes: 0.52 / 0.52 / 0.525 / 0.62 ms; run-to-run noise 1 %)
#endif
#ifndef KIN_WGS
#define KIN_WGS 3      // waves per SIMD asked of the register allocation (168 VGPRs; 13.3 KB of LDS per one-wavefront workgroup: twelve share a CU)
#endif
#define K_EPB (KIN_BLOCK / KG)   // sample-evals per workgroup
#define K_WAVES (KIN_BLOCK / 64)
// LDS layout, in doubles.
//   header (whole workgroup, staged once): the sphere tables.  (The box loop reads boxes through uniform scalar loads; the
//   winner's box, whose index differs between lanes, comes through the vector L1 -- a copy of the box table in this header was
//   measured slower once two spheres per lane hide that latency, 0.580 against 0.572 ms.)
#define KH_SPH 0                       // [sphere][4]: offset x y z, radius
// then [sphere] two ints (joint id, pose slot) from 4 * ns on; the header is 5 * ns doubles rounded up to an even count (16-byte aligned
// wave areas), sized by the problem's own sphere count so that TWELVE wavefronts' areas fit the CU's 160 KB (three per SIMD)
__host__ __device__ inline int kinova_header_doubles(int ns) { return (5 * ns + 1) & ~1; }
//   then one area per wavefront (8 sample-evals, el = 0..7):
//     el * KW_E  + KL_S   s_j                     7 x 6
//     el * KW_E  + KL_RO  first two columns of R and o of up to 4 sphere-carrying links   4 x 9  (the third column is their cross product)
//     KW_X + el * KW_XE   IS_j, BtS_j             7 x 12        (phase D -> E)
//     KW_X + el * 7 ns    sphere rows [sphere][7] (after phase E: the rows of the wave's 8 sample-evals are CONTIGUOUS, in the
//                         order they have in pg_pz, so the copy-out is a straight 16-byte-per-lane stream with no index arithmetic)
#define KL_S 0
#define KL_RO 42
#define KL_ROS 9   // doubles per pose slot
#define KW_E 78
#define KW_X (8 * KW_E)
#define KW_XE 84
__host__ __device__ inline int kinova_wave_doubles(int ns) { return KW_X + 8 * ((7 * ns > KW_XE) ? 7 * ns : KW_XE); }
__host__ __device__ inline int kinova_lds_doubles(int ns) { return kinova_header_doubles(ns) + K_WAVES * kinova_wave_doubles(ns); }

RD void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inertia about the world origin: mass, first moment hv = m c, rotational inertia IO (symmetric)
struct InertiaO {
    double m;
    V3 hv;
    double xx, xy, xz, yy, yz, zz;
};
RD V3 symmul(double xx, double xy, double xz, double yy, double yz, double zz, V3 a) {
    return V3{xx * a.x + xy * a.y + xz * a.z, xy * a.x + yy * a.y + yz * a.z, xz * a.x + yz * a.y + zz * a.z};
}
RD SV applyO(const InertiaO& I, SV x) {
    return SV{symmul(I.xx, I.xy, I.xz, I.yy, I.yz, I.zz, x.a) + cross(I.hv, x.l), x.l * I.m + cross(x.a, I.hv)};
}
// Q x,  Q = [[QTL, [t]x], [-[t]x, 0]]
RD SV applyQ(const double* Q, V3 t, SV x) {
    return SV{symmul(Q[0], Q[1], Q[2], Q[3], Q[4], Q[5], x.a) + cross(t, x.l), cross(x.a, t)};
}
// Cross-lane moves inside a 16-lane row by DPP (a VALU modifier: no LDS round trip, unlike ds_bpermute behind __shfl_*).
// The 8-lane groups are half rows, so a shift by D < 8 never leaves the row for the lanes that use the value; lanes whose
// source falls outside their own group must be masked by the caller.
//   dpp_down<D>(x): lane l receives x of lane l + D   (row_shl:D)
//   dpp_up<D>(x)  : lane l receives x of lane l - D   (row_shr:D)
template <int CTRL>
RD double dpp_mov64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int D> RD double dpp_down(double x) { return dpp_mov64<0x100 + D>(x); }
template <int D> RD double dpp_up(double x) { return dpp_mov64<0x110 + D>(x); }
template <int D> RD V3 dpp_up(V3 a) { return V3{dpp_up<D>(a.x), dpp_up<D>(a.y), dpp_up<D>(a.z)}; }
template <int D> RD SV dpp_up(SV a) { return SV{dpp_up<D>(a.a), dpp_up<D>(a.l)}; }

// rigid transform of a link frame (rotation R, origin p) and the composition parent * child
struct XF {
    M3 R;
    V3 p;
};
RD XF compose(const XF& a, const XF& b) {
    XF c;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c.R.m[3 * i + j] = a.R.m[3 * i] * b.R.m[j] + a.R.m[3 * i + 1] * b.R.m[3 + j] + a.R.m[3 * i + 2] * b.R.m[6 + j];
    c.p = a.p + mul(a.R, b.p);
    return c;
}
template <int D> RD XF dpp_up(const XF& a) {
    XF c;
#pragma unroll
    for (int k = 0; k < 9; k++) c.R.m[k] = dpp_up<D>(a.R.m[k]);
    c.p = dpp_up<D>(a.p);
    return c;
}
// one round of the inclusive prefix scans of phase A over the 8 lanes of a group (Hillis-Steele, distance D)
template <int D> RD void scan_xf(XF& x, int g) {
    const XF left = dpp_up<D>(x);
    if (g >= D) x = compose(left, x);
}
template <int D> RD void scan_sv(SV& x, int g) {
    const SV left = dpp_up<D>(x);
    if (g >= D) x = x + left;
}

// ---- BoxCollisionAvoidance (Constraints/src/BoxCollisionAvoidance.cpp:92-137, 290-373) ----
// Running arg-min over the boxes.  `key` = sign * |diff|^2 (0 below SQUARE_ROOT_THRESHOLD) orders candidates exactly as the
// reference's dist = sign * sqrt(|diff|^2) does whenever the keys differ by more than rounding; only near-ties take the two
// square roots the reference compares, so first-minimum-wins is kept bit for bit while the box loop carries no sqrt.
struct BoxBest {
    double key, d2, sign;
    V3 dl;    // point - closest point, in the box frame
    int idx;
};
RD double box_signed_distance(double d2, double sign) { return (d2 > 1e-12) ? sign * sqrt(d2) : 0.0; }  // SQUARE_ROOT_THRESHOLD
RD double box_best_distance(const BoxBest& b) { return box_signed_distance(b.d2, b.sign); }
RD void box_candidate(const DevBox* __restrict__ bx, int ob, V3 point, BoxBest& best) {
    M3 Rb;
#pragma unroll
    for (int k = 0; k < 9; k++) Rb.m[k] = bx->R[k];
    const V3 lp = mulT(Rb, point - v3(bx->center[0], bx->center[1], bx->center[2]));
    const double lpv[3] = {lp.x, lp.y, lp.z};
    double lc[3], dface[3][2];
    double isInside = -1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double hs = bx->size[k] / 2.0;
        dface[k][0] = lpv[k] + hs;
        dface[k][1] = hs - lpv[k];
        lc[k] = fmax(-hs, fmin(lpv[k], hs));
        if (lc[k] != lpv[k]) isInside = 1.0;
    }
    if (isInside == -1.0) {  // inside: snap to the nearest face, first minimum wins (:112-131)
        double md = 1e19;
        int face = 0, dir = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (dface[k][0] < md) { md = dface[k][0]; face = k; dir = 0; }
            if (dface[k][1] < md) { md = dface[k][1]; face = k; dir = 1; }
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (k == face) lc[k] = (dir == 0) ? -bx->size[k] / 2.0 : bx->size[k] / 2.0;
    }
    // |point - closest| is rotation invariant: measure it in the box frame
    const V3 dl = v3(lpv[0] - lc[0], lpv[1] - lc[1], lpv[2] - lc[2]);
    const double d2 = dot(dl, dl);
    const double key = (d2 > 1e-12) ? isInside * d2 : 0.0;
    const double diff = key - best.key;
    bool closer = diff < 0.0;
    if (fabs(diff) <= 1e-14 * fmax(fabs(key), fabs(best.key)))  // near-tie: the reference's own comparison (:305-308, strict)
        closer = box_signed_distance(d2, isInside) < box_signed_distance(best.d2, best.sign);
    if (closer) {
        best.key = key;
        best.idx = ob;
        best.d2 = d2;
        best.sign = isInside;
        best.dl = dl;
    }
}
// gradient of the arg-min box (:359-372) as a wrench: the clamped-axis mask of the reference drops out (diff has no
// component along unclamped axes), d dist/dz_c = sign * diff . dp/dz_c / |diff| and diff . (w x (p - o)) = s_c . [p x diff ; diff]
RD SV box_gradient_wrench(const DevBox* __restrict__ bx, const BoxBest& best, V3 point, double w0) {
    M3 Rb;
#pragma unroll
    for (int k = 0; k < 9; k++) Rb.m[k] = bx->R[k];
    const double scale = (best.d2 > 1e-12) ? best.sign * w0 / sqrt(best.d2) : 0.0;
    const V3 dw = mul(Rb, best.dl) * scale;
    return SV{cross(point, dw), dw};
}

// ---- the same distance in the form the fused kernel's box loop uses ----
// e_k = |lp_k| - hs_k per box axis (lp = R^T point - R^T centre).  Outside (some e_k > 0): |diff|^2 = sum max(e_k, 0)^2 -- the
// reference's lp_k - clamp(lp_k) is exactly +-max(e_k, 0).  Inside: the nearest face is the first axis of largest e_k (the
// reference's first-minimum scan over x-, x+, y-, y+, z-, z+ of hs_k -+ lp_k), |diff| = -e_k.  The loop orders boxes by the signed
// square key = +-|diff|^2 (0 below SQUARE_ROOT_THRESHOLD), monotone in the reference's sign * sqrt(|diff|^2): strict `<` keeps
// first-minimum-wins on ties, and only (key, index) live through the loop; the winner's direction is rebuilt once per sphere.
RD void box_local(const double* __restrict__ R, const double* __restrict__ cl, const double* __restrict__ hs, V3 p, V3& lp, V3& e) {
    // (the centre is subtracted last: one scalar operand per instruction, no copies of box constants into vector registers)
    lp.x = (R[0] * p.x + (R[3] * p.y + R[6] * p.z)) - cl[0];
    lp.y = (R[1] * p.x + (R[4] * p.y + R[7] * p.z)) - cl[1];
    lp.z = (R[2] * p.x + (R[5] * p.y + R[8] * p.z)) - cl[2];
    e = v3(fabs(lp.x) - hs[0], fabs(lp.y) - hs[1], fabs(lp.z) - hs[2]);
}
RD double box_key(const DevBox* __restrict__ bx, V3 p) {
    V3 lp, e;
    box_local(bx->R, bx->cl, bx->hs, p, lp, e);
    const double emax = fmax(e.x, fmax(e.y, e.z));
    const double mx = fmax(e.x, 0.0), my = fmax(e.y, 0.0), mz = fmax(e.z, 0.0);
    const double d2o = mx * mx + (my * my + mz * mz), d2i = emax * emax;
    const double key = (emax > 0.0) ? d2o : -d2i;
    return (fabs(key) > 1e-12) ? key : 0.0;
}
// sqrt(x) and 1 / sqrt(x) together for x in the normal range (here 1e-12 < x < 1e4): the hardware reciprocal square root
// refined by the coupled Newton steps a correctly rounded sqrt uses; both results are good to ~1 ulp
RD void sqrt_rsqrt(double x, double& r, double& ir) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    e = fma(-h, g, 0.5);
    h = fma(h, e, h);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    r = g;
    ir = h + h;
}
struct BoxHit {
    double dist;  // signed distance (BoxCollisionAvoidance.cpp:300-303)
    V3 n;         // d dist / d point in world axes (zero below the threshold, :359-372)
};
// b15 = R (9), R^T centre (3), half sizes (3) of the winning box: the head of a DevBox
RD BoxHit box_hit(const double* __restrict__ b15, V3 p) {
    double R[9], cl[3], hs[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = b15[k];
#pragma unroll
    for (int k = 0; k < 3; k++) { cl[k] = b15[9 + k]; hs[k] = b15[12 + k]; }
    V3 lp, e;
    box_local(R, cl, hs, p, lp, e);
    const double emax = fmax(e.x, fmax(e.y, e.z));
    const bool outside = emax > 0.0;
    const int kmax = (e.x >= e.y && e.x >= e.z) ? 0 : (e.y >= e.z ? 1 : 2);
    const double mx = fmax(e.x, 0.0), my = fmax(e.y, 0.0), mz = fmax(e.z, 0.0);
    V3 dl;  // point - closest point, box frame
    dl.x = outside ? (lp.x > 0.0 ? mx : -mx) : (kmax == 0 ? (lp.x > 0.0 ? e.x : -e.x) : 0.0);
    dl.y = outside ? (lp.y > 0.0 ? my : -my) : (kmax == 1 ? (lp.y > 0.0 ? e.y : -e.y) : 0.0);
    dl.z = outside ? (lp.z > 0.0 ? mz : -mz) : (kmax == 2 ? (lp.z > 0.0 ? e.z : -e.z) : 0.0);
    const double d2 = outside ? mx * mx + (my * my + mz * mz) : emax * emax;
    BoxHit h;
    h.dist = 0.0;
    h.n = v3(0, 0, 0);
    if (d2 > 1e-12) {
        double r, ir;
        sqrt_rsqrt(d2, r, ir);
        const double sgn = outside ? 1.0 : -1.0;
        h.dist = sgn * r;
        const double sc = sgn * ir;
        h.n = v3(R[0] * dl.x + R[1] * dl.y + R[2] * dl.z, R[3] * dl.x + R[4] * dl.y + R[5] * dl.z, R[6] * dl.x + R[7] * dl.y + R[8] * dl.z) * sc;
    }
    return h;
}
// one sphere of KinovaCustomizedConstraints: its tables from the LDS header, its centre from the published link pose
struct SphereInfo {
    int jid;
    double radius;
    V3 point;
};
RD SphereInfo sphere_info(const double* hdr, int ns, const double* se, int sp) {
    SphereInfo si;
    const int2 js = reinterpret_cast<const int2*>(hdr + 4 * ns)[sp];
    const double* t = hdr + KH_SPH + 4 * sp;
    const double* ro = se + KL_RO + js.y * KL_ROS;
    const V3 c0 = v3(ro[0], ro[1], ro[2]), c1 = v3(ro[3], ro[4], ro[5]);
    const V3 c2 = cross(c0, c1);  // third column of the link's rotation
    si.jid = js.x;
    si.radius = t[3];
    si.point = v3(ro[6], ro[7], ro[8]) + (c0 * t[0] + c1 * t[1] + c2 * t[2]);
    return si;
}
// Wave-level copy-out: `len` doubles to dst (8-byte aligned), 16 bytes per lane, from a contiguous LDS run
RD void wave_store_run(double* __restrict__ dst, const double* src, int len, int lane) {
    const int head = (int)(((size_t)dst >> 3) & 1);  // 16-byte alignment of the run
    if (head && lane == 0 && len > 0) dst[0] = src[0];
    const int pairs = (len - head) >> 1;
    double2* dst2 = reinterpret_cast<double2*>(dst + head);
    for (int pidx = lane; pidx < pairs; pidx += 64) {
        const int t = head + 2 * pidx;
        dst2[pidx] = make_double2(src[t], src[t + 1]);
    }
    if (((len - head) & 1) && lane == 63) dst[len - 1] = src[len - 1];
}
template <int CTRL> RD int dpp_mov32(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }

// RECORD (the longer-horizon path, kinova_expand_kernel): instead of Jacobian rows the kernel leaves a per-sample-eval record in
// `JAC` -- d tau / d(q, q_d, q_dd) as three 7 x 7 blocks and d dist / d q per sphere -- because PiecewiseBezierCurves columns
// carry their own weight triples; the g rows are written as usual.
#define KREC_D 0                                   // D[x][j][k] at x * 49 + j * 7 + k   (x = q, q_d, q_dd; row j, joint k)
#define KREC_G 147                                 // gq[sphere][k]
#define KREC_STRIDE (147 + 7 * RAPTOR_MAX_SPHERES) // 315 doubles
// Phase accounting: a build with -DKIN_STOP=k ends the kernel after phase k, storing a checksum of what is live
// there so that the phase is not optimised away.  Measurement builds only (scripts/kinova_phases.sh); never defined in the product.
#ifdef KIN_STOP
#define KIN_PHASE_END(k, expr) if (KIN_STOP == (k)) { if (own && G != nullptr) Gb[ro + L.off_tau + i * NV + g] = (expr); return; }
#else
#define KIN_PHASE_END(k, expr)
#endif
RD double sv_sum(SV x) { return x.a.x + x.a.y + x.a.z + x.l.x + x.l.y + x.l.z; }
template <bool DERIV, bool RECORD = false>
__global__ __launch_bounds__(KIN_BLOCK, KIN_WGS) void kinova_fused_kernel(
    const DevModel* __restrict__ M, const DevTraj* __restrict__ T, const double* __restrict__ basis,
    const double* __restrict__ wsum, const DevSpheres* __restrict__ S, const DevBox* __restrict__ boxes,
    const DevLayout L, const double* __restrict__ Z, const long long E, double* __restrict__ G,
    double* __restrict__ JAC) {
    extern __shared__ double lds[];
    constexpr int NV = 7;
    const int tid = threadIdx.x;
    const int g = tid & (KG - 1);
    const int lane = tid & 63, wv = tid >> 6;
    const int el = (tid >> 3) & 7;  // sample-eval within the wave
    const long long e0 = (long long)blockIdx.x * K_EPB;
    const long long ew0 = e0 + wv * 8;  // first sample-eval of this wave
    const long long e = ew0 + el;
    const int N = T->N;
    const int ns = S->num_spheres;
    const int O = S->num_obstacles;
    const int rowlen = NV * ns;  // doubles of sphere rows per sample-eval
    double* wbase = lds + kinova_header_doubles(ns) + wv * kinova_wave_doubles(ns);
    double* se = wbase + el * KW_E;          // s_j, link poses
    double* sx = wbase + KW_X + el * KW_XE;  // IS_j, BtS_j
    double* srow = wbase + KW_X + el * rowlen;

    // ---- stage the sphere tables in the LDS header (one barrier, at the start where the waves are aligned) ----
    for (int t = tid; t < ns; t += KIN_BLOCK) {
        double* d = lds + KH_SPH + 4 * t;
        d[0] = S->offset[t][0]; d[1] = S->offset[t][1]; d[2] = S->offset[t][2]; d[3] = S->radius[t];
        reinterpret_cast<int2*>(lds + 4 * ns)[t] = make_int2(S->joint_id[t], S->slot[t]);
    }
    __syncthreads();

    // sample index / decision vector of this lane's sample-eval: ONE scalar 64-bit division per wavefront (its first sample-eval),
    // then a wrap per lane -- a per-lane e / N, e % N is ~100 VALU instructions of the wave's 3400
    const long long ew0u = ((long long)__builtin_amdgcn_readfirstlane((int)(ew0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)ew0);
    const long long b0v = ew0u / N;
    const long long b0 = ((long long)__builtin_amdgcn_readfirstlane((int)(b0v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b0v);
    const int i0 = (int)(ew0u - b0 * N);
    int i = i0 + el;
    int wrap = 0;
    while (i >= N) { i -= N; wrap++; }  // (N >= 8 in every configuration of the reference: at most once)
    const long long b = b0 + wrap;
    // the run structure of this wave's sample-evals (wave-uniform): they start at sample i0 of decision vector b0
    const int cntw = (int)((E - ew0u < 8) ? ((E - ew0u > 0) ? E - ew0u : 0) : 8);

    // ---- every global load of the dynamics phases is issued (and awaited) HERE, before the first store: the vector-memory
    // counter of gfx9 retires loads and stores in issue order, so a load behind a store waits for that store's round trip to HBM ----
    const bool live = e < E;
    const int gl_ = (g < NV) ? g : NV - 1;
    double pq = 0.0, pqd = 0.0, pqdd = 0.0;
    if (live) traj_joint(T, basis, Z + b * L.n, i, gl_, pq, pqd, pqdd);
    int pjt = M->jtype[gl_];
    double pplR[9], pplp[3], plev[3], pIc[6];
#pragma unroll
    for (int k = 0; k < 9; k++) pplR[k] = M->plR[gl_][k];
#pragma unroll
    for (int k = 0; k < 3; k++) { pplp[k] = M->plp[gl_][k]; plev[k] = M->lever[gl_][k]; }
#pragma unroll
    for (int k = 0; k < 6; k++) pIc[k] = M->Ic[gl_][k];
    double pmass = M->mass[gl_], parm = M->armature[gl_], pdamp = M->damping[gl_], pfric = M->friction[gl_];
    int pslot = S->ro_slot[gl_];
    double pw0 = 1.0, pw1 = 0.0, pw2 = 0.0;
    if (DERIV && !RECORD && live) { pw0 = wsum[i * 3]; pw1 = wsum[i * 3 + 1]; pw2 = wsum[i * 3 + 2]; }
    // the terms of tau and of its diagonal Jacobian entry that involve only this lane's own joint: formed now, so that qd, qdd and
    // the three joint constants do not stay live into the register peak of phase D (InverseDynamics.cpp:115-121)
    double ptau_own = parm * pqdd + pdamp * pqd + pfric * sign_eps(pqd);
    double pdiag_own = parm * pw2 + pdamp * pw1;
#define PIN(x) asm volatile("" : "+v"(x))
    PIN(pq); PIN(pqd); PIN(pqdd); PIN(pmass); PIN(parm); PIN(pdamp); PIN(ptau_own); PIN(pdiag_own); PIN(pw0); PIN(pw1); PIN(pw2);
#pragma unroll
    for (int k = 0; k < 9; k++) PIN(pplR[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) { PIN(pplp[k]); PIN(plev[k]); }
#pragma unroll
    for (int k = 0; k < 6; k++) PIN(pIc[k]);
    asm volatile("" : "+v"(pjt), "+v"(pslot));
#undef PIN

    if (e < E) {
        const double* z = Z + b * L.n;  // L.n == 7 for ArmourBezierCurves
        // row addresses = a wave-uniform 64-bit base (decision vector b0) + a 32-bit per-lane offset (a lane of the next decision
        // vector adds m): no 64-bit per-lane row index stays live through the kernel
        double* const Gb = G + (size_t)b0 * L.m;
        double* const Jb = JAC + (size_t)b0 * L.m * NV;
        const int ro = wrap * L.m;
        const bool own = g < NV;  // lane 7 owns no joint
        const int gl = own ? g : NV - 1;

        // ---- phase 0: joint g's trajectory value and sincos ----
        const double q = pq, qd = pqd, qdd = pqdd;
        double sq, cq;
        sincos_joint(q, &sq, &cq);
        if (own && G != nullptr) {  // JointLimits.cpp:34, VelocityLimits.cpp:33
            Gb[ro + L.off_q + i * NV + g] = q;
            Gb[ro + L.off_qd + i * NV + g] = qd;
        }
        if (DERIV && !RECORD && own) {
            // pq_pz / pq_d_pz rows (JointLimits.cpp:40-44, VelocityLimits.cpp:39-43; w * identity per sample, ArmourBezierCurves.cpp:147-151)
            // straight from registers, like the torque rows of phase E: lane g writes column g of the 7 x 7 block, the 7 lanes of a
            // group one 56-byte row per store
            double* qrow = Jb + (ro + L.off_q + i * NV) * NV + g;
            double* vrow = Jb + (ro + L.off_qd + i * NV) * NV + g;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                qrow[j * NV] = (j == g) ? pw0 : 0.0;
                vrow[j * NV] = (j == g) ? pw1 : 0.0;
            }
        }
        KIN_PHASE_END(1, q + qd + qdd + sq + cq)

        // ---- phase A: world-frame kinematics of link g as prefix scans over the group's 8 lanes ----
        // lane g forms its joint's transform in the parent link's frame, X_g = placement_g * Rot/Trans(q_g); an inclusive
        // scan of rigid-transform products gives the world pose of every link in three rounds.  The joint axis is a column
        // of that pose, v_g = sum_{k <= g} s_k qd_k and a_g = -a_grav + sum_{k <= g} (s_k qdd_k + (v_k x s_k) qd_k) are two
        // more scans of 6-vectors ((v_{k-1} x s_k qd_k) = (v_k x s_k) qd_k because s_k x s_k = 0).
        const int jt = pjt;
        const int aj = jt < 0 ? -jt : jt;
        const double sg = jt < 0 ? -1.0 : 1.0;
        XF X;
        SV s;
        {
            M3 P;
#pragma unroll
            for (int k = 0; k < 9; k++) P.m[k] = pplR[k];
            V3 o = v3(pplp[0], pplp[1], pplp[2]);
            joint_world(jt, q, cq, sq, P, X.R, o, s);  // the axis it returns is in the parent's frame: recomputed below
            X.p = o;
            if (!own) {  // lane 7 owns no joint: identity, so that it ends up with link 6's pose (and zero mass below)
#pragma unroll
                for (int k = 0; k < 9; k++) X.R.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
                X.p = v3(0, 0, 0);
            }
        }
        scan_xf<1>(X, g);
        scan_xf<2>(X, g);
        scan_xf<4>(X, g);
        const M3 Rw = X.R;
        const V3 ow = X.p;
        {
            // a rotation about local axis k leaves column k of the frame alone: the world axis is a column of the link's own pose
            const V3 ax = col_sel(Rw, (aj <= 3 ? aj : aj - 3) - 1) * (own ? sg : 0.0);
            s = (aj <= 3) ? SV{ax, cross(ow, ax)} : SV{v3(0, 0, 0), ax};
        }
        SV v = s * qd;
        scan_sv<1>(v, g);
        scan_sv<2>(v, g);
        scan_sv<4>(v, g);
        const SV sd = crm(v, s);  // = v x s
        SV a = s * qdd + sd * qd;
        scan_sv<1>(a, g);
        scan_sv<2>(a, g);
        scan_sv<4>(a, g);
        a.l = a.l + v3(M->neg_gravity[0], M->neg_gravity[1], M->neg_gravity[2]);
        if (own) {
            lds_put(se + KL_S + g * 6, s);
            const int slot = pslot;
            if (slot >= 0) {
                double* ro = se + KL_RO + slot * KL_ROS;
                ro[0] = Rw.m[0]; ro[1] = Rw.m[3]; ro[2] = Rw.m[6];  // column 0
                ro[3] = Rw.m[1]; ro[4] = Rw.m[4]; ro[5] = Rw.m[7];  // column 1
                ro[6] = ow.x; ro[7] = ow.y; ro[8] = ow.z;
            }
        }

        KIN_PHASE_END(2, sv_sum(v) + sv_sum(a) + sv_sum(s) + sv_sum(sd) + Rw.m[0] + Rw.m[4] + Rw.m[8] + Rw.m[1] + Rw.m[5] + Rw.m[6] + ow.x + ow.y + ow.z)

        // ---- phase B: link g's inertia, momentum, wrench, Coriolis term (world frame, about the origin) ----
        InertiaO I;
        SV h, f;
        double Qt[6];
        {
            Inertia Ic;
            const V3 c = ow + mul(Rw, v3(plev[0], plev[1], plev[2]));
            rotate_inertia(Rw, pIc, Ic);
            const double m = own ? pmass : 0.0;
            const double keep = own ? 1.0 : 0.0;
            I.m = m;
            I.hv = c * m;
            const double cc = dot(c, c);
            I.xx = keep * Ic.xx + m * (cc - c.x * c.x);
            I.yy = keep * Ic.yy + m * (cc - c.y * c.y);
            I.zz = keep * Ic.zz + m * (cc - c.z * c.z);
            I.xy = keep * Ic.xy - m * c.x * c.y;
            I.xz = keep * Ic.xz - m * c.x * c.z;
            I.yz = keep * Ic.yz - m * c.y * c.z;
            h = applyO(I, v);
            f = applyO(I, a) + crf(v, h);
            if (DERIV) {
                // QTL = W + W^T - (hv u^T + u hv^T) + 2 (u.hv) 1,   W = [w]x IO
                const V3 w = v.a, u = v.l;
                const V3 c0 = cross(w, v3(I.xx, I.xy, I.xz));  // W[:,0]
                const V3 c1 = cross(w, v3(I.xy, I.yy, I.yz));
                const V3 c2 = cross(w, v3(I.xz, I.yz, I.zz));
                const double uh = 2.0 * dot(u, I.hv);
                Qt[0] = 2.0 * c0.x - 2.0 * I.hv.x * u.x + uh;
                Qt[1] = c1.x + c0.y - (I.hv.x * u.y + u.x * I.hv.y);
                Qt[2] = c2.x + c0.z - (I.hv.x * u.z + u.x * I.hv.z);
                Qt[3] = 2.0 * c1.y - 2.0 * I.hv.y * u.y + uh;
                Qt[4] = c2.y + c1.z - (I.hv.y * u.z + u.y * I.hv.z);
                Qt[5] = 2.0 * c2.z - 2.0 * I.hv.z * u.z + uh;
            }
        }

        // ---- phase C: composites = suffix sums over the chain (8-lane scan by DPP row shifts, lane 7 contributes zero) ----
        SV F = f, H = h;
#define SCAN(x, D) { const double y_ = dpp_down<D>(x); if (g + D < KG) x += y_; }
#define SCAN_ROUND(D)                                                                                   \
        SCAN(F.a.x, D) SCAN(F.a.y, D) SCAN(F.a.z, D) SCAN(F.l.x, D) SCAN(F.l.y, D) SCAN(F.l.z, D)       \
        if (DERIV) {                                                                                    \
            SCAN(H.a.x, D) SCAN(H.a.y, D) SCAN(H.a.z, D) SCAN(H.l.x, D) SCAN(H.l.y, D) SCAN(H.l.z, D)   \
            SCAN(I.m, D) SCAN(I.hv.x, D) SCAN(I.hv.y, D) SCAN(I.hv.z, D)                                \
            SCAN(I.xx, D) SCAN(I.xy, D) SCAN(I.xz, D) SCAN(I.yy, D) SCAN(I.yz, D) SCAN(I.zz, D)         \
            SCAN(Qt[0], D) SCAN(Qt[1], D) SCAN(Qt[2], D) SCAN(Qt[3], D) SCAN(Qt[4], D) SCAN(Qt[5], D)   \
        }
        SCAN_ROUND(1) SCAN_ROUND(2) SCAN_ROUND(4)
#undef SCAN_ROUND
#undef SCAN

        KIN_PHASE_END(3, sv_sum(F) + sv_sum(H) + I.m + I.hv.x + I.hv.y + I.hv.z + I.xx + I.xy + I.xz + I.yy + I.yz + I.zz + Qt[0] + Qt[1] + Qt[2] + Qt[3] + Qt[4] + Qt[5] + sv_sum(sd) + sv_sum(v) + sv_sum(a))

        // ---- phase D: joint torque, and the per-joint 6-vectors of the analytic derivative ----
        if (own && G != nullptr) {
            // InverseDynamics.cpp:115-121 (damping, friction); armature inside pinocchio::rnea (>= 3)
            Gb[ro + L.off_tau + i * NV + g] =
                dot(s, F) + ptau_own;
        }
        double w0 = 0.0;
        if (DERIV) {
            w0 = pw0;
            const double w1 = pw1, w2 = pw2;
            // staged so that few of the 6-vectors are live together (this is the kernel's register peak):
            //   p = IS w2 + cv w1 + cq w0,  cv = 2 I sd + Q s + s x* H,  cq = I psi + Q sd + sd x* H + s x* F,  BtS = Q s - s x* H
            const SV psi = crm(a, s) + crm(v, sd);
            const SV IS = applyO(I, s);
            SV p = IS * w2;
            SV cv, cq_, BtS;
            {
                const SV QS = applyQ(Qt, H.l, s);
                const SV HS = crf(s, H);
                BtS = QS - HS;
                cv = applyO(I, sd) * 2.0 + QS + HS;
            }
            p = p + cv * w1;
            cq_ = applyO(I, psi) + applyQ(Qt, H.l, sd) + crf(sd, H) + crf(s, F);
            p = p + cq_ * w0;
            const SV gam = s * w2 + sd * (2.0 * w1) + psi * w0;
            const SV bet = s * w1 + sd * w0;
            if (own) {
                lds_put(sx + g * 12, IS);
                lds_put(sx + g * 12 + 6, BtS);
            }
            wave_sync();
            // ---- phase E: lane g = column g of ptau_pz(i); the 7 lanes of a group write one 56-byte row per store, the
            // groups of a wave (consecutive samples) adjacent blocks of 392 bytes: no LDS staging for these rows ----
            if (RECORD) {
                if (own) {  // lane g = joint g: column g of the three blocks
                    double* rec = JAC + (size_t)e * KREC_STRIDE + KREC_D + g;
#pragma unroll
                    for (int j = 0; j < NV; j++) {
                        double dq, dv, da;
                        if (j < g) {
                            const SV sj = lds_sv(se + KL_S + j * 6);
                            dq = dot(sj, cq_);
                            dv = dot(sj, cv);
                            da = dot(sj, IS);
                        } else {
                            const SV ISj = lds_sv(sx + j * 12), BtSj = lds_sv(sx + j * 12 + 6);
                            dq = dot(ISj, psi) + dot(BtSj, sd);
                            dv = 2.0 * dot(ISj, sd) + dot(BtSj, s);
                            da = dot(ISj, s);
                            if (j == g) {
                                da += parm;
                                dv += pdamp;
                            }
                        }
                        rec[j * NV] = dq;
                        rec[49 + j * NV] = dv;
                        rec[98 + j * NV] = da;
                    }
                }
            } else if (own) {
                double* trow = Jb + (ro + L.off_tau + i * NV) * NV + g;
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    double d;
                    if (j < g) {
                        d = dot(lds_sv(se + KL_S + j * 6), p);
                    } else {
                        d = dot(lds_sv(sx + j * 12), gam) + dot(lds_sv(sx + j * 12 + 6), bet);
                        if (j == g) d += pdiag_own;
                    }
                    trow[j * NV] = d;
                }
            }
        }
        wave_sync();  // the IS / BtS area of the whole wave is reused for the sphere rows below
        KIN_PHASE_END(4, w0 + sv_sum(s))

        // ---- spheres: KinovaCustomizedConstraints.cpp:88-157, BoxCollisionAvoidance.cpp:92-137, 290-373 ----
        // Whole rounds give one sphere to each of the 8 lanes.  A short last round (17 spheres: one left over) would leave
        // seven lanes idle through the whole box loop, so its spheres are taken one at a time with the BOXES dealt to the
        // lanes and an 8-lane arg-min that keeps the reference's first-minimum-wins rule (lowest box index on a tie).
        const int full = (ns / KG) * KG, rem = ns - full;
        const bool split_tail = rem > 0 && rem * ((O + KG - 1) / KG + 2) < O;
        const int lim = split_tail ? full : ns;
        // two spheres per lane and pass: one set of box constants (uniform scalar loads) serves both, and their two dependent
        // chains (key, square root, gradient) interleave in the issue slots that one chain leaves empty
        for (int sp = g; sp < lim; sp += 2 * KG) {
            const bool two = sp + KG < lim;
            const int sp2 = two ? sp + KG : sp;
            const int row = ro + L.off_obs + i * ns + sp;
            const int row2 = ro + L.off_obs + i * ns + sp2;
            if (O == 0) {  // :80-82, g / pg_pz keep their zero initialisation
                if (G != nullptr) { Gb[row] = 0.0; Gb[row2] = 0.0; }
                if (DERIV)
#pragma unroll
                    for (int c = 0; c < NV; c++) { srow[sp * NV + c] = 0.0; srow[sp2 * NV + c] = 0.0; }
                continue;
            }
            const SphereInfo si = sphere_info(lds, ns, se, sp), sj = sphere_info(lds, ns, se, sp2);
            double bkey = 1e300, bkey2 = 1e300;
            int bidx = 0, bidx2 = 0;
#pragma unroll 2
            for (int ob = 0; ob < O; ob++) {
                const double key = box_key(boxes + ob, si.point), key2 = box_key(boxes + ob, sj.point);
                if (key < bkey) { bkey = key; bidx = ob; }  // strict: the first minimum wins (BoxCollisionAvoidance.cpp:305-308)
                if (key2 < bkey2) { bkey2 = key2; bidx2 = ob; }
            }
            const BoxHit hit = box_hit(reinterpret_cast<const double*>(boxes + bidx), si.point);
            const BoxHit hit2 = box_hit(reinterpret_cast<const double*>(boxes + bidx2), sj.point);
            if (G != nullptr) {  // KinovaCustomizedConstraints.cpp:145
                Gb[row] = hit.dist - si.radius;
                if (two) Gb[row2] = hit2.dist - sj.radius;
            }
            if (DERIV) {
                const V3 dw = hit.n * w0, dw2 = hit2.n * w0;
                const SV Wr = SV{cross(si.point, dw), dw};  // diff . (w x (p - o)) = s_c . [p x diff ; diff]
                const SV Wr2 = SV{cross(sj.point, dw2), dw2};
#pragma unroll
                for (int c = 0; c < NV; c++) {
                    const SV sc = lds_sv(se + KL_S + c * 6);
                    srow[sp * NV + c] = (c < si.jid) ? dot(sc, Wr) : 0.0;
                    if (two) srow[sp2 * NV + c] = (c < sj.jid) ? dot(sc, Wr2) : 0.0;
                }
            }
        }
        if (split_tail) {
#pragma unroll 1
            for (int sp = full; sp < ns; sp++) {
                const int row = ro + L.off_obs + i * ns + sp;
                const SphereInfo si = sphere_info(lds, ns, se, sp);
                double bkey = 1e300;
                int bidx = 0x7fffffff;
                for (int ob = g; ob < O; ob += KG) {
                    const double key = box_key(boxes + ob, si.point);
                    if (key < bkey) { bkey = key; bidx = ob; }
                }
                // 8-lane arg-min, lowest box index on a tie: butterfly over i <-> 7 - i, i <-> i ^ 2, i <-> i ^ 1 (DPP)
#define ARGMIN_STEP(CTRL) { const double ok_ = dpp_mov64<CTRL>(bkey); const int oi_ = dpp_mov32<CTRL>(bidx); \
                            if (ok_ < bkey || (ok_ == bkey && oi_ < bidx)) { bkey = ok_; bidx = oi_; } }
                ARGMIN_STEP(0x141) ARGMIN_STEP(0x4E) ARGMIN_STEP(0xB1)
#undef ARGMIN_STEP
                const BoxHit hit = box_hit(reinterpret_cast<const double*>(boxes + bidx), si.point);
                if (g == 0 && G != nullptr) Gb[row] = hit.dist - si.radius;
                if (DERIV) {  // lane c = column c, with its own joint axis still in registers
                    const V3 dw = hit.n * w0;
                    const SV Wr = SV{cross(si.point, dw), dw};
                    if (own) srow[sp * NV + g] = (g < si.jid) ? dot(s, Wr) : 0.0;
                }
            }
        }
    }

#ifdef KIN_STOP
    if (KIN_STOP == 5) return;
#endif
    if (!DERIV) return;
    wave_sync();  // the rows of this wave's sample-evals are complete; no other wave touches them
    const double* rows = wbase + KW_X;
    if (RECORD) {  // sphere gradients per joint -> the records
        for (int k = 0; k < cntw; k++)
            for (int r = lane; r < rowlen; r += 64) JAC[(size_t)(ew0u + k) * KREC_STRIDE + KREC_G + r] = rows[k * rowlen + r];
        return;
    }
    // ---- copy-out of the obstacle rows: one contiguous run per decision vector touched by this wave, 16 bytes per lane ----
    int done = 0, ia = i0;
    long long bb = b0;
    while (done < cntw) {
        const int cnt = (cntw - done < N - ia) ? cntw - done : N - ia;
        wave_store_run(JAC + ((size_t)bb * L.m + L.off_obs + (size_t)ia * ns) * NV, rows + done * rowlen, cnt * rowlen, lane);
        done += cnt;
        bb++;
        ia = 0;
    }
}

#endif
