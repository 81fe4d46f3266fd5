// K_legged : ConstrainedInverseDynamics + TorqueLimits / ConstrainedJointLimits / RectangleSurfaceContactConstraints
// for floating-base robots with a planted stance foot (Talos, Unitree-G1; reference:
// KinematicsDynamics/src/ConstrainedInverseDynamics.cpp:45-184, DynamicsConstraints.cpp:36-103,
// Examples/Talos/src/TalosDynamicsConstraints.cpp:208-350, Constraints/src/{ConstrainedJointLimits,
// RectangleSurfaceContactConstraints,TorqueLimits}.cpp).
//
// MI355X design -- NOT the reference's algorithm order.  The reference fills the six floating-base coordinates from the
// stance-foot constraint c(q) = 0, builds J, P_dep = -J_dep^-1 J_indep and six derivative blocks from FK Hessians and
// third-order tensors, runs the floating-base RNEA derivatives (18 x 18), solves J_dep^T lambda = tau_dep and chains
// ~10 dense products per sample.  On the constraint manifold all of that is the dynamics of a FIXED-BASE tree rooted
// at the stance-foot link:
//     tau_indep               = joint torques of the re-rooted tree (principle of virtual work: G^T tau_full),
//     lambda (contact wrench) = total root wrench taken about the foot point, RPY rows through A_des = (d rpy/d omega)^-1,
//     d(tau, lambda)/dz       = analytic world-frame tree derivatives (same identities as kinova_kernel.h),
//     q_dep, P_dep            = pose / velocity map of the torso body seen from the foot.
// No 6x6 solve, no FK tensors; identical results (numpy model: scripts/proto_rerooted.py agrees with the oracle's
// literal restatement to 1e-15).
//
// Two kernels (since round 3 the first also does the second's work whenever the sizes allow: FUSE instances, phase G):
//   legged_core_kernel  : LG lanes per sample-eval (lane = body of the re-rooted tree, DFS pre-order so that a subtree
//                         is a lane range and composites are suffix-scan differences on wave shuffles).  Writes the
//                         g rows and a compact per-sample record: D_q, D_v, D_a (ni x ni), d lambda (6 x ni x 3), P_dep.
//   legged_expand_kernel: streams the Jacobian rows out in Optimizer order, 16 bytes per lane, from those records:
//                         row(., t*K + k) = D_q(., t) B_k + D_v(., t) dB_k + D_a(., t) ddB_k   (BezierCurves.cpp:165-169)
//                         -- this kernel is the HBM-bound one (8 * (n + 1) bytes per row).
#ifndef RAPTOR_LEGGED_KERNEL_H
#define RAPTOR_LEGGED_KERNEL_H

#include "kernels.h"
#include "kinova_kernel.h"

#define LEG_BLOCK 64      // core kernel: one wavefront per workgroup (no workgroup barrier anywhere)
#ifndef LEG_EXP_BLOCK
#define LEG_EXP_BLOCK 256  // streaming kernel: threads per workgroup (one workgroup per sample-eval)
#endif
#define LEG_VEC 36        // per-joint 6-vectors kept in LDS: s, sd, psi, IS, cv, cq
#ifndef LEG_STAGGER
#define LEG_STAGGER 2     // fused core kernel: start offset between the hardware slots of a CU, in s_sleep(127) units of ~3.5 us (measured 0 .. 32)
#endif
// Measurement builds of the fused core kernel's phase G: -DLEG_G_PROBE=1 drops its stores (one lane in 2^20 keeps the
// values alive), =2 ends the phase after the records are back in LDS.
#if defined(LEG_G_PROBE) && LEG_G_PROBE == 1
#define LEG_G_ON(c) ((c) && (tid + (int)blockIdx.x * 64) % 1048576 == 12345)
#else
#define LEG_G_ON(c) (c)
#endif
#define LEG_LOC 18        // per-joint local transform for the kinematic walk: X (9), t (3), axis (3), qd, qdd, rev

typedef double v2d_t __attribute__((ext_vector_type(2)));
// 16-byte chunk c of a per-joint vector whose chunks are swapped in pairs when sw2 != 0 (chunk c lives at c ^ 1): `base` is the
// vector's first double (an even chunk), sw2 = 0 or 2 doubles (-2 when `base` itself starts at an odd chunk)
__device__ __forceinline__ v2d_t* vec_chunk_ptr(double* base, int sw2, int c) {
    return reinterpret_cast<v2d_t*>(base + 2 * c + ((c & 1) ? -sw2 : sw2));
}

template <int LG>
struct LegLds {
    // phases 0/A: per-joint local transform (LEG_LOC doubles) ; phases E/F: per-joint 6-vectors (LEG_VEC doubles) -- same area
    static constexpr int V = 0;
    static constexpr int TS = LG * LEG_VEC;     // per-column scratch: T_q, T_v, T_a per joint
    static constexpr int TSS = 4;               // doubles per joint in that scratch (q, v, a, pad): a 16-byte and an 8-byte access
    static constexpr int STRIDE = TS + LG * TSS;  // LG = 32: 1280 doubles -> 2 sample-evals = 20 KB per wavefront, 8 waves per CU exactly
};

// Utils::wrapToPi (Utils/include/Utils.h:59-68)
RD double wrap_to_pi(double a) {
    // the reference's loops, bounded so that every wave finishes: beyond 64 turns one remainder step comes first, and a
    // non-finite angle (the reference would never return) is handed back as it is
    if (!(fabs(a) <= 128.0 * M_PI)) {
        if (!isfinite(a)) return a;
        a = remainder(a, 2.0 * M_PI);
    }
    while (a > M_PI) a -= 2.0 * M_PI;
    while (a < -M_PI) a += 2.0 * M_PI;
    return a;
}
// HigherOrderDerivatives.h:10-38
__host__ __device__ inline double safe_asin(double x) { return x > 1.0 ? M_PI_2 : (x < -1.0 ? -M_PI_2 : asin(x)); }
__host__ __device__ inline double safe_dasin(double x) { return x >= 1.0 ? 1e10 : (x <= -1.0 ? -1e10 : 1.0 / sqrt(1.0 - x * x)); }

// d(roll, pitch, yaw)/dt of Transform::getRPY (Transform.cpp:379-390) for a frame R rotating with world-frame angular
// velocity w: the rows of ForwardKinematicsSolver::getRPYJacobian (ForwardKinematics.cpp:358-384) with dR = [w]x R.
__host__ __device__ inline void rpy_rates(const double* R, const double* w, double* out) {
    const double dR02 = -w[2] * R[5] + w[1] * R[8];
    const double dR00 = -w[2] * R[3] + w[1] * R[6];
    const double dR01 = -w[2] * R[4] + w[1] * R[7];
    const double dR12 = w[2] * R[2] - w[0] * R[8];
    const double dR22 = -w[1] * R[2] + w[0] * R[5];
    const double rollDenom = R[5] * R[5] + R[8] * R[8];
    const double yawDenom = R[0] * R[0] + R[1] * R[1];
    out[0] = (R[5] * dR22 - R[8] * dR12) / rollDenom;
    out[1] = safe_dasin(R[2]) * dR02;
    out[2] = (R[1] * dR00 - R[0] * dR01) / yawDenom;
}
RD V3 rpy_rates(const M3& R, V3 w) {
    const double wv[3] = {w.x, w.y, w.z};
    double o[3];
    rpy_rates(R.m, wv, o);
    return v3(o[0], o[1], o[2]);
}

RD M3 mulBT(const M3& A, const double* B) {  // A * B^T, B row-major 3x3
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[3 * i] * B[3 * j] + A.m[3 * i + 1] * B[3 * j + 1] + A.m[3 * i + 2] * B[3 * j + 2];
    return C;
}

// RectangleSurfaceContactConstraints.cpp:52-93: row r of the 7 contact rows as a linear map of d(lambda.tail(6)) = p
RD double contact_row(int r, const double* lam, const double* p, double mu, double gamma, double Lx, double Ly) {
    switch (r) {
        case 0: return p[2];
        case 1: {
            const double ff = sqrt(lam[0] * lam[0] + lam[1] * lam[1]);
            return (ff <= 1e-8) ? -mu * p[2] : (lam[0] * p[0] + lam[1] * p[1]) / ff - mu * p[2];
        }
        case 2: return sign_eps(lam[5]) * p[5] - gamma * p[2];
        case 3: return p[3] - Lx * p[2];
        case 4: return -Lx * p[2] - p[3];
        case 5: return p[4] - Ly * p[2];
        default: return -Ly * p[2] - p[4];
    }
}

// Streams pg_pz out in Optimizer order.  One workgroup per sample-eval: the sample's record and the per-column Bernstein
// weights are staged in LDS once; the three row blocks of this sample are contiguous pieces of the output
// (rows i*R .. i*R + R - 1 of a block), written with one 16-byte store per lane and no integer division in the loop.
//   row(., t*K + k) = D_q(., t) B_k + D_v(., t) dB_k + D_a(., t) ddB_k     (BezierCurves.cpp:165-169 chained into
//   TorqueLimits.cpp:36-40, ConstrainedJointLimits.cpp:59-68, RectangleSurfaceContactConstraints.cpp:52-93)
// 16-byte store of two adjacent columns
RD void g_st(double* p, const v2d_t v) {  // phase G of the fused core kernel
#ifdef LEG_G_NT
    __builtin_nontemporal_store(v, reinterpret_cast<v2d_t*>(p));
#else
    *reinterpret_cast<v2d_t*>(p) = v;
#endif
}
RD void store2_stream(double* p, double a, double b) {
    v2d_t v = {a, b};
#ifdef RAPTOR_NT_STORES
    __builtin_nontemporal_store(v, reinterpret_cast<v2d_t*>(p));
#else
    *reinterpret_cast<v2d_t*>(p) = v;  // plain stores measured faster than nontemporal ones at the 4 GB bench size
#endif
}

template <int PIECE, int BLK = LEG_EXP_BLOCK>
RD void legged_expand_piece(const int tid, const int n, const int ndyn, const int ni, const int R, const int cnt,
                            double* __restrict__ dst, const double* __restrict__ rec0, const int recstride,
                            const double* __restrict__ wB0, const int* __restrict__ tcol, const int* __restrict__ rowinfo,
                            const DevLegged* __restrict__ Q) {
    // `cnt` (1 or 2) consecutive samples of one decision vector: their rows of a block are adjacent, so the piece is one
    // contiguous run of cnt * R rows
    const double mu = Q->mu, gam = Q->gamma, Lx = Q->Lx, Ly = Q->Ly;
    const int dr = (2 * BLK) / n, dc = (2 * BLK) - dr * n;
    const int total = cnt * R * n;
    int el = 2 * tid;
    int r = el / n, c = el - r * n;
    for (; el < total; el += 2 * BLK) {
        const int sidx = (r >= R) ? 1 : 0;
        const int rr_ = r - sidx * R;
        const double* rec = rec0 + sidx * recstride;
        const double* wB = wB0 + sidx * 3 * ndyn;
        double o[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int cc = c + h;
            double val = 0.0;
            if (cc < ndyn) {  // the v_reset / lambda_reset tail never enters (TalosSingleStepOptimizer.cpp:50-51)
                const int t = tcol[cc];
                const double B0 = wB[cc];
                if (PIECE == 0) {  // TorqueLimits.cpp:36-40 on ConstrainedInverseDynamics::ptau_pz
                    const double* d = rec + Q->c_D + (t * 3) * ni + rr_;
                    val = d[0] * B0 + d[ni] * wB[ndyn + cc] + d[2 * ni] * wB[2 * ndyn + cc];
                } else if (PIECE == 1) {  // ConstrainedJointLimits.cpp:59-68
                    const double* Pb = rec + Q->c_P;
                    const double* Pl = rec + Q->c_Pl;
                    const int info = rowinfo[rr_];  // >= 0: independent index; < 0: -1 - dependent index
                    if (info >= 0) {
                        val = (info == t) ? B0 : 0.0;
                    } else {
                        const int dd = -1 - info;
                        double g;
                        if (dd < 6) g = Pb[dd * ni + t];
                        else g = (t == rowinfo[RAPTOR_MAX_NV + 2 * dd]) ? Pl[(dd - 6) * 2] : ((t == rowinfo[RAPTOR_MAX_NV + 2 * dd + 1]) ? Pl[(dd - 6) * 2 + 1] : 0.0);
                        val = g * B0;
                    }
                } else {  // RectangleSurfaceContactConstraints.cpp:52-93
                    double p[6];
                    const double B1 = wB[ndyn + cc], B2 = wB[2 * ndyn + cc];
                    const double* l = rec + Q->c_L + t;
#pragma unroll
                    for (int q = 0; q < 6; q++) p[q] = l[q * ni] * B0 + l[(6 + q) * ni] * B1 + l[(12 + q) * ni] * B2;
                    val = contact_row(rr_, rec + Q->c_lam, p, mu, gam, Lx, Ly);
                }
            }
            o[h] = val;
        }
        store2_stream(dst + el, o[0], o[1]);
        r += dr;
        c += dc;
        if (c >= n) { c -= n; r++; }
    }
}

// customized rows r0 .. r0 + nrows - 1: ONE contiguous run per workgroup (a workgroup that wrote the rows of its own samples
// would scatter 1 KB rows over the block -- the block is ordered by quantity, then sample -- which costs 15 % of the write rate of
// the whole kernel, scripts/bw_probe.py).  A row is (ca X[qa] + cb X[qb]) differentiated: coef[r][t] per actuator t (formed in the
// prologue from the records of the row's sample) times the order-0 Bernstein weight of its sample.
template <int BLK = LEG_EXP_BLOCK>
RD void legged_expand_cust(const int tid, const int n, const int ndyn, const int ni, const int K, const int nrows,
                           double* __restrict__ dst, const double* __restrict__ coef, const int* __restrict__ rsample,
                           const double* __restrict__ B0, const int* __restrict__ tcol) {
    const int dr = (2 * BLK) / n, dc = (2 * BLK) - dr * n;
    const int total = nrows * n;
    int el = 2 * tid;
    int r = el / n, c = el - r * n;
    for (; el < total; el += 2 * BLK) {
        const double* cf = coef + r * ni;
        const double* w = B0 + rsample[r] * K;
        double o[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int cc = c + h;
            double val = 0.0;
            if (cc < ndyn) {
                const int t = tcol[cc];
                val = cf[t] * w[cc - t * K];
            }
            o[h] = val;
        }
        store2_stream(dst + el, o[0], o[1]);
        r += dr;
        c += dc;
        if (c >= n) { c -= n; r++; }
    }
}

// Phase accounting: -DLEG_STOP=k ends the kernel after phase k with a checksum store.  Measurement builds only.
#ifdef LEG_STOP
#define LEG_PHASE_END(k, expr) if (LEG_STOP == (k)) { rec[g] = (expr); return; }
#else
#define LEG_PHASE_END(k, expr)
#endif
template <int LG, bool DERIV, bool FUSE = false, bool KEVEN = true>
__global__ __launch_bounds__(LEG_BLOCK, 2) void legged_core_kernel(
    const DevModel* __restrict__ M, const DevTraj* __restrict__ T, const DevLegged* __restrict__ Q,
    const double* __restrict__ basis, const DevLayout L, const double* __restrict__ Z, const long long E,
    const double* __restrict__ loops, const DevCustRow* __restrict__ cust, const int* __restrict__ cust_list,
    const int* __restrict__ cust_start, double* __restrict__ G, double* __restrict__ core, double* __restrict__ lamrec,
    double* __restrict__ JAC = nullptr, const int stagger_wgs = 0, const int stagger_units = (4 * LEG_STAGGER) << 8 | LEG_STAGGER) {
    // lamrec (may be null; the component view raptor_constrained_id_lambda_host only): per sample-eval the tree torque of every
    // closed-loop joint and its derivative per unit (q, v, a) of every actuator,  tau[nrec] | T[it][x][nrec]
    extern __shared__ __align__(16) double lds[];
    using LL = LegLds<LG>;
    constexpr int EPB = LEG_BLOCK / LG;
    const int tid = threadIdx.x;
    if (FUSE && (int)blockIdx.x < stagger_wgs) {
        // Fused form: every wavefront alternates a compute phase (bound by fp64 issue) with a store phase (bound by the write path).
        // Launched together, the resident wavefronts of a CU run in lock step -- all compute, then all store -- and nothing overlaps
        // (measured: no gain at all over two kernels).  The wavefronts of the FIRST round therefore start spread over a store phase by
        // their hardware slot (SIMD, wave slot parity); later rounds inherit the offsets because every wavefront does the same work.
        const unsigned simd = __builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4);  // HW_REG_HW_ID: wave_id [3:0], simd_id [5:4]
        const unsigned wslot = __builtin_amdgcn_s_getreg((1 - 1) << 11 | 0 << 6 | 4);
        // stagger_units = (units per wave slot) << 8 | (units per SIMD)
        const int delay = (int)wslot * (stagger_units >> 8) + (int)simd * (stagger_units & 255);
        for (int k = 0; k < delay; k++) __builtin_amdgcn_s_sleep(127);  // 127 * 64 clocks ~ 3.5 us each
    }
    const int g = tid & (LG - 1);
    const int eg = tid / LG;
    // a lane group past the end repeats the last sample-eval (same values to the same addresses): every lane of the wavefront
    // stays active, which the packed table of phase F2 (an entry per lane, read with v_readlane) relies on
#ifdef LEG_G_TRACE  // measurement build (scripts/fused_trace.py): where a wavefront's life goes -- 100 MHz timestamps left in the torque
                    // values of the wavefront's first sample
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long trG = 0, trS = 0, trA = 0, trB = 0, trC = 0, trD = 0;
#define LEG_TR(x) x = __builtin_amdgcn_s_memrealtime();
#else
#define LEG_TR(x)
#endif
    const long long e0 = (long long)blockIdx.x * EPB;  // (uniform; < E by the launch grid)
    const long long e = min(e0 + eg, E - 1);
    double* sl = lds + eg * LL::STRIDE;
    // The sample-evals of a wavefront sit STRIDE doubles apart, a multiple of the 64 banks: a broadcast read of joint k's vector by every
    // lane group hit the same banks once per group (SQ_LDS_BANK_CONFLICT ~ 1.2 cycles per LDS instruction in round 2).  Odd groups now
    // keep the 16-byte chunks of the per-joint vector area swapped in pairs (chunk c at c ^ 1): the groups' reads fall on disjoint banks.
    const int sw2 = (eg & 1) * 2;  // doubles: even chunks move up by it, odd chunks down
    const int N = T->N;
    // one 64-bit division per wavefront (its first sample-eval), then a wrap per lane group: a per-lane e / N, e % N is ~150 VALU
    const long long b0 = e0 / N;
    int i = (int)(e0 - b0 * N) + (int)(e - e0);
    int wrap = 0;
    while (i >= N) { i -= N; wrap++; }
    const long long b = b0 + wrap;
    const double* z = Z + b * L.n;
    const size_t rowbase = (size_t)b * L.m;
    double* rec = core + (size_t)e * Q->c_stride;
    const int nb = Q->nb, ni = Q->ni, nv = Q->nv;
    const bool has_body = g < nb;
    const bool has_joint = has_body && g > 0;
    const int gb = has_body ? g : 0;
    const int jnt = has_joint ? Q->body_joint[gb] : 0;
    const int tj = has_joint ? Q->body_traj[gb] : -1;
    const int rc = has_joint ? Q->body_rec[gb] : -1;
    const int gend = has_body ? Q->body_end[gb] : LG;
    // tables this lane walks later, fetched now: inside the loops a dependent global load per step is what the two wavefronts
    // per SIMD cannot hide
    const double arm_j = has_joint ? M->armature[jnt] : 0.0, damp_j = has_joint ? M->damping[jnt] : 0.0;

    // ---- phase 0: this body's joint value, rates, chain-rule weights towards the independent columns, sincos ----
    const int flat_pre = DERIV ? Q->flat[tid] : 0;  // phase F2's work list: loaded before any store (in-order vector-memory counter)
    double q = 0.0, qd = 0.0, qdd = 0.0, sq = 0.0, cq = 1.0;
    double w0[3] = {1.0, 0.0, 0.0}, w1[3] = {0.0, 0.0, 0.0};
    {
        // Every lane evaluates (up to) two trajectory columns -- an actuator its own, a closed-loop joint its loop's drivers -- with
        // all coefficient, basis and loop-record loads in flight together: the three traj_joint calls this replaces sat under divergent
        // branches, each a loop of K load -> wait -> FMA steps (~18 dependent round trips at the start of every wavefront).
        const int ca = (tj >= 0) ? tj : ((rc >= 0) ? Q->body_col[gb][0] : -1);
        const int cb = (tj < 0 && rc >= 0) ? Q->body_col[gb][1] : -1;
        double xa = 0.0, xda = 0.0, xdda = 0.0, xb = 0.0, xdb = 0.0, xddb = 0.0;
        const int K = T->K;
        double lrv[10];
        {
            const double* lp = loops != nullptr ? loops + ((size_t)e * Q->nrec + (rc >= 0 ? rc : 0)) * 10 : nullptr;
#pragma unroll
            for (int k = 0; k < 10; k++) lrv[k] = (lp != nullptr) ? lp[k] : 0.0;
        }
        if (T->type == RAPTOR_TRAJ_BEZIER && K <= 8) {  // (uniform)
            const int ja = ca >= 0 ? ca : 0, jb = cb >= 0 ? cb : 0;
            const double* B0 = basis + (size_t)(i * 3) * K;
            double za[8], zb[8], b0[8], b1[8], b2[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int kk = k < K ? k : 0;
                za[k] = z[ja * K + kk];  // column-major reshape (Utils.h:142-146)
                zb[k] = z[jb * K + kk];
                b0[k] = B0[kk];
                b1[k] = B0[K + kk];
                b2[k] = B0[2 * K + kk];
            }
            double fa[3] = {0.0, 0.0, 0.0}, fb[3] = {0.0, 0.0, 0.0};
            if (T->nfixed > 0) {  // rows pinned by the constrained initial state (BezierCurves.cpp:84-96): table after the basis
                const int Nact = T->Nact;
                const double* fx = basis + (size_t)T->N * 3 * K + (size_t)(i * 3) * Nact;
#pragma unroll
                for (int o = 0; o < 3; o++) { fa[o] = fx[o * Nact + ja]; fb[o] = fx[o * Nact + jb]; }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k < K) {
                    xa += za[k] * b0[k]; xda += za[k] * b1[k]; xdda += za[k] * b2[k];
                    xb += zb[k] * b0[k]; xdb += zb[k] * b1[k]; xddb += zb[k] * b2[k];
                }
            }
            xa += fa[0]; xda += fa[1]; xdda += fa[2];
            xb += fb[0]; xdb += fb[1]; xddb += fb[2];
            if (ca < 0) { xa = 0.0; xda = 0.0; xdda = 0.0; }
            if (cb < 0) { xb = 0.0; xdb = 0.0; xddb = 0.0; }
        } else {
            if (ca >= 0) traj_joint(T, basis, z, i, ca, xa, xda, xdda);
            if (cb >= 0) traj_joint(T, basis, z, i, cb, xb, xdb, xddb);
        }
        if (tj >= 0) {
            q = xa; qd = xda; qdd = xdda;
        } else if (rc >= 0) {
            // closed-loop joint: theta = y(x), theta' = Y1 x', theta'' = Y1 x'' + x'^T Y2 x'   (digit_kernel.h records)
            const double* lr = lrv;
            q = lr[0];
            qd = lr[1] * xda + lr[2] * xdb;
            qdd = lr[1] * xdda + lr[2] * xddb + lr[3] * xda * xda + 2.0 * lr[4] * xda * xdb + lr[5] * xdb * xdb;
            w0[0] = lr[1];
            w0[1] = 2.0 * (lr[3] * xda + lr[4] * xdb);
            w0[2] = lr[3] * xdda + lr[4] * xddb + lr[6] * xda * xda + 2.0 * lr[7] * xda * xdb + lr[8] * xdb * xdb;
            w1[0] = lr[2];
            w1[1] = 2.0 * (lr[4] * xda + lr[5] * xdb);
            w1[2] = lr[4] * xdda + lr[5] * xddb + lr[7] * xda * xda + 2.0 * lr[8] * xda * xdb + lr[9] * xdb * xdb;
        }
    }
    // local transform across this joint in the direction from the foot to this body:
    //   forward joint  R' = R (plR Rq),   o' = o + R plp,   axis = R (plR e) = R' (X^T plR e),   joint origin o'
    //   reversed joint R' = R (plR Rq)^T, o' = o - R' plp,  axis = -R' (plR e),                  joint origin o = o' + R' plp
    // i.e. with (X, t) the step R' = R X, o' = o + R t, the world axis and joint origin follow from the body's OWN world pose and
    // two constant vectors c_ax, c_o -- no parent pose is needed.
    M3 Xl;
#pragma unroll
    for (int k = 0; k < 9; k++) Xl.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
    V3 tl = v3(0, 0, 0), c_ax = v3(0, 0, 0), c_o = v3(0, 0, 0);
    if (has_joint) {
        sincos_joint(q, &sq, &cq);
        M3 P;
#pragma unroll
        for (int k = 0; k < 9; k++) P.m[k] = M->plR[jnt][k];
        const V3 pl = v3(M->plp[jnt][0], M->plp[jnt][1], M->plp[jnt][2]);
        M3 X;
        V3 dummy = v3(0, 0, 0);
        SV sj;
        joint_world(M->jtype[jnt], q, cq, sq, P, X, dummy, sj);  // X = plR Rq(q), sj.a = signed axis in the parent link frame
        const bool rev = Q->body_rev[gb] != 0;
        M3 Xt;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) Xt.m[3 * r + c] = X.m[3 * c + r];
        if (rev) {
            Xl = Xt;
            tl = -mul(Xt, pl);
            c_ax = -sj.a;
            c_o = pl;
        } else {
            Xl = X;
            tl = pl;
            c_ax = mul(Xt, sj.a);
        }
    } else if (g == 0) {  // the root body: the planted foot's world pose
#pragma unroll
        for (int k = 0; k < 9; k++) Xl.m[k] = Q->R0[k];
        tl = v3(Q->o0[0], Q->o0[1], Q->o0[2]);
    }
    LEG_PHASE_END(1, q + qd + qdd + sq + cq + w0[0] + w0[1] + w0[2] + w1[0] + w1[1] + w1[2] + Xl.m[0] + tl.x)

    // ---- phase A: world-frame kinematics by pointer doubling over the tree (round 3; rounds 1-2 walked every lane's own root path,
    // up to 15 dependent steps of ~75 instructions for Digit).  After round r a lane holds the transform / velocity sum / acceleration
    // sum over the last 2^(r+1) bodies of its root path: T_g <- T_anc(g, 2^r) o T_g through an exchange area in LDS, ceil(log2(depth + 1))
    // rounds (4 for Digit).  v_g = sum s_k qd_k and a_g = a0 + sum (s_k qdd_k + (v_k x s_k) qd_k) are path sums of per-body terms
    // (v_parent x s_k qd_k = v_k x s_k qd_k since s x s = 0). ----
    const unsigned jumps = has_body ? Q->jump_pk[gb] : 0u;
    const int nrounds = Q->nrounds;
    double* const xch = sl + LL::V + g * LEG_LOC;  // this lane's exchange slot (18 doubles, 16-byte aligned)
    M3 Rw = Xl;
    V3 ow = tl;
#pragma unroll 1
    for (int r = 0; r < nrounds; r++) {
        v2d_t* xw = reinterpret_cast<v2d_t*>(xch);
        xw[0] = v2d_t{Rw.m[0], Rw.m[1]}; xw[1] = v2d_t{Rw.m[2], Rw.m[3]}; xw[2] = v2d_t{Rw.m[4], Rw.m[5]};
        xw[3] = v2d_t{Rw.m[6], Rw.m[7]}; xw[4] = v2d_t{Rw.m[8], ow.x}; xw[5] = v2d_t{ow.y, ow.z};
        wave_sync();
        const unsigned j = jumps >> (6 * r);
        const bool take = (j & 32u) != 0;
        const v2d_t* xr = reinterpret_cast<const v2d_t*>(sl + LL::V + (take ? (int)(j & 31u) : g) * LEG_LOC);
        v2d_t u[6];
#pragma unroll
        for (int c = 0; c < 6; c++) u[c] = xr[c];
        wave_sync();
        if (take) {
            const double A[9] = {u[0].x, u[0].y, u[1].x, u[1].y, u[2].x, u[2].y, u[3].x, u[3].y, u[4].x};
            const V3 ta = v3(u[4].y, u[5].x, u[5].y);
            M3 Am;
#pragma unroll
            for (int k = 0; k < 9; k++) Am.m[k] = A[k];
            ow = ta + mul(Am, ow);
            Rw = mul(Am, Rw.m);
        }
    }
    const V3 axw = mul(Rw, c_ax);
    const V3 ojw = ow + mul(Rw, c_o);
    const SV s = has_joint ? SV{axw, cross(ojw, axw)} : sv_zero();
    SV vA = s * qd;
#pragma unroll 1
    for (int r = 0; r < nrounds; r++) {
        lds_put(xch, vA);
        wave_sync();
        const unsigned j = jumps >> (6 * r);
        const bool take = (j & 32u) != 0;
        const SV up = lds_sv(sl + LL::V + (take ? (int)(j & 31u) : g) * LEG_LOC);
        wave_sync();
        if (take) vA = vA + up;
    }
    const SV sdA = crm(vA, s);
    const SV a0 = SV{v3(0, 0, 0), v3(M->neg_gravity[0], M->neg_gravity[1], M->neg_gravity[2])};
    SV aA = (g == 0) ? a0 : (s * qdd + sdA * qd);  // the root body carries the base acceleration -a_grav into every path sum
#pragma unroll 1
    for (int r = 0; r < nrounds; r++) {
        lds_put(xch, aA);
        wave_sync();
        const unsigned j = jumps >> (6 * r);
        const bool take = (j & 32u) != 0;
        const SV up = lds_sv(sl + LL::V + (take ? (int)(j & 31u) : g) * LEG_LOC);
        wave_sync();
        if (take) aA = aA + up;
    }
    if (!has_body) aA = a0;

    LEG_PHASE_END(2, sv_sum(vA) + sv_sum(aA) + sv_sum(s) + sv_sum(sdA) + Rw.m[0] + Rw.m[4] + Rw.m[8] + Rw.m[1] + Rw.m[5] + Rw.m[6] + ow.x + ow.y + ow.z)

    // ---- phase B: this body's inertia, momentum, wrench, Coriolis block (world frame, about the origin) ----
    InertiaO I;
    SV h, f;
    double Qt[6];
    {
        const int link = Q->body_link[gb];
        Inertia Ic;
        const V3 c = ow + mul(Rw, v3(M->lever[link][0], M->lever[link][1], M->lever[link][2]));
        rotate_inertia(Rw, M->Ic[link], Ic);
        const double m = has_body ? M->mass[link] : 0.0;
        const double keep = has_body ? 1.0 : 0.0;
        I.m = m;
        I.hv = c * m;
        const double cc = dot(c, c);
        I.xx = keep * Ic.xx + m * (cc - c.x * c.x);
        I.yy = keep * Ic.yy + m * (cc - c.y * c.y);
        I.zz = keep * Ic.zz + m * (cc - c.z * c.z);
        I.xy = keep * Ic.xy - m * c.x * c.y;
        I.xz = keep * Ic.xz - m * c.x * c.z;
        I.yz = keep * Ic.yz - m * c.y * c.z;
        h = applyO(I, vA);
        f = applyO(I, aA) + crf(vA, h);
        if (DERIV) {
            const V3 w = vA.a, u = vA.l;
            const V3 c0 = cross(w, v3(I.xx, I.xy, I.xz));
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

    // (Rounds 2-3: v, a and sd used to wait in LDS across the scans of phase C.  With phase E ahead of the value phase the kernel
    // needs 222 registers and spills nothing, and keeping them in registers runs the record kernels 1.262 -> 1.252 ms.)
    wave_sync();  // every lane's walk has read the transforms before phase E rewrites the area

    // ---- phase C: subtree composites = suffix scan over the DFS order minus the suffix at the subtree's end ----
    SV F = f, H = h;
    {
        const int src = gend & (LG - 1);
        const bool cut = gend < LG;
#define SCAN(x) { const double y_ = __shfl_down(x, d, LG); if (take) x += y_; }
#define CUT(x) { const double y_ = __shfl(x, src, LG); if (cut) x -= y_; }
#pragma unroll
        for (int d = 1; d < LG; d <<= 1) {
            const bool take = (g + d) < LG;
            SCAN(F.a.x) SCAN(F.a.y) SCAN(F.a.z) SCAN(F.l.x) SCAN(F.l.y) SCAN(F.l.z)
            if (DERIV) {
                SCAN(H.a.x) SCAN(H.a.y) SCAN(H.a.z) SCAN(H.l.x) SCAN(H.l.y) SCAN(H.l.z)
                SCAN(I.m) SCAN(I.hv.x) SCAN(I.hv.y) SCAN(I.hv.z)
                SCAN(I.xx) SCAN(I.xy) SCAN(I.xz) SCAN(I.yy) SCAN(I.yz) SCAN(I.zz)
                SCAN(Qt[0]) SCAN(Qt[1]) SCAN(Qt[2]) SCAN(Qt[3]) SCAN(Qt[4]) SCAN(Qt[5])
            }
        }
        CUT(F.a.x) CUT(F.a.y) CUT(F.a.z) CUT(F.l.x) CUT(F.l.y) CUT(F.l.z)
        if (DERIV) {
            CUT(H.a.x) CUT(H.a.y) CUT(H.a.z) CUT(H.l.x) CUT(H.l.y) CUT(H.l.z)
            CUT(I.m) CUT(I.hv.x) CUT(I.hv.y) CUT(I.hv.z)
            CUT(I.xx) CUT(I.xy) CUT(I.xz) CUT(I.yy) CUT(I.yz) CUT(I.zz)
            CUT(Qt[0]) CUT(Qt[1]) CUT(Qt[2]) CUT(Qt[3]) CUT(Qt[4]) CUT(Qt[5])
        }
#undef SCAN
#undef CUT
    }

    LEG_PHASE_END(3, sv_sum(F) + sv_sum(H) + I.m + I.hv.x + I.hv.y + I.hv.z + I.xx + I.xy + I.xz + I.yy + I.yz + I.zz + Qt[0] + Qt[1] + Qt[2] + Qt[3] + Qt[4] + Qt[5] + sv_sum(sdA) + sv_sum(vA) + sv_sum(aA))

    // ---- phase E (runs BEFORE the value phase since round 3): per-joint 6-vectors of the analytic derivative.  After it the composite
    // inertia, momentum and Coriolis block (22 doubles) are dead and IS / BtS (12) take their place, so the value phase below -- where
    // the kernel's register peak sits, with every table and record load of phases D-F2 in flight before the first store -- runs with
    // 20 registers more.  This phase has no global store, so those loads are still ahead of every store.
    const bool is_torso = has_body && g == Q->torso_body;
    SV BtS = sv_zero();
    if (DERIV) {
        const SV v = vA, a = aA, sd = sdA;
        const SV psi = crm(a, s) + crm(v, sd);
        const SV IS = applyO(I, s);
        {
            const SV IdS = applyO(I, sd);
            const SV Ipsi = applyO(I, psi);
            const SV QS = applyQ(Qt, H.l, s);
            const SV QdS = applyQ(Qt, H.l, sd);
            const SV HS = crf(s, H);
            const SV HdS = crf(sd, H);
            const SV cF = crf(s, F);
            BtS = QS - HS;
            const SV cv = IdS * 2.0 + QS + HS;
            const SV cq_ = Ipsi + QdS + HdS + cF;
            if (has_joint) {
                double* vv = sl + LL::V + g * LEG_VEC;
                const SV six[6] = {s, sd, psi, IS, cv, cq_};
#pragma unroll
                for (int q6 = 0; q6 < 6; q6++) {
                    const v2d_t c0 = v2d_t{six[q6].a.x, six[q6].a.y}, c1 = v2d_t{six[q6].a.z, six[q6].l.x}, c2 = v2d_t{six[q6].l.y, six[q6].l.z};
                    *vec_chunk_ptr(vv, sw2, 3 * q6) = c0;
                    *vec_chunk_ptr(vv, sw2, 3 * q6 + 1) = c1;
                    *vec_chunk_ptr(vv, sw2, 3 * q6 + 2) = c2;
                }
            }
        }
        if (is_torso) {  // the torso's twist and acceleration, which the value phase turns into the floating-base coordinates' rates
            lds_put(sl + LL::TS + 24, v);
            lds_put(sl + LL::TS + 30, a);
        }
        wave_sync();
    }
    LEG_PHASE_END(4, sv_sum(F) + sv_sum(BtS))

    // ---- phase D: values.  tau rows, q rows, contact rows; the same numbers go to the compact record ----
    const double mu = Q->mu, gam = Q->gamma, Lx = Q->Lx, Ly = Q->Ly;
    const V3 pf = v3(Q->p_foot[0], Q->p_foot[1], Q->p_foot[2]);
    // tree-joint torque; ConstrainedInverseDynamics.cpp:131-132 (damping only; armature lives inside pinocchio::rnea >= 3)
    const double tt = has_joint ? dot(s, F) + arm_j * qdd + damp_j * qd : 0.0;
    // Every table look-up and record load of phases D and E is issued here, before the first store of the kernel: the vector-memory
    // counter retires loads and stores in issue order, so a load behind a store waits for that store's round trip.
    // an actuator's lane collects, from the lanes of the closed-loop joints it drives, their G = Y1 entry and torque
    // (unconditional loads from valid addresses, selected afterwards: a load under `d < ndep` becomes a branch per entry and the
    // twelve look-ups a chain of twelve waits)
    const int tjc = tj >= 0 ? tj : 0;
    const int myslot_ld = Q->ind_slot[tjc], ndep_ld = Q->ind_ndep[tjc];
    int depb[6];  // bodies of the closed-loop joints this actuator drives
#pragma unroll
    for (int d = 0; d < 6; d++) depb[d] = Q->ind_dep_body[tjc][d];
    const int myslot = (tj >= 0) ? myslot_ld : 0;
    const int ndep = (tj >= 0) ? ndep_ld : 0;
    double gdep[6], tdep[6];
#pragma unroll
    for (int d = 0; d < 6; d++) depb[d] = (d < ndep) ? (depb[d] & (RAPTOR_LEG_MAX_BODIES - 1)) : 0;
    int recb[6];  // their records in the loop buffer (entry 0 beyond ndep: a valid address, weighted by zero below)
#pragma unroll
    for (int d = 0; d < 6; d++) recb[d] = Q->body_rec[depb[d]];
#pragma unroll
    for (int d = 0; d < 6; d++) recb[d] = (d < ndep) ? recb[d] : 0;
#pragma unroll
    for (int d = 0; d < 6; d++) {
        const int db = depb[d];
        const double ga = __shfl(w0[0], db, LG), gb_ = __shfl(w1[0], db, LG);
        const double td = __shfl(tt, db, LG);
        gdep[d] = (d < ndep) ? (myslot ? gb_ : ga) : 0.0;
        tdep[d] = (d < ndep) ? td : 0.0;
    }
    // (dG^T/dq) tau_tree: an actuator's torque also changes because the loop Jacobian Y1 moves with the drivers
    double Kq[2] = {0.0, 0.0};
    if (DERIV && ndep > 0) {  // all twelve loads in flight together (a branch per joint would wait for each pair in turn)
        double k0[6], k1[6];
#pragma unroll
        for (int d = 0; d < 6; d++) {  // selected, not weighted by zero: a non-finite record 0 must not leak into a healthy actuator
            const double* ld = loops + ((size_t)e * Q->nrec + recb[d]) * 10;
            const double l0 = ld[3 + myslot], l1 = ld[4 + myslot];
            k0[d] = (d < ndep) ? l0 : 0.0;
            k1[d] = (d < ndep) ? l1 : 0.0;
        }
#pragma unroll
        for (int d = 0; d < 6; d++) {
            Kq[0] += k0[d] * tdep[d];
            Kq[1] += k1[d] * tdep[d];
        }
    }
    // What phase F2's projection needs of these tables, packed into two registers (six 5-bit bodies; count and the loop's two
    // minimal columns): held as ~10 separate values across phases D-F1 they were spilled to scratch, and every scratch reload at the
    // start of F2 is a load behind F1's thirty stores.  ind_loop_col is read here, before the first store, for the same reason.
    const int jdep_ld = Q->joint_dep[jnt];
    const int f1col = (LG == 32) ? (g & 15) : g;  // phase F1's column of this lane; its support list, fetched ahead of the stores too
    const unsigned long long cp0_ld = (DERIV && f1col < ni) ? Q->col_pack[f1col][0] : 0ull;
    const unsigned long long cp1_ld = (DERIV && f1col < ni) ? Q->col_pack[f1col][1] : 0ull;
    unsigned dep_pk = 0u;
#pragma unroll
    for (int d = 0; d < 6; d++) dep_pk |= (unsigned)depb[d] << (5 * d);
    const int lc0_ld = Q->ind_loop_col[tjc][0], lc1_ld = Q->ind_loop_col[tjc][1];
    unsigned aux_pk = (unsigned)ndep | ((unsigned)((tj >= 0 ? lc0_ld : -1) & 15) << 3) | ((unsigned)((tj >= 0 ? lc1_ld : -1) & 15) << 7);
    asm volatile("" : "+v"(dep_pk), "+v"(aux_pk));  // packed HERE (the compiler would sink the packing to F2 and spill the pieces)
    // the first round of customized rows: table entries before the stores as well
    const int cs0 = (Q->ncust > 0 && G != nullptr) ? cust_start[i] : 0, cs1 = (Q->ncust > 0 && G != nullptr) ? cust_start[i + 1] : 0;
    int crow0 = 0;
    DevCustRow cr0 = {};
    if (cs0 + g < cs1) {
        crow0 = cust_list[cs0 + g];
        cr0 = cust[crow0];
    }
    if (lamrec != nullptr && rc >= 0) lamrec[(size_t)e * (Q->nrec * (1 + 3 * ni)) + rc] = tt;
    if (has_joint) {
        rec[Q->c_q + jnt] = q;
        rec[Q->c_v + jnt] = qd;
        rec[Q->c_a + jnt] = qdd;
        if (G != nullptr) G[rowbase + Q->off_q + (size_t)i * nv + jnt] = wrap_to_pi(q);  // ConstrainedJointLimits.cpp:57
    }
    if (DERIV && rc >= 0) {  // P_dep rows of the closed-loop joints (= Y1, DigitDynamicsConstraints.cpp:394)
        double* pl = rec + Q->c_Pl + (jdep_ld - 6) * 2;
        pl[0] = w0[0];
        pl[1] = w1[0];
    }
    if (tj >= 0) {
        // actuated torque = G^T tau_tree: own joint plus the closed-loop joints this actuator drives
        // (== tau_indep - J_indep^T lambda, ConstrainedInverseDynamics.cpp:175)
        double tau = tt;
#pragma unroll
        for (int d = 0; d < 6; d++) tau += gdep[d] * tdep[d];
        rec[Q->c_tau + tj] = tau;
        if (G != nullptr) G[rowbase + Q->off_tau + (size_t)i * ni + tj] = tau;  // TorqueLimits.cpp:30-34
    }
    if (g == 0) {  // root: contact wrench lambda = [f ; A_des^T (n - p x f)]   (== tail of J_dep^-T tau_dep, :154)
        const V3 np = F.a - cross(pf, F.l);
        double lam[6] = {F.l.x, F.l.y, F.l.z,
                         Q->Ades[0] * np.x + Q->Ades[3] * np.y + Q->Ades[6] * np.z,
                         Q->Ades[1] * np.x + Q->Ades[4] * np.y + Q->Ades[7] * np.z,
                         Q->Ades[2] * np.x + Q->Ades[5] * np.y + Q->Ades[8] * np.z};
#pragma unroll
        for (int k = 0; k < 6; k++) rec[Q->c_lam + k] = lam[k];
        if (FUSE) {  // phase G reads it from the root body's vector slot, which nothing else uses
#pragma unroll
            for (int k = 0; k < 6; k++) sl[LL::V + k] = lam[k];
        }
        if (G != nullptr) {  // RectangleSurfaceContactConstraints.cpp:35-50
            double* gc = G + rowbase + Q->off_contact + (size_t)i * 7;
            const double ff = sqrt(lam[0] * lam[0] + lam[1] * lam[1]);
            gc[0] = lam[2];
            gc[1] = ff - mu * lam[2];
            gc[2] = fabs(lam[5]) - gam * lam[2];
            gc[3] = lam[3] - Lx * lam[2];
            gc[4] = -Lx * lam[2] - lam[3];
            gc[5] = lam[4] - Ly * lam[2];
            gc[6] = -Ly * lam[2] - lam[4];
        }
    }
    // The torso's lane and the swing foot's lane both need roll / pitch / yaw of a frame (three atan2 / asin evaluations of a
    // few hundred instructions): they run that code together, once per wave, instead of one lane after the other.
    const bool is_sf = Q->ncust > 0 && has_body && g == Q->sf_body;
    M3 Rsf;
#pragma unroll
    for (int k = 0; k < 9; k++) Rsf.m[k] = 0.0;
    V3 psf = v3(0, 0, 0);
    V3 rpy = v3(0, 0, 0);
    if (is_torso || is_sf) {
        M3 Rf = Rw;
        if (is_sf) {
            Rsf = mul(Rw, Q->sf_R);
            psf = ow + mul(Rw, v3(Q->sf_p[0], Q->sf_p[1], Q->sf_p[2]));
            Rf = Rsf;
        }
        rpy = v3(atan2(-Rf.m[5], Rf.m[8]), safe_asin(Rf.m[2]), atan2(-Rf.m[1], Rf.m[0]));  // Transform::getRPY (Transform.cpp:379-390)
    }
    // The torso's and the swing foot's poses are what phase F1's column lanes need: they go to the (still unused) per-column scratch
    // of F2 now, so that Rw / ow / Rsf / psf -- 24 doubles per lane -- do not stay live through phases E and F1 just to be shuffled.
    if (DERIV && is_torso) {
        double* o = sl + LL::TS;
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = Rw.m[k];
        o[9] = ow.x; o[10] = ow.y; o[11] = ow.z;
    }
    if (DERIV && is_sf) {
        double* o = sl + LL::TS + 12;
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = Rsf.m[k];
        o[9] = psf.x; o[10] = psf.y; o[11] = psf.z;
    }
    if (is_torso) {
        // floating-base coordinates from the torso pose (TalosDynamicsConstraints.cpp:208-222) and their rates
        const SV v = DERIV ? lds_sv(sl + LL::TS + 24) : vA, a = DERIV ? lds_sv(sl + LL::TS + 30) : aA;
        const V3 pd = v.l + cross(v.a, ow);
        const V3 rd = rpy_rates(Rw, v.a);
        const V3 al = a.a;  // gravity enters only the linear part of a0
        const V3 pdd = (a.l - a0.l) + cross(al, ow) + cross(v.a, pd);
        double sr, cr;
        sincos(rpy.x, &sr, &cr);
        const V3 a1 = v3(1, 0, 0), a2 = v3(0, cr, sr), a3 = col(Rw, 2);
        const V3 a2d = cross(a1 * rd.x, a2), a3d = cross(a1 * rd.x + a2 * rd.y, a3);
        const V3 rdd = rpy_rates(Rw, al - a2d * rd.y - a3d * rd.z);
        const double qv[6] = {ow.x, ow.y, ow.z, rpy.x, rpy.y, rpy.z};
        const double vv[6] = {pd.x, pd.y, pd.z, rd.x, rd.y, rd.z};
        const double av[6] = {pdd.x, pdd.y, pdd.z, rdd.x, rdd.y, rdd.z};
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int dj = Q->dep_joint[k];
            rec[Q->c_q + dj] = qv[k];
            rec[Q->c_v + dj] = vv[k];
            rec[Q->c_a + dj] = av[k];
            if (G != nullptr) G[rowbase + Q->off_q + (size_t)i * nv + dj] = wrap_to_pi(qv[k]);
        }
    }
    // customized block (DigitCustomizedConstraints.cpp:88-125 / TalosCustomizedConstraints.cpp:88-124): the swing foot's
    // lane evaluates this sample's rows from its own pose and the floating-base coordinates of the torso's lane
    if (Q->ncust > 0) {
        double X[12];
        {
            const double mine[6] = {ow.x, ow.y, ow.z, rpy.x, rpy.y, rpy.z};
#pragma unroll
            for (int k = 0; k < 6; k++) X[6 + k] = __shfl(mine[k], Q->torso_body, LG);
        }
        {
            const double mine[6] = {psf.x, psf.y, psf.z, rpy.x, rpy.y, rpy.z};
#pragma unroll
            for (int k = 0; k < 6; k++) X[k] = __shfl(mine[k], Q->sf_body, LG);
        }
        if (is_sf) {
#pragma unroll
            for (int k = 0; k < 6; k++) rec[Q->c_sf + k] = X[k];
        }
        if (G != nullptr) {  // this sample's rows, one per lane and round: the table look-ups of a row are a chain of dependent loads
            for (int rr = cs0 + g; rr < cs1; rr += LG) {
                const bool first = rr == cs0 + g;
                const int row = first ? crow0 : cust_list[rr];
                const DevCustRow cr = first ? cr0 : cust[row];
                double xa = 0.0, xb = 0.0;
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    xa = (k == cr.qa) ? X[k] : xa;
                    xb = (k == cr.qb) ? X[k] : xb;
                }
                const double val = cr.ca * xa + cr.cb * xb + cr.add;
                G[rowbase + Q->off_cust + row] = cr.wrap ? wrap_to_pi(val) : val;
            }
        }
    }
    if (!DERIV) return;
    LEG_PHASE_END(5, sv_sum(F) + sv_sum(BtS) + tt)


    // FUSE: what the column lanes produce stays on chip for phase G -- d(lambda) in registers (read there by shuffles with
    // constant register indices), the P_dep rows of the floating base and of the swing foot in registers until phase F2 has
    // finished with its scratch, then in LDS.  Neither the D blocks nor these go through the global record and back.
    // (PREC, LG = 32: these 30 values per column lane go through the record after all -- kept in registers across phase F2 the Digit
    // instance spilled 44 of them; they are 720 doubles per wavefront, read back once, against 1764 for the whole records)
#ifndef LEG_PREC16
#define LEG_PREC16 0
#endif
#ifndef LEG_PREC32
#define LEG_PREC32 0
#endif
    constexpr int PMODE = !FUSE ? 0 : (LG == 32 ? LEG_PREC32 : (LEG_PREC16 ? 2 : 0));  // 0: registers, 1: P rows via the record, 2: L too
    constexpr bool PREC = PMODE >= 1, LREC = PMODE >= 2;
    double keepL[FUSE ? 18 : 1], keepP[FUSE ? 12 : 1];
    if (FUSE) {
#pragma unroll
        for (int k = 0; k < 18; k++) keepL[FUSE ? k : 0] = 0.0;
#pragma unroll
        for (int k = 0; k < 12; k++) keepP[FUSE ? k : 0] = 0.0;
    }
    // ---- phase F1: column pass.  Lane t < ni = actuator column t: d(root wrench), torso twist, swing-foot twist per unit
    // (q, v, a) of actuator t, summed over the tree joints that move with it (<= 8).  These are whole-tree quantities with
    // no row structure, so one lane per column is the natural mapping.
    {
        const int tb = Q->torso_body, sb = (Q->ncust > 0) ? Q->sf_body : 0;
        const bool has_sf = Q->ncust > 0;
        // LG = 32: a column's (up to) eight support joints are shared by lanes t and t + 16, four each, and the halves are added by
        // one exchange afterwards; the loop ends when no lane of the wavefront has an entry left (Talos / G1: one pass instead of eight).
        // Reads are unconditional (an absent entry reads joint 1 with zero weights): under `if (on)` they were a divergent branch with
        // its own LDS round trip per pass.
        constexpr bool SPLIT = LG == 32;
        const int t_ = SPLIT ? (g & 15) : g;
        const bool upper = SPLIT && g >= 16;
        const bool col = t_ < ni && (!SPLIT || g < 16);   // the lanes that own a column's outputs
        const int t = (t_ < ni) ? t_ : 0;
        unsigned long long cp = (t_ < ni) ? (upper ? cp1_ld : cp0_ld) : 0ull;
        SV Fq = sv_zero(), Fv = sv_zero(), Fa = sv_zero(), twt = sv_zero(), tws = sv_zero();
        unsigned long long cpn = SPLIT ? 0ull : ((t_ < ni) ? cp1_ld : 0ull);  // unsplit: entries 4..7 follow
        const int has_cust = Q->ncust > 0;
#pragma unroll 1
        for (int r = 0; r < (SPLIT ? 4 : 8); r++) {
            if (!SPLIT && r == 4) cp = cpn;
            const unsigned ent = (unsigned)cp & 0xffffu;
            cp >>= 16;
            const bool on = (ent >> 12 & 1u) != 0;
            if (__builtin_amdgcn_ballot_w64(on) == 0ull && (SPLIT || r >= 4 || __builtin_amdgcn_ballot_w64(cpn != 0ull) == 0ull)) break;  // (uniform)
            const int k = on ? (int)(ent & 31u) : 1;
            const bool s1 = (ent >> 11 & 1u) != 0;
            // (both slots are fetched: the slot is the REQUESTING lane's choice, a select on the source side would use the source's)
            const double a0 = __shfl(w0[0], k, LG), a1 = __shfl(w0[1], k, LG), a2 = __shfl(w0[2], k, LG);
            const double b0 = __shfl(w1[0], k, LG), b1 = __shfl(w1[1], k, LG), b2 = __shfl(w1[2], k, LG);
            const double g0 = on ? (s1 ? b0 : a0) : 0.0, hv = on ? (s1 ? b1 : a1) : 0.0, hq = on ? (s1 ? b2 : a2) : 0.0;
            double* vk = sl + LL::V + k * LEG_VEC;
            v2d_t u[12];
#pragma unroll
            for (int c = 0; c < 3; c++) u[c] = *vec_chunk_ptr(vk, sw2, c);
#pragma unroll
            for (int c = 3; c < 12; c++) u[c] = *vec_chunk_ptr(vk, sw2, 6 + c);
            const SV sk = SV{v3(u[0].x, u[0].y, u[1].x), v3(u[1].y, u[2].x, u[2].y)};
            const SV ISk = SV{v3(u[3].x, u[3].y, u[4].x), v3(u[4].y, u[5].x, u[5].y)};
            const SV cvk = SV{v3(u[6].x, u[6].y, u[7].x), v3(u[7].y, u[8].x, u[8].y)};
            const SV cqk = SV{v3(u[9].x, u[9].y, u[10].x), v3(u[10].y, u[11].x, u[11].y)};
            Fa = Fa + ISk * g0;
            Fv = Fv + cvk * g0 + ISk * hv;
            Fq = Fq + cqk * g0 + cvk * (0.5 * hv) + ISk * hq;
            const int kend = (int)(ent >> 5 & 63u);
            const double gt = (tb >= k && tb < kend) ? g0 : 0.0;
            const double gs = (has_cust && sb >= k && sb < kend) ? g0 : 0.0;
            twt = twt + sk * gt;
            tws = tws + sk * gs;
        }
        if (SPLIT) {
#define XADD(x) x += __shfl_xor(x, 16, LG);
#define XADDSV(w) XADD(w.a.x) XADD(w.a.y) XADD(w.a.z) XADD(w.l.x) XADD(w.l.y) XADD(w.l.z)
            XADDSV(Fq) XADDSV(Fv) XADDSV(Fa) XADDSV(twt) XADDSV(tws)
#undef XADDSV
#undef XADD
        }
        LEG_PHASE_END(6, sv_sum(BtS) + Kq[0] + Kq[1] + sv_sum(Fq) + sv_sum(Fv) + sv_sum(Fa) + sv_sum(twt) + sv_sum(tws))
        // the torso's and the swing foot's poses, published in phase D (a wave_sync lies between): read after the loop -- held across it
        // two of these 24 doubles went to scratch, and their reloads sat behind this phase's own stores
        const double* tp = sl + LL::TS;
        M3 Rt, Rs;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            Rt.m[k] = tp[k];
            Rs.m[k] = has_sf ? tp[12 + k] : 0.0;
        }
        const V3 ot = v3(tp[9], tp[10], tp[11]);
        const V3 ps = has_sf ? v3(tp[21], tp[22], tp[23]) : v3(0, 0, 0);
        if (col) {
#pragma unroll
            for (int x = 0; x < 3; x++) {  // d(lambda.tail(6)) = root wrench about the contact point, RPY rows through A_des
                const SV w = (x == 0) ? Fq : ((x == 1) ? Fv : Fa);
                const V3 np = w.a - cross(pf, w.l);
                const double lm[6] = {w.l.x, w.l.y, w.l.z,
                                      Q->Ades[0] * np.x + Q->Ades[3] * np.y + Q->Ades[6] * np.z,
                                      Q->Ades[1] * np.x + Q->Ades[4] * np.y + Q->Ades[7] * np.z,
                                      Q->Ades[2] * np.x + Q->Ades[5] * np.y + Q->Ades[8] * np.z};
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    if (FUSE && !LREC) keepL[FUSE ? x * 6 + r : 0] = lm[r];
                    else rec[Q->c_L + (x * 6 + r) * ni + t] = lm[r];  // [x][r][it]: the column lanes write runs of 8 * ni bytes
                }
            }
            {   // P_dep rows of the floating base (= -J_dep^-1 J_indep, TalosDynamicsConstraints.cpp:243-246)
                const V3 lin = twt.l + cross(twt.a, ot);
                const V3 rot = rpy_rates(Rt, twt.a);
                const double pc[6] = {lin.x, lin.y, lin.z, rot.x, rot.y, rot.z};
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    rec[Q->c_P + k * ni + t] = pc[k];  // (the reset-map kernel reads these rows of the last sample)
                    if (FUSE && !PREC) keepP[FUSE ? k : 0] = pc[k];
                }
            }
            if (Q->ncust > 0) {  // swing-foot pose derivative (FK Jacobians times pq_pz, DigitCustomizedConstraints.cpp:80-82)
                const V3 lin = tws.l + cross(tws.a, ps);
                const V3 rot = rpy_rates(Rs, tws.a);
                const double pc[6] = {lin.x, lin.y, lin.z, rot.x, rot.y, rot.z};
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    if (FUSE && !PREC) keepP[FUSE ? 6 + k : 0] = pc[k];
                    else rec[Q->c_Psf + k * ni + t] = pc[k];
                }
            }
        }
    }

    LEG_PHASE_END(7, sv_sum(BtS) + Kq[0] + Kq[1])

    // ---- phase F2: row pass.  Lane = tree joint j (row), loop over actuator columns `it` and the joints k that move with
    // it.  One code path for both relations: with (A, U1, U2, U3) = (s_j; IS_k, cv_k, cq_k) when j is a strict ancestor of
    // k and (IS_j; s_k, sd_k, psi_k) when k is j or an ancestor of j,
    //     M_jk = A.U1,   dtau_j/dv_k = f A.U2 + BtS_j.U1,   dtau_j/dq_k = A.U3 + BtS_j.U2     (f = 1 / 2, BtS terms only in
    // the second case); both vector triples sit 18 doubles apart in the LDS slots, so the choice is an address offset.
    // The (column, joint) pairs come from one packed table, a lane each, read with v_readlane: no memory access decides
    // what the next step works on.
    // Two columns that share joints (Digit's toe_A / toe_B drive the same six joints of a toe loop) are walked TOGETHER, with two
    // accumulator sets: the pair quantities (five dot products, 18 doubles from LDS) of a shared joint do not depend on the column
    // and are formed once (42 -> 30 steps for Digit).  Entry: body | end << 5 | slot << 11 | last << 12 | mode << 13 | column of set 0
    // << 15 | column of set 1 << 19 (15: none); mode 0 / 1: into set 0 / 1 with weight slot `slot`, 2: into both, slot 0 / slot 1.
    const int flat = flat_pre;  // loaded in phase 0
    const int nflat = Q->flat_n;
    const int cD = Q->c_D;  // (read once: behind the wave_sync fences of the loop the compiler reloads every Q-> field it touches)
    // the shuffles below address lanes through a lane id formed HERE (two VALU): the one from the top of the kernel had been spilled,
    // and its reload -- a scratch load behind F1's thirty stores -- was the first thing this phase waited for
    int lane_f2;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_f2));
    const int grp_f2 = lane_f2 & ~(LG - 1);
    auto shfl_f2 = [&](const double x, const int k) {
        const int a = (grp_f2 | k) << 2;
        const long long xi = __double_as_longlong(x);
        const int lo = __builtin_amdgcn_ds_bpermute(a, (int)(xi & 0xffffffffll)), hi = __builtin_amdgcn_ds_bpermute(a, (int)(xi >> 32));
        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    };
    struct F2Ops {
        SV U1, U2, U3;
    };
    auto f2_fetch = [&](const int idx) {
        F2Ops P;
        const unsigned ent = (unsigned)__builtin_amdgcn_readlane(flat, idx);
        const int k = (int)(ent & 31u);
        const bool anc = has_joint && k > g && k < gend;
        // 16-byte reads (the slot, the wavefront's area and the anc offset are all multiples of 16 bytes): ds_read_b128 moves 256 B per
        // LDS clock, the ds_read2_b64 the compiler picks for 8-byte-aligned doubles half of that -- and the LDS is as busy as the VALU here
        // (the second half starts at chunk 9, an odd one: the pair swap changes direction with `anc`)
        double* vk = sl + LL::V + k * LEG_VEC + (anc ? 18 : 0);
        const int swk = anc ? -sw2 : sw2;
        v2d_t u[9];
#if defined(LEG_F2_PROBE) && LEG_F2_PROBE == 3  // measurement build: no vector reads
        for (int c = 0; c < 9; c++) u[c] = v2d_t{s.a.x + c, IS.l.y + k};
#else
#pragma unroll
        for (int c = 0; c < 9; c++) u[c] = *vec_chunk_ptr(vk, swk, c);
#endif
        P.U1 = SV{v3(u[0].x, u[0].y, u[1].x), v3(u[1].y, u[2].x, u[2].y)};
        P.U2 = SV{v3(u[3].x, u[3].y, u[4].x), v3(u[4].y, u[5].x, u[5].y)};
        P.U3 = SV{v3(u[6].x, u[6].y, u[7].x), v3(u[7].y, u[8].x, u[8].y)};
        return P;
    };
    // this joint's own s and IS come back from its LDS slot (written in phase E): held in registers from phase E on they were 24 of
    // the value phase's and F1's live registers
    SV sR = sv_zero(), ISR = sv_zero();
    if (has_joint) {
        double* own = sl + LL::V + g * LEG_VEC;
        const v2d_t o0 = *vec_chunk_ptr(own, sw2, 0), o1 = *vec_chunk_ptr(own, sw2, 1), o2 = *vec_chunk_ptr(own, sw2, 2);
        const v2d_t o9 = *vec_chunk_ptr(own, sw2, 9), o10 = *vec_chunk_ptr(own, sw2, 10), o11 = *vec_chunk_ptr(own, sw2, 11);
        sR = SV{v3(o0.x, o0.y, o1.x), v3(o1.y, o2.x, o2.y)};
        ISR = SV{v3(o9.x, o9.y, o10.x), v3(o10.y, o11.x, o11.y)};
    }
    double Tq = 0.0, Tv = 0.0, Ta = 0.0, Xq = 0.0, Xv = 0.0, Xa = 0.0;
    auto f2_step = [&](const int idx, const F2Ops& P) {
        const unsigned ent = (unsigned)__builtin_amdgcn_readlane(flat, idx);
        const int k = (int)(ent & 31u), kend = (int)(ent >> 5 & 63u);
        const bool s1 = (ent >> 11 & 1u) != 0;  // uniform
        const unsigned mode = ent >> 13 & 3u;   // uniform
        // chain-rule weights of joint k towards the step's column(s): the slot is chosen before the shuffle (uniform), the second
        // set is fetched only where two columns are walked together
        double wa[3], wb[3];
        {
            const bool sa = (mode == 2u) ? false : s1;
#if defined(LEG_F2_PROBE) && LEG_F2_PROBE == 2  // measurement build: no weight shuffles
            for (int c = 0; c < 3; c++) wa[c] = sa ? w1[c] : w0[c];
            if (false) {
#else
            if (sa) {  // a uniform branch, kept a branch (six selects per step otherwise: the row pass is bound by VALU issue)
                asm volatile("");
                for (int c = 0; c < 3; c++) wa[c] = shfl_f2(w1[c], k);
            } else {
                asm volatile("");
                for (int c = 0; c < 3; c++) wa[c] = shfl_f2(w0[c], k);
            }
            if (mode == 2u) {
#endif
#pragma unroll
                for (int c = 0; c < 3; c++) wb[c] = shfl_f2(w1[c], k);
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) wb[c] = wa[c];
            }
        }
        {
            const bool anc = has_joint && k > g && k < gend;     // this joint is a strict ancestor of k
            const bool dsc = has_joint && g >= k && g < kend;    // k is this joint or one of its ancestors
            const SV A = anc ? sR : ISR;  // registers: per-lane LDS slots 288 B apart would bank-conflict on 128-bit reads
            // six-term FMA chains (the row pass is bound by VALU issue: 30 instead of 35 instructions for the five products)
#define DOT6(p_, q_) __builtin_fma((p_).a.x, (q_).a.x, __builtin_fma((p_).a.y, (q_).a.y, __builtin_fma((p_).a.z, (q_).a.z, \
                     __builtin_fma((p_).l.x, (q_).l.x, __builtin_fma((p_).l.y, (q_).l.y, (p_).l.z * (q_).l.z)))))
            double da = DOT6(A, P.U1);
            const double t2 = DOT6(A, P.U2), t3 = DOT6(A, P.U3);
            const double e2 = DOT6(BtS, P.U1), e3 = DOT6(BtS, P.U2);
#undef DOT6
            double dv = anc ? t2 : 2.0 * t2 + e2;
            double dq = anc ? t3 : t3 + e3;
            if (g == k) {  // ConstrainedInverseDynamics.cpp:144-146 and the armature on the diagonal of M
                da += arm_j;
                dv += damp_j;
            }
            const double rel = (anc || dsc) ? 1.0 : 0.0;
            da *= rel; dv *= rel; dq *= rel;
            if (mode != 1u) {  // set 0
                const double g0 = wa[0], hv = wa[1], hq = wa[2];
                Ta += da * g0;
                Tv += dv * g0 + da * hv;
                Tq += dq * g0 + dv * (0.5 * hv) + da * hq;
            }
            if (mode != 0u) {  // set 1
                const double g0 = wb[0], hv = wb[1], hq = wb[2];
                Xa += da * g0;
                Xv += dv * g0 + da * hv;
                Xq += dq * g0 + dv * (0.5 * hv) + da * hq;
            }
        }
        if ((ent >> 12 & 1u) == 0) return;  // more joints move with this column / pair of columns (uniform)
#if defined(LEG_F2_PROBE) && LEG_F2_PROBE == 1  // measurement build: no projection / stores (sums kept alive by the last column's store)
        if (idx + 1 < nflat) return;
#endif
#pragma unroll 1
        for (int set = 0; set < 2; set++) {
            const int it = (int)(set == 0 ? (ent >> 15 & 15u) : (ent >> 19 & 15u));
            if (it == 15) continue;  // uniform
            // (FUSE) the body whose joint is actuator `it`: the lane of the first lane group that carries trajectory index `it`
            const int g_it = FUSE ? (int)(__builtin_ctzll(__builtin_amdgcn_ballot_w64(tj == it) | (1ull << 63)) & (LG - 1)) : 0;
            const double Sq = set == 0 ? Tq : Xq, Sv = set == 0 ? Tv : Xv, Sa = set == 0 ? Ta : Xa;
            {   // every lane writes its slot; a lane without a joint writes exact zeros (selected, not 0 * x: a non-finite vector of
                // some joint must not reach the actuators that read slot 0 for their absent entries)
                double* ts = sl + LL::TS + g * LL::TSS;
                *reinterpret_cast<v2d_t*>(ts) = v2d_t{has_joint ? Sq : 0.0, has_joint ? Sv : 0.0};
                ts[2] = has_joint ? Sa : 0.0;
            }
            if (lamrec != nullptr && rc >= 0) {
                const int nrec = Q->nrec;
                double* o = lamrec + (size_t)e * (nrec * (1 + 3 * ni)) + nrec + (it * 3) * nrec + rc;
                o[0] = Sq; o[nrec] = Sv; o[2 * nrec] = Sa;
            }
            wave_sync();
            if (tj >= 0) {  // project the rows: G^T
                double Rq = Sq, Rv = Sv, Ra = Sa;
                unsigned dpk = dep_pk, apk = aux_pk;
                asm volatile("" : "+v"(dpk), "+v"(apk));  // unpacked here, every time: hoisted out of the loop they are ten registers again
                // all reads in flight together, then selected: a read under `d < ndep` is a divergent branch with its own LDS round trip, six
                // in a row per column (round 2's form: the projection and its stores were 0.08 of the row pass's 0.30 ms)
                v2d_t tqv[6];
                double ta_[6];
#pragma unroll
                for (int d = 0; d < 6; d++) {
                    const double* ts = sl + LL::TS + (int)(dpk >> (5 * d) & 31u) * LL::TSS;
                    tqv[d] = *reinterpret_cast<const v2d_t*>(ts);
                    ta_[d] = ts[2];
                }
#pragma unroll
                for (int d = 0; d < 6; d++) {  // absent entries: body 0's slot (exact zeros, above) times gdep = 0
                    Rq += gdep[d] * tqv[d].x; Rv += gdep[d] * tqv[d].y; Ra += gdep[d] * ta_[d];
                }
                if (it == (int)(apk >> 3 & 15u)) Rq += Kq[0];
                if (it == (int)(apk >> 7 & 15u)) Rq += Kq[1];
                if (FUSE) {
                    // column `it` is done with the vectors of the joints it moves: its D block (3 ni <= 36 doubles) takes the slot of
                    // its own joint (that joint moves with no other column)
                    double* Dl = sl + LL::V + g_it * LEG_VEC + tj;
                    Dl[0] = Rq; Dl[ni] = Rv; Dl[2 * ni] = Ra;
                } else {
                    double* D = rec + cD + (it * 3) * ni + tj;  // the actuators' lanes write 8 * ni contiguous bytes per store
                    D[0] = Rq; D[ni] = Rv; D[2 * ni] = Ra;
                }
            }
            wave_sync();
        }
        Tq = 0.0; Tv = 0.0; Ta = 0.0; Xq = 0.0; Xv = 0.0; Xa = 0.0;
    };
    // (Measured, round 3: requesting step idx + 1's vectors before the arithmetic of step idx -- the loop unrolled by two, two operand
    // sets -- runs the record kernels 1.273 ms against 1.259: with two wavefronts per SIMD the other wavefront already covers the LDS
    // latency, and the second copy of the loop body costs more than it hides.)
#pragma unroll 1
    for (int idx = 0; idx < nflat; idx++) f2_step(idx, f2_fetch(idx));

    // ---- phase G (FUSE builds): every row of pg_pz but the reset map's, from what this wavefront has on chip (the D blocks in the LDS
    // slots of the actuators' joints, d(lambda) in the column lanes' registers, the P rows parked in LDS below).  The expansion is
    // bound by the write path and the phases above by fp64 issue: in one kernel the stores of one wavefront drain while the other
    // wavefront of the SIMD computes, which two launches (or two streams: the core kernel holds all of a CU's LDS) never did.
    // With two wavefronts per SIMD nothing hides a chain of dependent LDS look-ups per element (the flat element loop of
    // legged_expand_kernel run here took 0.97 ms on top of the core kernel's 0.66), so the mapping is lane = FIXED 16-byte column chunk:
    // the actuator(s) of the lane's two columns and their Bernstein weights per sample live in registers, a row costs 1-3 LDS reads at
    // constant offsets, a few FMAs and one store.  KEVEN (K even): both columns of a chunk belong to one actuator and share the reads.
    // Host conditions (legged_can_fuse): N even (a pair of sample-evals = consecutive samples of ONE decision vector), n even,
    // ni * K <= 128 (chunks 64.. of a row are structural zeros), n <= 256, ni <= 12 and room for the parked P rows, a sample has at most
    // LG customized rows.  Pairs of sample-evals are expanded one after the other (LG = 16: two pairs per wavefront).
    if constexpr (FUSE) {
        LEG_TR(trG)
        constexpr int NH = KEVEN ? 1 : 2;  // distinct actuators a chunk can touch
        const int K = Q->K, n = L.n;
        const int ndyn = ni * K, cpr = n >> 1;
        const int ntail = cpr - LEG_BLOCK;  // chunks of a row beyond the wavefront's 64 (> 0 for Digit: 66 chunks)
        const int cc = 2 * tid;
        const bool lane_on = tid < cpr;
        bool lv[2];
        int th[2], kh[2], bth[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            lv[h] = cc + h < ndyn;
            th[h] = lv[h] ? (cc + h) / K : 0;
            kh[h] = lv[h] ? cc + h - th[h] * K : 0;
            bth[h] = Q->ind_body[th[h]];  // the body whose vector slot holds the D block of column th (phase F2)
        }
        const int off_tau = Q->off_tau, off_q = Q->off_q, off_contact = Q->off_contact;
        const v2d_t zero2 = v2d_t{0.0, 0.0};
        // lane j < nv: what row j of the joint-limit block is (independent column, floating-base row, closed-loop row with its two
        // driver columns and the body that carries the joint) -- fetched by the lanes in parallel and handed out with v_readlane:
        // three dependent scalar loads per row were most of this phase's time
        int ri_info = 0, ri_c0 = -1, ri_c1 = -1, ri_body = 0;
        if (tid < nv) {
            const int ind = Q->joint_ind[tid];
            const int dd = Q->joint_dep[tid];
            ri_info = ind >= 0 ? ind : -1 - dd;
            if (ind < 0 && dd >= 6) { ri_c0 = Q->dep_col[dd][0]; ri_c1 = Q->dep_col[dd][1]; ri_body = Q->dep_body[dd]; }
        }
        const int ncust = Q->ncust, off_cust = Q->off_cust;
        // Where the twelve P rows (floating base 0..5, swing foot 6..11; ni doubles each) wait in a sample-eval's LDS area once phase F2
        // is done: its per-joint scratch (LG * TSS doubles), then the vector slots of the lanes without a body (nb .. LG - 1), three rows
        // per slot.  The actuators' slots hold the D blocks, the root body's slot lambda.  (Host: the rows fit, legged_can_fuse.)
        const int nts = (LG * LL::TSS) / ni;
        auto prow = [&](const int k) {  // (uniform)
            const int r = k - nts;
            const int s3 = (r >= 6) ? 2 : ((r >= 3) ? 1 : 0);
            return (k < nts) ? LL::TS + k * ni : LL::V + (nb + s3) * LEG_VEC + (r - 3 * s3) * ni;
        };
        wave_sync();  // phase F2's last reads of its scratch
        {
            const int t_ = (LG == 32) ? (g & 15) : g;
            if (t_ < ni && (LG != 32 || g < 16)) {  // the column lanes of phase F1
                double pk[12];
#pragma unroll
                for (int k = 0; k < 12; k++) pk[k] = PREC ? rec[((k < 6) ? Q->c_P + k * ni : Q->c_Psf + (k - 6) * ni) + t_] : keepP[PREC ? 0 : k];
#pragma unroll
                for (int k = 0; k < 12; k++) sl[prow(k) + t_] = pk[k];
            }
        }
        wave_sync();
        auto bcast = [&](const double x, const int l) {
            const long long xi = __double_as_longlong(x);
            return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(xi >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)xi, l));
        };
        auto shfl64 = [&](const double x, const int l) {  // any lane of the wavefront
            const long long xi = __double_as_longlong(x);
            const int lo = __builtin_amdgcn_ds_bpermute(l << 2, (int)(xi & 0xffffffffll)), hi = __builtin_amdgcn_ds_bpermute(l << 2, (int)(xi >> 32));
            return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
        };
#pragma unroll 1
        for (int rd = 0; rd < EPB / 2; rd++) {
            const long long ea = e0 + 2 * rd;
            if (ea >= E) break;  // (uniform)
            int ia = (int)(e0 - b0 * N) + 2 * rd;
            long long ba = b0;
            while (ia >= N) { ia -= N; ba++; }
            // customized rows of these two samples: lane g of lane group 2 rd + sx holds the g-th row of sample ia + sx
            const int l0 = 2 * rd * LG;  // first lane of the pair's two lane groups
            const double* sa[2] = {lds + (2 * rd) * LL::STRIDE, lds + (2 * rd + 1) * LL::STRIDE};  // the two sample-evals' LDS areas
            int cu_row = 0, cu_qa = 0, cu_qb = -1, cu_n = 0;
            double cu_ca = 0.0, cu_cb = 0.0;
            if (ncust > 0 && (eg >> 1) == rd) {
                const int cs = cust_start[ia + (eg & 1)];
                cu_n = cust_start[ia + (eg & 1) + 1] - cs;
                if (g < cu_n) {
                    cu_row = cust_list[cs + g];
                    const DevCustRow cr = cust[cu_row];
                    cu_qa = cr.qa; cu_qb = cr.qb; cu_ca = cr.ca; cu_cb = cr.cb;
                }
            }
            double W[2][3][2];  // [sample][order][column of the chunk]
#pragma unroll
            for (int sx = 0; sx < 2; sx++)
#pragma unroll
                for (int o = 0; o < 3; o++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const double w = basis[(size_t)((ia + sx) * 3 + o) * K + kh[h]];
                        W[sx][o][h] = lv[h] ? w : 0.0;  // the structural zeros beyond ni K columns are products with this zero (finite entries: exact)
                    }
#ifdef LEG_G_TRACE
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (rd == 0) { LEG_TR(trS) }
#endif
#if defined(LEG_G_PROBE) && LEG_G_PROBE == 2  // measurement build: the phase's set-up alone
            if (tid == 0) JAC[(size_t)ea] = sa[0][tid] + sa[1][5] + W[0][0][0] + W[1][2][1];
            continue;
#endif
            double* base = JAC + (size_t)ba * L.m * n + cc;
            // Every piece below works in blocks of rows: all LDS reads of a block are issued before the first of them is used (with two
            // wavefronts per SIMD nothing else hides an LDS round trip per row), rows past the end repeat the last row's reads and store nothing.
            // A row is cpr 16-byte chunks, the wavefront stores 64 of them per instruction: the chunks beyond (structural zeros, ntail per
            // row) follow right after their block -- left to the end of the piece they were 32-byte holes in lines that L2 had written
            // back before the hole was filled (measured: the kernel is SLOWER without the tail stores, 1.24 -> 1.37 ms; after every row
            // instead of every block: 1.18 -> 1.22 ms).
            const int tl_u = (ntail > 0) ? tid / ntail : LEG_BLOCK, tl_c = (ntail > 0) ? 2 * (LEG_BLOCK + tid % ntail) - cc : 0;
            auto tail_rows = [&](double* rows, const int count) {  // rows: address of the lane's chunk in the first row of the run
                if (ntail > 0 && LEG_G_ON(tl_u < count)) g_st(rows + (size_t)tl_u * n + tl_c, zero2);
            };
            // -- TorqueLimits.cpp:36-40 on ConstrainedInverseDynamics::ptau_pz: D[x][row] of column t from the slot of t's own joint
            {
                double* dst = base + ((size_t)off_tau + (size_t)ia * ni) * n;
#pragma unroll
                for (int sx = 0; sx < 2; sx++) {
                    const double* dh[2] = {sa[sx] + LL::V + bth[0] * LEG_VEC, sa[sx] + LL::V + bth[1] * LEG_VEC};
#pragma unroll 1
                    for (int r0 = 0; r0 < ni; r0 += 6) {
                        double d0[6][NH], d1[6][NH], d2[6][NH];
#pragma unroll
                        for (int u = 0; u < 6; u++) {
                            const int rr = min(r0 + u, ni - 1);
#pragma unroll
                            for (int h = 0; h < NH; h++) { d0[u][h] = dh[h][rr]; d1[u][h] = dh[h][ni + rr]; d2[u][h] = dh[h][2 * ni + rr]; }
                        }
#pragma unroll
                        for (int u = 0; u < 6; u++) {
                            const double v0 = d0[u][0] * W[sx][0][0] + d1[u][0] * W[sx][1][0] + d2[u][0] * W[sx][2][0];
                            const double v1 = d0[u][NH - 1] * W[sx][0][1] + d1[u][NH - 1] * W[sx][1][1] + d2[u][NH - 1] * W[sx][2][1];
                            if (LEG_G_ON(lane_on && r0 + u < ni)) g_st(dst + (size_t)(sx * ni + r0 + u) * n, v2d_t{v0, v1});
                        }
                        tail_rows(dst + (size_t)(sx * ni + r0) * n, min(6, ni - r0));
                    }
                }
            }
            if (rd == 0) { LEG_TR(trA) }
            // -- ConstrainedJointLimits.cpp:59-68: row j is B_k at the joint's own column (independent joint) or P_dep(j, t) B_k
            // (floating-base rows: the P rows parked in LDS; closed-loop rows: Y1 of the joint, still in the registers of its body's lane).
            // The kind of a row is uniform: one scalar branch per row, and the commonest kind -- an independent joint -- touches neither
            // LDS nor another lane (as one branch-free block of rows every row paid for all three kinds: 60 instructions a row).
            {
                double* dst = base + ((size_t)off_q + (size_t)ia * nv) * n;
#pragma unroll 1
                for (int j = 0; j < nv; j++) {
                    const int info = __builtin_amdgcn_readlane(ri_info, j);  // (uniform)
                    double m[2][2];
                    if (info >= 0) {
                        bool me[NH];
#pragma unroll
                        for (int h = 0; h < NH; h++) me[h] = info == th[h];
#pragma unroll
                        for (int sx = 0; sx < 2; sx++) {  // (a dead column's weight is zero)
                            m[sx][0] = me[0] ? W[sx][0][0] : 0.0;
                            m[sx][1] = me[NH - 1] ? W[sx][0][1] : 0.0;
                        }
                    } else if (info >= -6) {
                        const int off = prow(-1 - info);
#pragma unroll
                        for (int sx = 0; sx < 2; sx++) {
                            double pb[NH];
#pragma unroll
                            for (int h = 0; h < NH; h++) pb[h] = sa[sx][off + th[h]];
                            m[sx][0] = pb[0] * W[sx][0][0];
                            m[sx][1] = pb[NH - 1] * W[sx][0][1];
                        }
                    } else {
                        const int c0 = __builtin_amdgcn_readlane(ri_c0, j), c1 = __builtin_amdgcn_readlane(ri_c1, j);
                        const int bj = __builtin_amdgcn_readlane(ri_body, j);
#pragma unroll
                        for (int sx = 0; sx < 2; sx++) {
                            const double p0 = bcast(w0[0], l0 + sx * LG + bj), p1 = bcast(w1[0], l0 + sx * LG + bj);
                            double pl[NH];
#pragma unroll
                            for (int h = 0; h < NH; h++) pl[h] = (th[h] == c0) ? p0 : ((th[h] == c1) ? p1 : 0.0);
                            m[sx][0] = pl[0] * W[sx][0][0];
                            m[sx][1] = pl[NH - 1] * W[sx][0][1];
                        }
                    }
#pragma unroll
                    for (int sx = 0; sx < 2; sx++)
                        if (LEG_G_ON(lane_on)) g_st(dst + (size_t)(sx * nv + j) * n, v2d_t{m[sx][0], m[sx][1]});
                    if (j % 6 == 5 || j == nv - 1) {  // the tail chunks of the last (up to) six rows, both samples
                        const int j0 = j - j % 6;
                        tail_rows(dst + (size_t)j0 * n, j - j0 + 1);
                        tail_rows(dst + (size_t)(nv + j0) * n, j - j0 + 1);
                    }
                }
            }
            if (rd == 0) { LEG_TR(trB) }
            // -- RectangleSurfaceContactConstraints.cpp:52-93: d(lambda.tail(6)) per unit (q, v, a) of actuator t from the registers
            // of t's column lane (phase F1), lambda from the root body's slot
            {
                double* dst = base + ((size_t)off_contact + (size_t)ia * 7) * n;
                double p[2][2][6];
#pragma unroll
                for (int sx = 0; sx < 2; sx++)
#pragma unroll
                    for (int q6 = 0; q6 < 6; q6++) {
                        double lq[3][NH];
#pragma unroll
                        for (int h = 0; h < NH; h++)
#pragma unroll
                            for (int x = 0; x < 3; x++)
                                lq[x][h] = LREC ? core[(size_t)(ea + sx) * Q->c_stride + Q->c_L + (x * 6 + q6) * ni + th[h]]
                                                : shfl64(keepL[LREC ? 0 : x * 6 + q6], l0 + sx * LG + th[h]);
                        p[sx][0][q6] = lq[0][0] * W[sx][0][0] + lq[1][0] * W[sx][1][0] + lq[2][0] * W[sx][2][0];
                        p[sx][1][q6] = lq[0][NH - 1] * W[sx][0][1] + lq[1][NH - 1] * W[sx][1][1] + lq[2][NH - 1] * W[sx][2][1];
                    }
#pragma unroll
                for (int sx = 0; sx < 2; sx++) {
                    const double* lam = sa[sx] + LL::V;
#pragma unroll
                    for (int rr = 0; rr < 7; rr++) {
                        const double v0 = contact_row(rr, lam, p[sx][0], mu, gam, Lx, Ly);
                        const double v1 = contact_row(rr, lam, p[sx][1], mu, gam, Lx, Ly);
                        if (LEG_G_ON(lane_on)) g_st(dst + (size_t)(sx * 7 + rr) * n, v2d_t{v0, v1});
                    }
                    tail_rows(dst + (size_t)(sx * 7) * n, 7);
                }
            }
            if (rd == 0) { LEG_TR(trC) }
            // -- *CustomizedConstraints.cpp (swing-foot / torso rows): row = (ca X[qa] + cb X[qb]) differentiated, X = [swing foot | base].
            // The block is ordered by quantity, then sample: the rows of one quantity at these two samples are adjacent, so the walk is
            // quantity-major -- two quantities x two samples per block, each pair of adjacent rows stored back to back.
            if (ncust > 0) {
                double* dst = base + (size_t)off_cust * n;
                const int n0 = __builtin_amdgcn_readlane(cu_n, l0), n1 = __builtin_amdgcn_readlane(cu_n, l0 + LG);
#pragma unroll 1
                for (int r0 = 0; r0 < max(n0, n1); r0 += 2) {
                    double da[4][NH], db[4][NH], ca[4], cb[4];
                    int row[4];
                    bool hasb[4], on[4];
#pragma unroll
                    for (int w = 0; w < 4; w++) {  // w = 2 * (quantity of the block) + sample
                        const int sx = w & 1, r = r0 + (w >> 1), cnt = sx ? n1 : n0;
                        on[w] = r < cnt;
                        const int sl_ = l0 + sx * LG + (on[w] ? r : 0);
                        row[w] = __builtin_amdgcn_readlane(cu_row, sl_);
                        const int qa = __builtin_amdgcn_readlane(cu_qa, sl_), qb = __builtin_amdgcn_readlane(cu_qb, sl_);
                        ca[w] = bcast(cu_ca, sl_);
                        cb[w] = bcast(cu_cb, sl_);
                        hasb[w] = qb >= 0;
                        const int qbb = hasb[w] ? qb : 0;
                        const int oa = prow((qa < 6) ? 6 + qa : qa - 6), ob = prow((qbb < 6) ? 6 + qbb : qbb - 6);  // X[0..5]: swing foot, X[6..11]: base
#pragma unroll
                        for (int h = 0; h < NH; h++) {
                            da[w][h] = sa[sx][oa + th[h]];
                            db[w][h] = sa[sx][ob + th[h]];
                        }
                    }
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const double cfa = ca[w] * da[w][0] + cb[w] * (hasb[w] ? db[w][0] : 0.0);
                        const double cfb = ca[w] * da[w][NH - 1] + cb[w] * (hasb[w] ? db[w][NH - 1] : 0.0);
                        if (LEG_G_ON(lane_on && on[w]))
                            g_st(dst + (size_t)row[w] * n, v2d_t{cfa * W[w & 1][0][0], cfb * W[w & 1][0][1]});
                    }
                    if (ntail > 0) {
                        const int rw = (tl_u == 0) ? row[0] : ((tl_u == 1) ? row[1] : ((tl_u == 2) ? row[2] : row[3]));
                        const bool ok = (tl_u == 0) ? on[0] : ((tl_u == 1) ? on[1] : ((tl_u == 2) ? on[2] : (tl_u == 3 && on[3])));
                        if (LEG_G_ON(ok)) g_st(dst + (size_t)rw * n + tl_c, zero2);
                    }
                }
            }
            if (rd == 0) { LEG_TR(trD) }
        }
#ifdef LEG_G_TRACE
        if (tid == 0 && G != nullptr) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long trE = __builtin_amdgcn_s_memrealtime();
            double* o = G + rowbase + Q->off_tau + (size_t)i * ni;
            o[0] = (double)(trG - tr0); o[1] = (double)(trS - trG); o[2] = (double)(trA - trS); o[3] = (double)(trB - trA);
            o[4] = (double)(trC - trB); o[5] = (double)(trD - trC); o[6] = (double)(trE - trD); o[7] = (double)(trE - tr0);
        }
#endif
    }
}

__global__ __launch_bounds__(LEG_EXP_BLOCK) void legged_expand_kernel(const DevTraj* __restrict__ T,
                                                                     const DevLegged* __restrict__ Q,
                                                                     const double* __restrict__ basis, const DevLayout L,
                                                                     const double* __restrict__ core,
                                                                     const DevCustRow* __restrict__ cust, const int spw,
                                                                     const int cper, const int pieces, double* __restrict__ JAC) {
    // pieces == 0: the customized block only (the other three came out of the fused core kernel)
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int N = T->N;
    const int groups = (N + spw - 1) / spw;   // workgroups per decision vector, spw (1 or 2) consecutive samples each
    const long long b = blockIdx.x / groups;
    const int i = (int)(blockIdx.x - b * groups) * spw;
    const int cnt = (i + spw <= N) ? spw : N - i;
    const long long e = b * N + i;
    const int ni = Q->ni, nv = Q->nv, nd = Q->nd, K = Q->K, n = L.n;
    const int ndyn = ni * K;
    const int stride = Q->c_stride;
    double* rec = lds;                                            // [spw][stride]
    double* wB = lds + spw * stride;                              // [spw][3][ndyn]
    int* tcol = reinterpret_cast<int*>(wB + spw * 3 * ndyn + cper * ni + N * K);  // [ndyn]
    int* rowinfo = tcol + ndyn;                                   // [MAX_NV] row kind, then [MAX_NV][2] dep_col
    int* rsample = rowinfo + 3 * RAPTOR_MAX_NV;                   // [cper] sample of each customized row of this workgroup's run
    double* coef = wB + spw * 3 * ndyn;                           // [cper][ni]
    double* B0 = coef + cper * ni;                                // [N][K] order-0 weights of every sample
    const int r0 = (int)(blockIdx.x - b * groups) * cper;         // this workgroup's run of customized rows
    const int crows = max(0, min(cper, Q->ncust - r0));
    const double* src = core + (size_t)e * stride;                // the records of consecutive samples are adjacent
    if (pieces)
        for (int k = tid; k < cnt * stride; k += LEG_EXP_BLOCK) rec[k] = src[k];
    for (int c = tid; c < (pieces ? cnt : 1) * ndyn; c += LEG_EXP_BLOCK) {
        const int sidx = (c >= ndyn) ? 1 : 0;
        const int cc = c - sidx * ndyn;
        const int t = cc / K, k = cc - t * K;
        const double* Bs = basis + (size_t)((i + sidx) * 3) * K + k;
        double* w = wB + sidx * 3 * ndyn;
        w[cc] = Bs[0];
        w[ndyn + cc] = Bs[K];
        w[2 * ndyn + cc] = Bs[2 * K];
        if (sidx == 0) tcol[cc] = t;
    }
    for (int k = tid; k < crows * ni; k += LEG_EXP_BLOCK) {
        const int r = k / ni, t = k - r * ni;
        const DevCustRow cr = cust[r0 + r];
        const double* rs = core + (size_t)(b * N + cr.sample) * stride;
        const double* Pb = rs + Q->c_P;
        const double* Ps = rs + Q->c_Psf;
        const double da = (cr.qa < 6) ? Ps[cr.qa * ni + t] : Pb[(cr.qa - 6) * ni + t];
        const double db = (cr.qb < 0) ? 0.0 : ((cr.qb < 6) ? Ps[cr.qb * ni + t] : Pb[(cr.qb - 6) * ni + t]);
        coef[k] = cr.ca * da + cr.cb * db;
        if (t == 0) rsample[r] = cr.sample;
    }
    if (crows > 0)
        for (int k = tid; k < N * K; k += LEG_EXP_BLOCK) B0[k] = basis[(size_t)((k / K) * 3) * K + (k % K)];
    if (tid < nv) rowinfo[tid] = (Q->joint_ind[tid] >= 0) ? Q->joint_ind[tid] : -1 - Q->joint_dep[tid];
    if (tid < nd) {
        rowinfo[RAPTOR_MAX_NV + 2 * tid] = Q->dep_col[tid][0];
        rowinfo[RAPTOR_MAX_NV + 2 * tid + 1] = Q->dep_col[tid][1];
    }
    __syncthreads();
    // the row blocks of these samples are contiguous pieces of the output; 16-byte aligned because n is even
    double* base = JAC + (size_t)b * L.m * n;
    if (pieces) {
    legged_expand_piece<0>(tid, n, ndyn, ni, ni, cnt, base + ((size_t)Q->off_tau + (size_t)i * ni) * n, rec, stride, wB, tcol, rowinfo, Q);
    legged_expand_piece<1>(tid, n, ndyn, ni, nv, cnt, base + ((size_t)Q->off_q + (size_t)i * nv) * n, rec, stride, wB, tcol, rowinfo, Q);
    legged_expand_piece<2>(tid, n, ndyn, ni, 7, cnt, base + ((size_t)Q->off_contact + (size_t)i * 7) * n, rec, stride, wB, tcol, rowinfo, Q);
    }
    if (crows > 0)
        legged_expand_cust(tid, n, ndyn, ni, K, crows, base + ((size_t)Q->off_cust + r0) * n, coef, rsample, B0, tcol);
}

// fallback for an odd number of columns: one entry per lane
__global__ __launch_bounds__(RAPTOR_BLOCK) void legged_expand_scalar_kernel(const DevTraj* __restrict__ T,
                                                                            const DevLegged* __restrict__ Q,
                                                                            const double* __restrict__ basis, const DevLayout L,
                                                                            const double* __restrict__ core,
                                                                            const DevCustRow* __restrict__ cust, const long long total,
                                                                            double* __restrict__ JAC) {
    const long long w = (long long)blockIdx.x * RAPTOR_BLOCK + threadIdx.x;
    if (w >= total) return;
    const int n = L.n, N = T->N, ni = Q->ni, nv = Q->nv, K = Q->K;
    const int c = (int)(w % n);
    const long long br = w / n;
    const int r = (int)(br % L.m);
    const long long b = br / L.m;
    if (Q->nperiod > 0 && r >= Q->off_period) return;  // the reset-map rows (if part of this problem) are written by legged_period_kernel
    double val = 0.0;
    if (c < ni * K) {
        const int t = c / K, k = c - t * K;
        int i, lr, piece;
        if (Q->ncust > 0 && r >= Q->off_cust && r < Q->off_cust + Q->ncust) { piece = 3; lr = r - Q->off_cust; i = cust[lr].sample; }
        else if (r >= Q->off_tau && r < Q->off_tau + N * ni) { piece = 0; i = (r - Q->off_tau) / ni; lr = (r - Q->off_tau) - i * ni; }
        else if (r >= Q->off_q && r < Q->off_q + N * nv) { piece = 1; i = (r - Q->off_q) / nv; lr = (r - Q->off_q) - i * nv; }
        else { piece = 2; i = (r - Q->off_contact) / 7; lr = (r - Q->off_contact) - i * 7; }
        const double* Bs = basis + (size_t)(i * 3) * K + k;
        const double B0 = Bs[0], B1 = Bs[K], B2 = Bs[2 * K];
        const double* rec = core + (size_t)(b * N + i) * Q->c_stride;
        if (piece == 0) {
            const double* d = rec + Q->c_D + (t * 3) * ni + lr;
            val = d[0] * B0 + d[ni] * B1 + d[2 * ni] * B2;
        } else if (piece == 1) {
            const int ti = Q->joint_ind[lr];
            if (ti >= 0) val = (ti == t) ? B0 : 0.0;
            else {
                const int dd = Q->joint_dep[lr];
                if (dd < 6) val = rec[Q->c_P + dd * ni + t] * B0;
                else val = ((t == Q->dep_col[dd][0]) ? rec[Q->c_Pl + (dd - 6) * 2] : ((t == Q->dep_col[dd][1]) ? rec[Q->c_Pl + (dd - 6) * 2 + 1] : 0.0)) * B0;
            }
        } else if (piece == 3) {
            const DevCustRow cr = cust[lr];
            const double* Pb = rec + Q->c_P;
            const double* Ps = rec + Q->c_Psf;
            const double da = (cr.qa < 6) ? Ps[cr.qa * ni + t] : Pb[(cr.qa - 6) * ni + t];
            const double db = (cr.qb < 0) ? 0.0 : ((cr.qb < 6) ? Ps[cr.qb * ni + t] : Pb[(cr.qb - 6) * ni + t]);
            val = (cr.ca * da + cr.cb * db) * B0;
        } else {
            double p[6];
            const double* l = rec + Q->c_L + t;
#pragma unroll
            for (int rr = 0; rr < 6; rr++) p[rr] = l[rr * ni] * B0 + l[(6 + rr) * ni] * B1 + l[(12 + rr) * ni] * B2;
            val = contact_row(lr, rec + Q->c_lam, p, Q->mu, Q->gamma, Q->Lx, Q->Ly);
        }
    }
    JAC[w] = val;
}

// Costs::compute reductions over the N samples of one decision vector, from the per-sample records:
// one workgroup per (z, cost), one lane per column of z.  MinimizePower.cpp:32-54, MinimizeTorque.cpp:32-50,
// MinimizeInitialVelocity.cpp:18-30 (the Initial* costs read the first sample's trajectory values).
__global__ __launch_bounds__(RAPTOR_BLOCK) void legged_cost_kernel(const DevTraj* __restrict__ T, const DevLegged* __restrict__ Q,
                                                                   const double* __restrict__ basis, const int n,
                                                                   const double* __restrict__ core, const int num_costs,
                                                                   const int* __restrict__ kinds, double* __restrict__ F,
                                                                   double* __restrict__ GRAD) {
    const long long b = blockIdx.x / num_costs;
    const int kc = (int)(blockIdx.x - b * num_costs);
    const int kind = kinds[kc];
    const int N = T->N, ni = Q->ni, K = Q->K;
    const int ndyn = ni * K;
    for (int c = threadIdx.x; c < n || c == 0; c += RAPTOR_BLOCK) {  // lane 0 also owns f
        const bool live = c < ndyn;
        const int t = live ? c / K : 0, k = live ? c - t * K : 0;
        double f = 0.0, g = 0.0;
        if (kind == RAPTOR_COST_MINIMIZE_POWER || kind == RAPTOR_COST_MINIMIZE_TORQUE) {
            for (int i = 0; i < N; i++) {
                const double* rec = core + (size_t)(b * N + i) * Q->c_stride;
                const double* Bs = basis + (size_t)(i * 3) * K + k;
                const double B0 = Bs[0], B1 = Bs[K], B2 = Bs[2 * K];
                const double* D = rec + Q->c_D + (t * 3) * ni;
                double sq = 0.0, s = 0.0;
                for (int j = 0; j < ni; j++) {
                    const double tau = rec[Q->c_tau + j];
                    const double ptau = D[j] * B0 + D[ni + j] * B1 + D[2 * ni + j] * B2;
                    if (kind == RAPTOR_COST_MINIMIZE_POWER) {
                        const double v = rec[Q->c_v + Q->ind_joint[j]];
                        const double power = tau * v;
                        sq += power * power;
                        s += (tau * ((j == t) ? B1 : 0.0) + v * ptau) * power;
                    } else {
                        sq += tau * tau;
                        s += ptau * tau;
                    }
                }
                const double nrm = sqrt(sq);
                if (kind == RAPTOR_COST_MINIMIZE_POWER) {
                    if (sq > 1e-8) { f += nrm; g += s / nrm; }
                } else {
                    f += nrm;
                    if (nrm > 1e-8) g += s / nrm;
                }
            }
            f /= N;
            g /= N;
        } else {
            const bool vel = kind == RAPTOR_COST_MINIMIZE_INITIAL_VELOCITY;
            const double* rec = core + (size_t)(b * N) * Q->c_stride;
            const double* r0 = rec + (vel ? Q->c_v : Q->c_a);
            double sq = 0.0;
            for (int j = 0; j < ni; j++) sq += r0[Q->ind_joint[j]] * r0[Q->ind_joint[j]];
            const double nrm = sqrt(sq);
            f = nrm;
            if (nrm > 1e-8) g = basis[(vel ? 1 : 2) * K + k] * r0[Q->ind_joint[t]] / nrm;
        }
        if (c == 0) F[b * num_costs + kc] = f;
        if (GRAD != nullptr && c < n) GRAD[((size_t)b * num_costs + kc) * n + c] = live ? g : 0.0;
    }
}

// ConstrainedInverseDynamics::ptau_pz / plambda_pz view for the parity entry point: [B][N][rows][n]
__global__ __launch_bounds__(RAPTOR_BLOCK) void legged_expand_id_kernel(const DevTraj* __restrict__ T,
                                                                        const DevLegged* __restrict__ Q,
                                                                        const double* __restrict__ basis, const int n,
                                                                        const double* __restrict__ core, const long long total,
                                                                        const int rows, const int lambda_rows,
                                                                        double* __restrict__ out) {
    const long long w = (long long)blockIdx.x * RAPTOR_BLOCK + threadIdx.x;
    if (w >= total) return;
    const int c = (int)(w % n);
    const long long er = w / n;
    const int lr = (int)(er % rows);
    const long long e = er / rows;
    const int ni = Q->ni, K = Q->K;
    double val = 0.0;
    if (c < ni * K) {
        const int t = c / K, k = c - t * K;
        const int i = (int)(e % T->N);
        const double* Bs = basis + (size_t)(i * 3) * K + k;
        const double* rec = core + (size_t)e * Q->c_stride;
        // D[it][x][row] for the torques, L[x][row][it] for lambda
        const double* D = lambda_rows ? rec + Q->c_L + lr * ni + t : rec + Q->c_D + (t * 3) * ni + lr;
        const int step = lambda_rows ? 6 * ni : ni;
        val = D[0] * Bs[0] + D[step] * Bs[K] + D[2 * step] * Bs[2 * K];
    }
    out[w] = val;
}

#endif
