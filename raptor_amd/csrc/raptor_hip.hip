// libraptor_hip.so -- C ABI (include/raptor_hip.h) over the gfx950 kernels in kernels.h.
// Host side only does set-up (constant tables, layout, bounds) and launches; there is NO CPU fallback:
// every evaluation entry point runs on the GPU or returns RAPTOR_ERR_HIP.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "capsule_kernel.h"
#include "dual_kernel.h"
#include "kinova_kernel.h"
#include "legged_kernel.h"
#include "digit_kernel.h"
#include "period_kernel.h"
#include "kinova_hess_kernel.h"
#include "kinova_hess_torque_kernel.h"
#include "sphere_rows_kernel.h"
#include "kinova_hess_blocks_kernel.h"
#include "digit_constants.h"
#include "digit_tables.h"

namespace {

thread_local std::string g_err;

struct hip_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct unsupported_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                               \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            throw hip_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));                \
    } while (0)

#define API_TRY try {
#define API_CATCH                                   \
    }                                               \
    catch (const std::invalid_argument& e) {        \
        g_err = e.what();                           \
        return RAPTOR_ERR_INVALID_ARGUMENT;         \
    }                                               \
    catch (const hip_error& e) {                    \
        g_err = e.what();                           \
        return RAPTOR_ERR_HIP;                      \
    }                                               \
    catch (const unsupported_error& e) {            \
        g_err = e.what();                           \
        return RAPTOR_ERR_UNSUPPORTED;              \
    }                                               \
    catch (const std::exception& e) {               \
        g_err = e.what();                           \
        return RAPTOR_ERR_RUNTIME;                  \
    }                                               \
    return RAPTOR_OK;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t count = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
    }
    void ensure(size_t n) {
        if (n <= count) return;
        release();
        HIP_CHECK(hipMalloc((void**)&p, n * sizeof(T)));
        count = n;
    }
    void upload(const T* src, size_t n) {
        ensure(n);
        HIP_CHECK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    }
};

inline double deg2rad(double d) { return d * M_PI / 180.0; }  // Utils.h:17-19

// What the *_host entry points need beside the device staging buffers: a private non-blocking stream (so that a host-buffer
// call waits for its own work only, never for the device) and one pinned host buffer for transfers small enough to be
// latency-bound (the Ipopt-driven single-problem calls); large batches go straight from / to the caller's pageable memory.
struct HostStage {
    hipStream_t st = nullptr;
    void* pin = nullptr;
    size_t pin_bytes = 0;
    static constexpr size_t kPinMax = (size_t)64 << 20;
    static constexpr size_t kZeroCopyMax = (size_t)4 << 20;  // below this the kernels work on the pinned buffer itself
    HostStage() {}
    HostStage(const HostStage&) = delete;
    HostStage& operator=(const HostStage&) = delete;
    ~HostStage() {
        if (pin) (void)hipHostFree(pin);
        for (int k = 0; k < 2; k++) {
            if (chunk[k]) (void)hipHostFree(chunk[k]);
            if (chunk_ev[k]) (void)hipEventDestroy(chunk_ev[k]);
        }
        if (st) (void)hipStreamDestroy(st);
    }
    hipStream_t stream() {
        if (!st) HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        return st;
    }
    // Results of batches (beyond the latency path, kZeroCopyMax): device -> two pinned chunk buffers by the copy engine, each chunk copied on to the caller's
    // (pageable) buffer by a few host threads while the next one is in flight.  A direct hipMemcpyAsync into pageable memory
    // reaches 17 GB/s on this platform; the pipeline is bound by the pinned transfer instead.
    static constexpr size_t kChunk = (size_t)32 << 20;
    void* chunk[2] = {nullptr, nullptr};
    hipEvent_t chunk_ev[2] = {nullptr, nullptr};
    static void host_copy(char* dst, const char* src, size_t bytes) {
        const unsigned hw = std::thread::hardware_concurrency();
        const int nt = bytes < ((size_t)4 << 20) ? 1 : (int)std::max(1u, std::min(4u, hw ? hw : 1u));
        if (nt == 1) {
            std::memcpy(dst, src, bytes);
            return;
        }
        std::vector<std::thread> th;
        const size_t per = ((bytes + nt - 1) / nt + 63) & ~(size_t)63;
        for (int k = 1; k < nt; k++) {
            const size_t lo = std::min(bytes, (size_t)k * per), hi = std::min(bytes, (size_t)(k + 1) * per);
            if (hi > lo) th.emplace_back([=] { std::memcpy(dst + lo, src + lo, hi - lo); });
        }
        std::memcpy(dst, src, std::min(bytes, per));
        for (std::thread& t : th) t.join();
    }
    void download(void* dst, const void* dsrc, size_t bytes) {  // synchronous: returns when dst is complete
        hipStream_t s_ = stream();
        for (int k = 0; k < 2; k++) {
            if (!chunk[k]) HIP_CHECK(hipHostMalloc(&chunk[k], kChunk, hipHostMallocDefault));
            if (!chunk_ev[k]) HIP_CHECK(hipEventCreateWithFlags(&chunk_ev[k], hipEventDisableTiming));
        }
        const size_t nchunks = (bytes + kChunk - 1) / kChunk;
        for (size_t c = 0; c <= nchunks; c++) {
            if (c < nchunks) {
                const size_t off = c * kChunk, len = std::min(kChunk, bytes - off);
                HIP_CHECK(hipMemcpyAsync(chunk[c & 1], static_cast<const char*>(dsrc) + off, len, hipMemcpyDeviceToHost, s_));
                HIP_CHECK(hipEventRecord(chunk_ev[c & 1], s_));
            }
            if (c > 0) {
                const size_t off = (c - 1) * kChunk, len = std::min(kChunk, bytes - off);
                HIP_CHECK(hipEventSynchronize(chunk_ev[(c - 1) & 1]));
                host_copy(static_cast<char*>(dst) + off, static_cast<const char*>(chunk[(c - 1) & 1]), len);
            }
        }
    }
    char* pinned(size_t bytes) {
        if (bytes > pin_bytes) {
            if (pin) (void)hipHostFree(pin);
            pin = nullptr;
            pin_bytes = 0;
            HIP_CHECK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
            pin_bytes = bytes;
        }
        return static_cast<char*>(pin);
    }
};

}  // namespace

struct raptor_model {
    DevModel h;  // host copy
};

struct raptor_problem {
    int kind = 0;  // 0: inverse dynamics only, 1: Kinova, 2: legged (Talos / G1)
    DevLegged hleg;
    DevBuf<DevLegged> d_leg;
    DevBuf<double> s_core;  // per-sample records of the legged core kernel (grown on demand, reused)
    DevBuf<double> d_glb, d_gub;  // the bounds on the device (raptor_violation_batch, uploaded on its first call)
    bool has_digit = false;
    DevBuf<DevDigit> d_digit;
    DevBuf<double> s_loops;  // per-sample closed-loop records (digit_kernel.h)
    int* cur_status = nullptr;      // set by raptor_eval_batch_status around one evaluation: per-z report of the loop solves
    int* cur_iterations = nullptr;
    DevBuf<int> s_status;           // staging of the *_status_host entry point: status [B] then iterations [B]
    DevBuf<int> d_kinds;
    DevBuf<DevCustRow> d_cust;       // customized block: row specs by row, per-sample row lists
    DevBuf<int> d_cust_list, d_cust_start;
    int cust_max_per_sample = 0;  // most customized rows any one sample has (the fused core kernel holds a sample's rows in its lane group)
    bool has_period = false;         // reset-map tables present (period_kernel.h)
    bool period_rows = false;        // ... and the block is part of this problem's rows
    DevPeriod hper;
    DevBuf<DevPeriod> d_per;
    DevBuf<double> s_ploops;         // per-z closed-loop records of the reset-map block (Digit)
    DevModel hmodel;
    DevTraj htraj;
    DevSpheres hspheres;
    DevLayout layout;
    int num_blocks = 0;
    int block_offsets[8] = {0};
    std::vector<double> tspan, g_lb, g_ub;
    // device-resident constants
    DevBuf<DevModel> d_model;
    DevBuf<DevTraj> d_traj;
    DevBuf<double> d_basis, d_wsum;
    DevBuf<DevSpheres> d_spheres;
    DevBuf<DevBox> d_boxes;
    DevBuf<double> s_hpart; // eval_h, split form: partial Hessians [32][7][28]
    DevBuf<DevHessLinks> d_hlinks;  // eval_h torque rows: per-link constants in axis-last frames (kind 1, 7-joint serial revolute chains)
    DevBuf<DevHessItems> d_hitems;  // eval_h over PiecewiseBezierCurves: the 154 block items dealt to four wavefronts
    DevBuf<double> s_phi;           // ... and the per-sample blocks Phi[chunk][182][N]
    DevBuf<int2> d_hpairs;          // ... and per (c1 <= c2) the columns and the sample range over which both weights are non-zero
    std::vector<double> hbasis;     // host copy of the column-weight tables (PiecewiseBezierCurves): constant cost Hessians
    DevBuf<double> d_qdes;  // KinovaOptimizer::q_des, tplan_n (raptor_kinova_set_objective); tplan_n < 0: not set
    int tplan_n = -1;
    // staging for the *_host entry points (grown on demand, reused)
    DevBuf<double> s_Z, s_G, s_J, s_a, s_b, s_c;
    HostStage hs;
    // optional per-kernel HIP-event timing (raptor_profile_begin / _end)
    bool profiling = false;
    std::vector<hipEvent_t> ev;   // 3 per evaluation: before / after the evaluation kernel (third: reserved)
    size_t ev_used = 0;
    // small-batch fork (launch_legged): a side stream for the reset-map kernels, events for fork / join (created on first use)
    hipStream_t side = nullptr;
    hipEvent_t ev_loops = nullptr, ev_core = nullptr, ev_side = nullptr;
    // Ordering between the asynchronous entry points and the *_host ones (which run on hs.stream()): every asynchronous call records
    // this event on the caller's stream when it has enqueued its work, every *_host call makes its private stream wait for it before
    // it touches the problem's scratch (s_core, s_loops, s_hpart, s_Z / s_G / s_J).  Created with the problem (no allocation later).
    hipEvent_t ev_async = nullptr;
    bool async_pending = false;
    ~raptor_problem() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : {ev_loops, ev_core, ev_side, ev_async})
            if (e) (void)hipEventDestroy(e);
        if (side) (void)hipStreamDestroy(side);
    }
};

namespace {

void fill_model(const raptor_model_desc* d, DevModel& m) {
    if (d == nullptr) throw std::invalid_argument("model: null descriptor");
    if (d->nv <= 0) throw std::invalid_argument("model: nv must be positive");
    if (d->nv > RAPTOR_MAX_NV) throw std::invalid_argument("model: nv exceeds RAPTOR_MAX_NV (36)");
    if (!d->parent || !d->jtype || !d->placement_R || !d->placement_p || !d->mass || !d->lever || !d->inertia)
        throw std::invalid_argument("model: null array");
    std::memset(&m, 0, sizeof(m));
    m.nv = d->nv;
    m.serial_chain = 1;
    for (int i = 0; i < d->nv; i++) {
        if (d->parent[i] >= i || d->parent[i] < -1) throw std::invalid_argument("model: parents must precede children");
        if (d->jtype[i] == 0 || std::abs(d->jtype[i]) > 6) throw std::invalid_argument("model: jtype must be +-1..6");
        if (d->jtype[i] == -6) throw unsupported_error("model: jtype -6 is mishandled by the reference itself (Spatial.h:100-104); refusing");
        m.parent[i] = d->parent[i];
        if (d->parent[i] != i - 1) m.serial_chain = 0;
        m.jtype[i] = d->jtype[i];
        for (int k = 0; k < 9; k++) m.plR[i][k] = d->placement_R[i * 9 + k];
        for (int k = 0; k < 3; k++) {
            m.plp[i][k] = d->placement_p[i * 3 + k];
            m.lever[i][k] = d->lever[i * 3 + k];
        }
        m.mass[i] = d->mass[i];
        const double* I = d->inertia + i * 9;
        m.Ic[i][0] = I[0]; m.Ic[i][1] = I[1]; m.Ic[i][2] = I[2];
        m.Ic[i][3] = I[4]; m.Ic[i][4] = I[5]; m.Ic[i][5] = I[8];
        m.armature[i] = d->armature ? d->armature[i] : 0.0;
        m.damping[i] = d->damping ? d->damping[i] : 0.0;
        m.friction[i] = d->friction ? d->friction[i] : 0.0;
    }
    for (int k = 0; k < 3; k++) m.neg_gravity[k] = -d->gravity[k];  // CustomizedInverseDynamics.cpp:212
}

// PiecewiseBezierCurves (Trajectories/src/PiecewiseBezierCurves.cpp:5-255): degree + 1 quintic pieces through `degree` interior
// control points (q, q_d, q_dd each), zero velocity / acceleration at both ends.  The trajectory is linear in z and column c
// moves joint c % Nact only, so the kernels get dense per-sample column weights W[i][order][c] (the entries of pq_pz, pq_d_pz,
// pq_dd_pz, :184-243) followed by the constant part fx[i][order][j] (q0 / qT when they are not optimised).
void fill_traj_piecewise(const raptor_traj_desc* d, DevTraj& t, std::vector<double>& tspan, std::vector<double>& basis,
                         std::vector<double>& wsum) {
    const int N = d->N, Nact = d->Nact, degree = d->degree;
    const double T = d->T;
    if (degree < 1) throw std::invalid_argument("traj: degree out of range");
    if (N % (degree + 1) != 0) throw std::invalid_argument("PiecewiseBezierCurves: N must be a multiple of (degree + 1)");  // :18-20
    t.degree = degree;
    int V = Nact * degree * 3;  // :34-45
    int begin_off = -1, end_off = -1;
    if (!d->q0) { begin_off = V; V += Nact; }
    if (!d->qT) { end_off = V; V += Nact; }
    t.varLength = V;
    t.K = V / Nact;
    tspan.assign(N, 0.0);
    for (int i = 0; i < N; i++) tspan[i] = (N == 1) ? T : (0.0 + i * ((T - 0.0) / (N - 1)));  // LinSpaced(N, 0, T), :22
    if (N > 1) tspan[N - 1] = T;
    wsum.assign((size_t)N * 3, 0.0);
    basis.assign((size_t)N * 3 * V + (size_t)N * 3 * Nact, 0.0);
    const int seg_size = N / (degree + 1);
    const double localT = T / (degree + 1);
    const double vf = localT / 5.0, af = localT * localT / 20.0;  // :74-75
    const double binom[6] = {1, 5, 10, 10, 5, 1};
    for (int x = 0; x < N; x++) {
        const int id = x / seg_size;
        const double tt = (tspan[x] - id * localT) / localT;
        double tA[6], tB[6], dtA[6], dtB[6], ddtA[6], ddtB[6], Bs[3][6];
        for (int k = 0; k < 6; k++) { tA[k] = 1; tB[k] = 1; dtA[k] = 0; dtB[k] = 0; ddtA[k] = 0; ddtB[k] = 0; }
        for (int j = 1; j <= 5; j++) {  // :151-160
            tA[j] = tt * tA[j - 1];
            tB[5 - j] = (1 - tt) * tB[5 - j + 1];
            dtA[j] = j * tA[j - 1];
            dtB[5 - j] = -j * tB[5 - j + 1];
            ddtA[j] = j * dtA[j - 1];
            ddtB[5 - j] = -j * dtB[5 - j + 1];
        }
        for (int k = 0; k < 6; k++) {  // :162-167
            Bs[0][k] = binom[k] * tA[k] * tB[k];
            Bs[1][k] = binom[k] * (dtA[k] * tB[k] + tA[k] * dtB[k]) / localT;
            Bs[2][k] = binom[k] * (ddtA[k] * tB[k] + 2 * dtA[k] * dtB[k] + tA[k] * ddtB[k]) / (localT * localT);
        }
        for (int o = 0; o < 3; o++) {
            const double* b = Bs[o];
            double* W = &basis[(size_t)(x * 3 + o) * V];
            double* fx = &basis[(size_t)N * 3 * V + (size_t)(x * 3 + o) * Nact];
            for (int i = 0; i < Nact; i++) {
                if (id < degree) {  // the control point that ends this piece
                    const int c0 = id * 3 * Nact + i;
                    W[c0] = b[3] + b[4] + b[5];
                    W[c0 + Nact] = -b[3] * 2.0 * vf - b[4] * vf;
                    W[c0 + 2 * Nact] = b[3] * af;
                } else if (end_off >= 0) {
                    W[end_off + i] = b[3] + b[4] + b[5];
                } else {
                    fx[i] += d->qT[i] * (b[3] + b[4] + b[5]);
                }
                if (id > 0) {  // the control point that begins this piece
                    const int c0 = (id - 1) * 3 * Nact + i;
                    W[c0] = b[0] + b[1] + b[2];
                    W[c0 + Nact] = b[1] * vf + b[2] * 2.0 * vf;
                    W[c0 + 2 * Nact] = b[2] * af;
                } else if (begin_off >= 0) {
                    W[begin_off + i] = b[0] + b[1] + b[2];
                } else {
                    fx[i] += d->q0[i] * (b[0] + b[1] + b[2]);
                }
            }
        }
    }
}

// Trajectory constructor + per-sample basis tables (Trajectories.cpp:45-69, BezierCurves.cpp:21-37,117-141,
// Polynomials.cpp:54-68, ArmourBezierCurves.cpp:63-69).  Pure set-up: runs once per problem.
void fill_traj(const raptor_traj_desc* d, DevTraj& t, std::vector<double>& tspan, std::vector<double>& basis,
               std::vector<double>& wsum) {
    if (d == nullptr) throw std::invalid_argument("traj: null descriptor");
    std::memset(&t, 0, sizeof(t));
    if (d->N < 1 || d->Nact < 1) throw std::invalid_argument("traj: N and Nact must be positive");
    if (d->Nact > RAPTOR_MAX_NV) throw std::invalid_argument("traj: Nact too large");
    t.type = d->type;
    t.N = d->N;
    t.Nact = d->Nact;
    const double T = d->T;
    const int N = d->N;
    if (d->type == RAPTOR_TRAJ_ARMOUR_BEZIER) {
        t.degree = 5;  // ARMOUR_BEZIER_CURVE_DEGREE
        t.varLength = d->Nact;
        if (!d->q0) throw std::invalid_argument("ArmourBezierCurves: q0.size() != Nact");
        if (!d->q_d0) throw std::invalid_argument("ArmourBezierCurves: q_d0.size() != Nact");
        if (!d->q_dd0) throw std::invalid_argument("ArmourBezierCurves: q_dd0.size() != Nact");
        for (int j = 0; j < d->Nact; j++) {
            t.armour_q0[j] = d->q0[j];
            t.armour_c[0][j] = d->q0[j];
            t.armour_c[1][j] = d->q0[j] + (T * d->q_d0[j]) / 5.0;
            t.armour_c[2][j] = d->q0[j] + (2.0 * T * d->q_d0[j]) / 5.0 + (T * T * d->q_dd0[j]) / 20.0;
        }
    } else if (d->type == RAPTOR_TRAJ_BEZIER || d->type == RAPTOR_TRAJ_POLYNOMIAL) {
        if (d->degree < 1 || d->degree > RAPTOR_MAX_DEGREE) throw std::invalid_argument("traj: degree out of range");
        t.degree = d->degree;
        t.varLength = (d->degree + 1) * d->Nact;
        if (d->type == RAPTOR_TRAJ_BEZIER) {  // BezierCurves::constrainInitialPosition / Velocity (BezierCurves.cpp:39-69)
            if (d->q_d0 && !d->q0) throw std::invalid_argument("BezierCurves: initial position must be constrained first");
            t.nfixed = d->q0 ? (d->q_d0 ? 2 : 1) : 0;
            t.varLength -= t.nfixed * d->Nact;
            if (d->degree + 1 - t.nfixed < 1) throw std::invalid_argument("traj: degree too low for the constrained initial state");
        }
    } else if (d->type == RAPTOR_TRAJ_PIECEWISE_BEZIER) {
        fill_traj_piecewise(d, t, tspan, basis, wsum);
        return;
    } else {
        throw std::invalid_argument("traj: unknown trajectory type");
    }
    t.K = t.degree + 1 - t.nfixed;
    tspan.assign(N, 0.0);
    if (d->time_discretization == RAPTOR_TIME_UNIFORM) {
        for (int i = 0; i < N; i++) tspan[i] = (N == 1) ? T : (0.0 + i * ((T - 0.0) / (N - 1)));
        if (N > 1) tspan[N - 1] = T;
    } else if (d->time_discretization == RAPTOR_TIME_CHEBYSHEV) {
        for (int i = 1; i < N - 1; i++) tspan[i] = 0.5 * T * (1 + std::cos(M_PI * (2 * (N - i) - 3) / (2 * (N - 2))));
        tspan[0] = 0;
        tspan[N - 1] = T;
    } else {
        throw std::invalid_argument("traj: unknown time discretization");
    }
    const int deg = t.degree, K = deg + 1;
    basis.assign((size_t)N * 3 * K, 0.0);
    wsum.assign((size_t)N * 3, 0.0);
    std::vector<double> binom(K, 1.0);
    for (int j = 1; j <= deg / 2; j++) {
        binom[j] = binom[j - 1] * (deg + 1 - j) / j;
        binom[deg - j] = binom[j];
    }
    for (int x = 0; x < N; x++) {
        double* B = &basis[(size_t)(x * 3) * K];
        double* dB = B + K;
        double* ddB = dB + K;
        if (d->type == RAPTOR_TRAJ_POLYNOMIAL) {
            const double tt = tspan[x];
            B[0] = 1; dB[0] = 0; ddB[0] = 0;
            B[1] = tt; dB[1] = 1; ddB[1] = 0;
            double powt = 1;
            for (int i = 2; i <= deg; i++) {
                ddB[i] = powt;
                dB[i] = tt * ddB[i] / (i - 1);
                B[i] = tt * dB[i] / i;
                powt *= tt;
            }
        } else {
            const double tt = tspan[x] / T;
            std::vector<double> tA(K, 1.0), tB(K, 1.0), dtA(K, 0.0), dtB(K, 0.0), ddtA(K, 0.0), ddtB(K, 0.0);
            for (int j = 1; j <= deg; j++) {
                tA[j] = tt * tA[j - 1];
                tB[deg - j] = (1 - tt) * tB[deg - j + 1];
                dtA[j] = j * tA[j - 1];
                dtB[deg - j] = -j * tB[deg - j + 1];
                ddtA[j] = j * dtA[j - 1];
                ddtB[deg - j] = -j * dtB[deg - j + 1];
            }
            for (int k = 0; k < K; k++) {
                B[k] = binom[k] * tA[k] * tB[k];
                dB[k] = binom[k] * (dtA[k] * tB[k] + tA[k] * dtB[k]) / T;
                ddB[k] = binom[k] * (ddtA[k] * tB[k] + 2 * dtA[k] * dtB[k] + tA[k] * ddtB[k]) / (T * T);
            }
        }
        if (d->type == RAPTOR_TRAJ_ARMOUR_BEZIER) {  // ArmourBezierCurves.cpp:148-150
            wsum[x * 3] = B[3] + B[4] + B[5];
            wsum[x * 3 + 1] = dB[3] + dB[4] + dB[5];
            wsum[x * 3 + 2] = ddB[3] + ddB[4] + ddB[5];
        }
    }
    if (t.nfixed > 0) {
        // constrained initial state: the kernels see basis tables of the FREE coefficients only (width t.K), followed by the
        // pinned rows' contribution per (sample, derivative, joint): q0 B_0 + (q0 + q_d0 T / 5.0) B_1   (BezierCurves.cpp:84-96,
        // 5.0 as written there, whatever the degree)
        const int Kz = t.K, off = t.nfixed, Nact = d->Nact;
        std::vector<double> packed((size_t)N * 3 * Kz + (size_t)N * 3 * Nact, 0.0);
        for (int x = 0; x < N; x++)
            for (int dd = 0; dd < 3; dd++) {
                const double* Bd = &basis[(size_t)(x * 3 + dd) * K];
                for (int k = 0; k < Kz; k++) packed[(size_t)(x * 3 + dd) * Kz + k] = Bd[off + k];
                double* fx = &packed[(size_t)N * 3 * Kz + (size_t)(x * 3 + dd) * Nact];
                for (int j = 0; j < Nact; j++) {
                    double v = d->q0[j] * Bd[0];
                    if (off == 2) v += (d->q0[j] + d->q_d0[j] * T / 5.0) * Bd[1];
                    fx[j] = v;
                }
            }
        basis.swap(packed);
    }
}

void upload_common(raptor_problem& p, const std::vector<double>& basis, const std::vector<double>& wsum) {
    p.d_model.upload(&p.hmodel, 1);
    p.d_traj.upload(&p.htraj, 1);
    p.d_basis.upload(basis.data(), basis.size());
    p.d_wsum.upload(wsum.data(), wsum.size());
}

double* scratch_device_address();  // (defined with the allocation-free eval_h entry point)
// last step of every *_problem_create: what later calls must not have to allocate, upload or look up
void finish_problem(raptor_problem& p) {
    if (p.kind == 1) (void)scratch_device_address();  // raptor_eval_h_batch_noalloc: this device's workspace address, resolved here
    if (p.kind == 1 && p.hmodel.nv == 7 && p.hmodel.serial_chain) {  // eval_h: the chain's constants in axis-last link frames
        bool revolute = true;
        for (int l = 0; l < 7; l++) revolute = revolute && std::abs(p.hmodel.jtype[l]) >= 1 && std::abs(p.hmodel.jtype[l]) <= 3;
        if (revolute) {
            DevHessLinks h{};
            int pc[3] = {0, 1, 2};  // the parent frame's re-labelling: new axis i = old axis pc[i]
            for (int l = 0; l < 7; l++) {
                const DevModel& m = p.hmodel;
                const int K = std::abs(m.jtype[l]) - 1, I = (K + 1) % 3, J = (K + 2) % 3;
                const int me[3] = {I, J, K};
                double* c = h.c[l];
                for (int r = 0; r < 3; r++)
                    for (int cc = 0; cc < 3; cc++) c[r * 3 + cc] = m.plR[l][pc[cc] * 3 + me[r]];  // (R^T)[me[r]][pc[cc]]
                for (int cc = 0; cc < 3; cc++) c[9 + cc] = m.plp[l][pc[cc]];
                c[12] = m.mass[l];
                for (int r = 0; r < 3; r++) c[13 + r] = m.lever[l][me[r]];
                auto sym = [&](int i, int j) { const int a = std::min(i, j), b = std::max(i, j); return m.Ic[l][(a * (5 - a)) / 2 + b]; };
                c[16] = sym(I, I); c[17] = sym(I, J); c[18] = sym(I, K); c[19] = sym(J, J); c[20] = sym(J, K); c[21] = sym(K, K);
                c[22] = m.jtype[l] < 0 ? -1.0 : 1.0;
                for (int r = 0; r < 3; r++) pc[r] = me[r];
            }
            p.d_hlinks.upload(&h, 1);
        }
    }
    if (!p.g_lb.empty()) {  // raptor_violation_batch reads the bounds on the device; they never change after create
        p.d_glb.upload(p.g_lb.data(), p.g_lb.size());
        p.d_gub.upload(p.g_ub.data(), p.g_ub.size());
    }
    HIP_CHECK(hipEventCreateWithFlags(&p.ev_async, hipEventDisableTiming));
}
// an asynchronous entry point has enqueued work of `p` on the caller's stream `st`
inline void mark_async(raptor_problem& p, hipStream_t st) {
    if (p.ev_async == nullptr || st == p.hs.st) return;
    HIP_CHECK(hipEventRecord(p.ev_async, st));
    p.async_pending = true;
}
// a *_host entry point is about to use the problem's scratch on stream `st` (its private stream, or the null stream)
inline void order_after_async(raptor_problem& p, hipStream_t st) {
    if (p.async_pending) HIP_CHECK(hipStreamWaitEvent(st, p.ev_async, 0));
}

inline unsigned grid_for(long long total) { return (unsigned)((total + RAPTOR_BLOCK - 1) / RAPTOR_BLOCK); }

void check_launchable(const raptor_problem& p, int B) {
    if (B < 0) throw std::invalid_argument("B < 0");
    const long long lanes = (long long)B * p.htraj.N * std::max(p.layout.n, p.hspheres.num_spheres);
    if (lanes / RAPTOR_BLOCK > 0x7fffffffLL) throw std::invalid_argument("batch too large for one launch");
}

template <bool DERIV>
void launch_id(raptor_problem& p, const DevLayout& L, const double* dZ, int B, double* dG, double* dJ, hipStream_t st) {
    const int lanes_per_e = DERIV ? L.n : 1;
    const long long total = (long long)B * p.htraj.N * lanes_per_e;
    if (total == 0) return;
    if (!p.hmodel.serial_chain) throw unsupported_error("inverse dynamics kernel: only serial-chain models are built so far");
    // one LDS slot per sample-eval touched by a block; the values-only kernel (one lane per e) uses 64-lane blocks
    const int block = DERIV ? RAPTOR_BLOCK : 64;
    const int slots = block / lanes_per_e + 2;
    const size_t lds_bytes = (size_t)slots * p.hmodel.nv * RAPTOR_ID_SLOT * sizeof(double);
    const unsigned grid = (unsigned)((total + block - 1) / block);
    switch (p.hmodel.nv) {
        case 7:
            hipLaunchKernelGGL((id_kernel<7, DERIV>), dim3(grid), dim3(block), lds_bytes, st, p.d_model.p,
                               p.d_traj.p, p.d_basis.p, p.d_wsum.p, L, dZ, total, dG, dJ);
            break;
        default:
            throw unsupported_error("inverse dynamics kernel: nv = " + std::to_string(p.hmodel.nv) + " not instantiated yet");
    }
    HIP_CHECK(hipGetLastError());
}

template <bool DERIV, bool RECORD = false>
void launch_kinova(raptor_problem& p, const double* dZ, int B, double* dG, double* dJ, hipStream_t st) {
    const long long E = (long long)B * p.htraj.N;
    if (E == 0) return;
    const size_t lds_bytes = (size_t)kinova_lds_doubles(p.hspheres.num_spheres) * sizeof(double);
    static bool attr_set[2] = {false, false};  // one flag pair per instantiation of this template
    if (!attr_set[DERIV]) {  // > 64 KiB of dynamic LDS needs the opt-in
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&kinova_fused_kernel<DERIV, RECORD>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(kinova_lds_doubles(RAPTOR_MAX_SPHERES) * sizeof(double))));
        attr_set[DERIV] = true;
    }
    const unsigned grid = (unsigned)((E + K_EPB - 1) / K_EPB);
    hipLaunchKernelGGL((kinova_fused_kernel<DERIV, RECORD>), dim3(grid), dim3(KIN_BLOCK), lds_bytes, st, p.d_model.p,
                       p.d_traj.p, p.d_basis.p, p.d_wsum.p, p.d_spheres.p, p.d_boxes.p, p.layout, dZ, E, dG, dJ);
    HIP_CHECK(hipGetLastError());
}

// eval_h for the longer-horizon Kinova problem (kinova_hess_blocks_kernel.h)
void hess_blocks_device(raptor_problem& p, const double* dZ, int B, const double* dLam, double* dH, hipStream_t st) {
    if (!p.hmodel.serial_chain) throw unsupported_error("eval_h kernel: serial-chain models only");
    for (int l = 0; l < p.hmodel.nv; l++)
        if (std::abs(p.hmodel.jtype[l]) > 3) throw unsupported_error("eval_h kernel: revolute joints only");
    if (!p.d_hitems.p) {  // deal the 154 block entries to the four wavefronts, longest first
        struct Item { int j1, j2, k1, k2, out, cost; };
        std::vector<Item> all;
        auto add = [&](int ja, int ka, int jb, int kb, int out) {  // (ja, ka) and (jb, kb) in any order -> j1 <= j2
            Item it = (ja <= jb) ? Item{ja, jb, ka, kb, out, 0} : Item{jb, ja, kb, ka, out, 0};
            it.cost = (7 - it.j2) * 1000 + (it.j2 - it.j1) * 250 + it.j1 * 120 + 300;
            if (it.k1 + it.k2 == 1) it.cost = 300 + 7 * 200 + (7 - (it.k1 == 0 ? it.j1 : it.j2)) * 500 + (7 - it.j2) * 400;  // written-out q_d variation, D1 from the q seed on
            if (it.k1 == 1 && it.k2 == 1) it.cost = 300 + 7 * 200 + (7 - it.j2) * 200;  // bilinear pass in plain doubles
            if ((it.k1 == 2) != (it.k2 == 2)) it.cost = 300 + 100 * 7 + (7 - (it.k1 == 0 ? it.j1 : it.j2)) * 250 + (7 - it.j2) * 150;  // reduced pass: D1 from the q seed on  // reduced first-order pass over the whole chain
            all.push_back(it);
        };
        for (int a = 0; a < 7; a++)
            for (int b = a; b < 7; b++) {
                const int sym = a * 7 - (a * (a - 1)) / 2 + (b - a);
                add(a, 0, b, 0, HB_QQ + sym);
                add(a, 1, b, 1, HB_VV + sym);
            }
        for (int a = 0; a < 7; a++)
            for (int b = 0; b < 7; b++) {
                add(a, 0, b, 1, HB_QV + a * 7 + b);  // d2 / dq_a dv_b
                add(a, 0, b, 2, HB_QA + a * 7 + b);  // d2 / dq_a da_b
            }
        std::sort(all.begin(), all.end(), [](const Item& x, const Item& y) { return x.cost > y.cost; });
        // table 0: the whole item set on one workgroup's four wavefronts (batched form); tables 1 .. HB_SPLIT: the item set over
        // HB_SPLIT - 1 workgroups, the last table empty (that workgroup owns the obstacle entries) -- small batches
        std::vector<DevHessItems> h(1 + HB_SPLIT);
        std::memset(h.data(), 0, h.size() * sizeof(DevHessItems));
        auto deal = [&](DevHessItems* tab, int ntab) {
            std::vector<long long> load((size_t)ntab * H_WAVES, 0);
            for (const Item& it : all) {
                size_t w = 0;
                for (size_t k = 1; k < load.size(); k++)
                    if (load[k] < load[w]) w = k;
                DevHessItems& t = tab[w / H_WAVES];
                const int wv = (int)(w % H_WAVES);
                if (t.count[wv] >= HB_MAX_ITEMS) throw std::runtime_error("eval_h: item table overflow");
                t.code[wv][t.count[wv]++] = it.j1 | (it.j2 << 3) | (it.k1 << 6) | (it.k2 << 8) | (it.out << 10);
                load[w] += it.cost;
            }
        };
        deal(h.data(), 1);
        deal(h.data() + 1, HB_SPLIT - 1);
        p.d_hitems.upload(h.data(), h.size());
    }
    const int N = p.htraj.N, V = p.layout.n;
    if (!p.d_hpairs.p) {
        std::vector<int2> rg(V);
        for (int c = 0; c < V; c++) {
            int lo = N, hi = 0;
            for (int i = 0; i < N; i++)
                for (int o = 0; o < 3; o++)
                    if (p.hbasis[(size_t)(i * 3 + o) * V + c] != 0.0) {
                        lo = std::min(lo, i);
                        hi = std::max(hi, i + 1);
                    }
            rg[c] = make_int2(lo, hi);
        }
        if (V > 0xffff || N > 0x7fff) throw unsupported_error("eval_h: problem too large for the pair table");
        std::vector<int2> pr;
        pr.reserve((size_t)V * (V + 1) / 2);
        for (int c1 = 0; c1 < V; c1++)
            for (int c2 = c1; c2 < V; c2++) {
                const int lo = std::max(rg[c1].x, rg[c2].x), hi = std::max(lo, std::min(rg[c1].y, rg[c2].y));
                pr.push_back(make_int2(c1 | (c2 << 16), lo | (hi << 16)));
            }
        p.d_hpairs.upload(pr.data(), pr.size());
    }
    const size_t per_z = (size_t)HB_NI * N;
    const int CH = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, ((size_t)1 << 26) / per_z));  // <= 512 MB of blocks in flight
    p.s_phi.ensure((size_t)CH * per_z);
    const size_t nh = (size_t)V * (V + 1) / 2;
    int SCH = 8;  // samples per LDS chunk of the contraction (power of two)
    if (const char* ev = getenv("RAPTOR_LH_HCH")) SCH = std::max(1, std::min(64, atoi(ev)));
    while (SCH & (SCH - 1)) SCH &= SCH - 1;
    auto lds_of = [&](int ch) { return ((size_t)HB_NI * (ch + 1) + (size_t)ch * 3 * V) * sizeof(double); };
    while (SCH > 1 && lds_of(SCH) > 64 * 1024) SCH >>= 1;
    if (lds_of(SCH) > 64 * 1024) throw unsupported_error("eval_h: too many decision variables");
    const size_t lds_bytes = lds_of(SCH);
    for (int b0 = 0; b0 < B; b0 += CH) {
        const int nb = std::min(CH, B - b0);
        const int passes = (N + H_CH - 1) / H_CH;
        static const long long split_max = getenv("RAPTOR_LH_SPLIT_MAX") ? atoll(getenv("RAPTOR_LH_SPLIT_MAX")) : 300;
        const bool split = (long long)nb * passes <= split_max;  // too few workgroups to fill the chip: spread items and passes
        hipLaunchKernelGGL(kinova_hess_blocks_kernel, split ? dim3((unsigned)nb, HB_SPLIT, (unsigned)passes) : dim3((unsigned)nb),
                           dim3(RAPTOR_BLOCK), 0, st, p.d_model.p, p.d_traj.p, p.d_basis.p, p.d_wsum.p, p.d_spheres.p, p.d_boxes.p,
                           p.layout, p.d_hitems.p + (split ? 1 : 0), dZ + (size_t)b0 * V, dLam + (size_t)b0 * p.layout.m, p.s_phi.p);
        const unsigned gy = split ? (unsigned)std::min<size_t>(16, (nh + RAPTOR_BLOCK - 1) / RAPTOR_BLOCK) : 1u;
        hipLaunchKernelGGL(kinova_hess_contract_kernel, dim3((unsigned)nb, gy), dim3(RAPTOR_BLOCK), lds_bytes, st, p.d_traj.p, p.d_basis.p,
                           p.d_hpairs.p, (int)nh, SCH, p.s_phi.p, dH + (size_t)b0 * nh);
        HIP_CHECK(hipGetLastError());
    }
}

// KinovaCustomizedConstraints rows for any trajectory whose columns move one joint each (sphere_rows_kernel.h)
void launch_sphere_rows(raptor_problem& p, const double* dZ, int B, double* dG, double* dJ, hipStream_t st) {
    const int ns = p.hspheres.num_spheres;
    const long long E = (long long)B * p.htraj.N;
    if (E == 0 || ns == 0) return;
    const unsigned grid = (unsigned)((E + SR_EPB - 1) / SR_EPB);
    if (dJ) hipLaunchKernelGGL((sphere_rows_kernel<true>), dim3(grid), dim3(RAPTOR_BLOCK), 0, st, p.d_model.p, p.d_traj.p, p.d_basis.p,
                               p.d_wsum.p, p.d_spheres.p, p.d_boxes.p, p.layout, dZ, E, dG, dJ);
    else hipLaunchKernelGGL((sphere_rows_kernel<false>), dim3(grid), dim3(RAPTOR_BLOCK), 0, st, p.d_model.p, p.d_traj.p, p.d_basis.p,
                            p.d_wsum.p, p.d_spheres.p, p.d_boxes.p, p.layout, dZ, E, dG, dJ);
    HIP_CHECK(hipGetLastError());
}

// core kernel (values + compact derivative records), then the streaming expansion into pg_pz
// fused form (round 3): the wide core kernel also writes the torque / joint-limit / contact rows of pg_pz (phase G) -- only the
// customized block is left for legged_expand_kernel
bool legged_can_fuse(const raptor_problem& p, const double* dJ, const long long E) {
    const bool no_fuse = std::getenv("RAPTOR_NO_FUSE") != nullptr;  // read per call: the tests and A/B scripts switch between the two forms
    // Small batches keep the two-kernel form: a handful of wavefronts expands its rows one after the other, the separate expansion
    // spreads them over the idle GPU (host-synchronised call, fused / two-kernel: Digit B = 1 174 / 164 us, 128 207 / 198, 256 230 / 236,
    // 1024 533 / 626; G1 B = 1 90 / 78, 128 102 / 97, 256 110 / 113, 1024 260 / 310 -- scripts/fuse_small_batch_sweep.py).
    // RAPTOR_FUSE_MIN=<sample-evals> moves the threshold (read per call: the parity test of the fused rows runs at B = 1 .. 5).
    const char* fm = std::getenv("RAPTOR_FUSE_MIN");
    if (E < (fm ? std::atoll(fm) : 3072)) return false;
    const int ndyn = p.hleg.ni * p.hleg.K;
    const int LG = p.hleg.nb > 16 ? 32 : 16;
    if (no_fuse || p.htraj.N % 2 != 0 || p.layout.n % 2 != 0 || ((size_t)dJ & 15) != 0) return false;
    if (ndyn > 2 * LEG_BLOCK || p.layout.n > 4 * LEG_BLOCK || p.layout.n < ndyn || p.cust_max_per_sample > LG) return false;
    // phase G finds the D blocks in the vector slots of the actuators' joints (3 ni <= 36 doubles) and parks twelve P rows in a
    // sample-eval's scratch and in the slots of the lanes without a body, three rows per slot
    const int ni = p.hleg.ni, nts = (LG * 4) / std::max(ni, 1);
    return ni <= 12 && nts >= 3 && nts + 3 * (LG - p.hleg.nb) >= 12;
}

// wavefronts of the fused core kernel resident at once on the current device (8 one-wavefront workgroups per CU: LDS and registers)
int fused_first_round() {
    static int per_dev[64] = {0};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return 2048;
    if (per_dev[dev] == 0) {
        int cus = 0;
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        per_dev[dev] = 8 * std::max(cus, 1);
    }
    return per_dev[dev];
}

void launch_legged_core(raptor_problem& p, const double* dZ, int B, double* dG, bool deriv, hipStream_t st, double* d_lamrec = nullptr,
                        hipEvent_t after_loops = nullptr, double* dJ_fused = nullptr) {
    const long long E = (long long)B * p.htraj.N;
    if (E == 0) return;
    p.s_core.ensure((size_t)E * p.hleg.c_stride);
    const double* d_loops = nullptr;
    if (p.has_digit) {  // closed loops first: one lane per (sample-eval, leg), knee and toe loops in separate launches
        p.s_loops.ensure((size_t)E * p.hleg.nrec * DIGIT_REC);
        const unsigned lgrid = 2 * (unsigned)((E + 63) / 64);  // a wavefront per (64 sample-evals, leg)
        // knee and toe loops side by side in one launch (measured: 2.426 -> 2.398 ms per Digit step, 211 -> 200 us for one problem)
        hipLaunchKernelGGL(digit_loops_both_kernel, dim3(2 * lgrid), dim3(64), 0, st, p.d_traj.p, p.d_digit.p, p.d_basis.p,
                           p.layout.n, dZ, E, (int)lgrid, p.s_loops.p, p.cur_status, p.cur_iterations);
        HIP_CHECK(hipGetLastError());
        d_loops = p.s_loops.p;
    }
    if (after_loops) HIP_CHECK(hipEventRecord(after_loops, st));
    const bool wide = p.hleg.nb > 16;
    const int LG = wide ? 32 : 16;
    const int epb = LEG_BLOCK / LG;
    const size_t lds_bytes = (size_t)epb * (wide ? LegLds<32>::STRIDE : LegLds<16>::STRIDE) * sizeof(double);
    const unsigned grid = (unsigned)((E + epb - 1) / epb);
    static const size_t lds_extra = std::getenv("RAPTOR_LEG_LDS_EXTRA") ? (size_t)std::atoll(std::getenv("RAPTOR_LEG_LDS_EXTRA")) : 0;  // experiments: fewer wavefronts per CU
    // Talos / G1 (LG = 16), fused form, launches of six rounds of wavefronts or more: SEVEN workgroups per CU instead of eight (the
    // extra LDS request does it).  Measured, G1, 8 / 7 / 6 per CU: 8192 problems 1.345 / 1.220 / 1.405 ms, 4096 problems 0.686 / 0.649 /
    // 0.737, 2048 problems 0.367 / 0.365 / 0.404, 1024 problems 0.202 / 0.220 / 0.238 -- the staggered start changes nothing for this
    // instance, an odd wavefront count per CU (one SIMD runs a single wavefront, at its own pace) does.  Digit (LG = 32) loses 22 % at seven.
    const bool seven = !wide && (long long)grid >= 6LL * fused_first_round();
    const size_t lds_fused = lds_bytes + lds_extra + ((seven && lds_extra == 0) ? 2900 : 0);
#define LEG_LAUNCH(LGV, DV)                                                                                          \
    hipLaunchKernelGGL((legged_core_kernel<LGV, DV>), dim3(grid), dim3(LEG_BLOCK), lds_bytes, st, p.d_model.p,        \
                       p.d_traj.p, p.d_leg.p, p.d_basis.p, p.layout, dZ, E, d_loops, p.d_cust.p, p.d_cust_list.p,      \
                       p.d_cust_start.p, dG, p.s_core.p, d_lamrec)
#define LEG_LAUNCH_FUSED(LGV, KE)                                                                                            \
    hipLaunchKernelGGL((legged_core_kernel<LGV, true, true, KE>), dim3(grid), dim3(LEG_BLOCK), lds_fused, st, p.d_model.p,       \
                       p.d_traj.p, p.d_leg.p, p.d_basis.p, p.layout, dZ, E, d_loops, p.d_cust.p, p.d_cust_list.p,                 \
                       p.d_cust_start.p, dG, p.s_core.p, d_lamrec, dJ_fused, stagger, stagger_units)
    if (deriv && dJ_fused != nullptr) {
        const bool keven = p.hleg.K % 2 == 0;
        // the staggered start (slot k of a CU waits k x units x 3.5 us) pays once the launch is more than one round of wavefronts
        // (measured, units 0 / 1 / 2 / 3 at 1.75 rounds: G1 1024 problems 0.218 / 0.214 / 0.207 / 0.223 ms, Digit 1024 problems
        // 0.354 / 0.348 / 0.330 / 0.330 ms; at 14 rounds 2 and 4 are equal); a single problem must not wait at all
        static const int units_env = std::getenv("RAPTOR_STAGGER_UNITS") ? std::atoi(std::getenv("RAPTOR_STAGGER_UNITS")) : -1;  // experiments
        const int first = fused_first_round();
        const bool rounds = 2LL * grid >= 3LL * first;
        const int stagger = rounds ? first : 0;
        // per wave slot, per SIMD (RAPTOR_STAGGER_UNITS=u: 4u and u; RAPTOR_STAGGER_AB="a,b": a per wave slot, b per SIMD)
        static const char* ab_env = std::getenv("RAPTOR_STAGGER_AB");
        int ua = 4 * (units_env >= 0 ? units_env : LEG_STAGGER), ub = units_env >= 0 ? units_env : LEG_STAGGER;
        if (ab_env) std::sscanf(ab_env, "%d,%d", &ua, &ub);
        const int stagger_units = (ua << 8) | (ub & 255);
        if (wide) { if (keven) LEG_LAUNCH_FUSED(32, true); else LEG_LAUNCH_FUSED(32, false); }
        else { if (keven) LEG_LAUNCH_FUSED(16, true); else LEG_LAUNCH_FUSED(16, false); }
#undef LEG_LAUNCH_FUSED
    } else if (wide) {
        if (deriv) LEG_LAUNCH(32, true); else LEG_LAUNCH(32, false);
    } else {
        if (deriv) LEG_LAUNCH(16, true); else LEG_LAUNCH(16, false);
    }
#undef LEG_LAUNCH
    HIP_CHECK(hipGetLastError());
}

// reset-map block: once per decision vector, from the first and last samples' records of launch_legged_core.
// next / dZ2 / dJ2 non-null: the multiple-step link form (period_kernel.h)
void launch_period(raptor_problem& p, const double* dZ, int B, double* dG, double* dJ, bool jac, hipStream_t st, int mrows, int roff,
                   raptor_problem* next, const double* dZ2, double* dJ2, hipEvent_t core_done = nullptr) {
    // core_done (small-batch fork): `st` is the side stream; the loop part below needs the loops kernel only (the caller made `st`
    // wait for it), the reset-map kernel itself waits for the core kernel's records here
    const double* d_pl = nullptr;
    if (p.has_digit) {
        p.s_ploops.ensure((size_t)B * p.hper.loop_stride);
        const long long total = (long long)B * 2;
        const unsigned lgrid = (unsigned)((total + 63) / 64);
        hipLaunchKernelGGL(digit_period_loops_kernel, dim3(2 * lgrid), dim3(64), 0, st, p.d_traj.p, p.d_digit.p, p.d_per.p,
                           p.d_basis.p, p.layout.n, p.hleg.nv, p.hleg.nd, dZ, total, (int)lgrid, p.s_loops.p, p.s_ploops.p);
        d_pl = p.s_ploops.p;
    }
    if (core_done) HIP_CHECK(hipStreamWaitEvent(st, core_done, 0));
    const DevTraj* T2 = next ? next->d_traj.p : nullptr;
    const double* basis2 = next ? next->d_basis.p : nullptr;
    const bool small = p.hleg.nv <= 18 && p.hleg.nd <= 6;  // Talos / G1; Digit takes the 36 / 24 instance
#define PERIOD_LAUNCH(NVM_, NDM_, DV_, dJ_)                                                                              \
    hipLaunchKernelGGL((legged_period_kernel<NVM_, NDM_, DV_>), dim3((unsigned)B), dim3(PL_BLOCK),                          \
                       (size_t)period_lds_doubles(NVM_, NDM_) * sizeof(double), st, p.d_model.p, p.d_traj.p, p.d_leg.p,    \
                       p.d_per.p, p.d_basis.p, p.layout, dZ, p.s_core.p, p.s_loops.p, d_pl, mrows, roff, T2, basis2, dZ2, dG, dJ_, dJ2)
    if (small) {
        if (jac) PERIOD_LAUNCH(18, 6, true, dJ); else PERIOD_LAUNCH(18, 6, false, nullptr);
    } else {
        if (jac) PERIOD_LAUNCH(RAPTOR_MAX_NV, 24, true, dJ); else PERIOD_LAUNCH(RAPTOR_MAX_NV, 24, false, nullptr);
    }
#undef PERIOD_LAUNCH
    HIP_CHECK(hipGetLastError());
}

void launch_legged(raptor_problem& p, const double* dZ, int B, double* dG, double* dJ, bool jac, hipStream_t st,
                   hipEvent_t between, hipEvent_t split1 = nullptr) {
    const long long E = (long long)B * p.htraj.N;
    // Small batches (one Ipopt-driven problem): the launches are a chain of latencies on a nearly empty GPU, so the reset-map kernels
    // go to a side stream -- their loop part beside the core kernel, the block itself beside the expansion (disjoint rows out, the
    // same records in) -- and join before this call's last launch returns the stream to the caller.
    static const bool no_fork = std::getenv("RAPTOR_NO_FORK") != nullptr;
    // Re-measured in round 3 (scripts/fork_sweep.sh, scripts/fork_latency.sh): with Digit's closed loops the fork wins from one problem
    // (172 -> 167 us) to 1024 problems (0.632 -> 0.619 ms per step) and loses at 4096 (2.05 -> 2.08 ms); Talos / G1 have no loop
    // kernels to run beside the core kernel and lose ~10 us at every size (one G1 problem: 87.6 us forked, 77.5 us on one stream).
    static const long long fork_max = std::getenv("RAPTOR_FORK_MAX") ? std::atoll(std::getenv("RAPTOR_FORK_MAX")) : 16384;
    static const bool fork_forced = std::getenv("RAPTOR_FORK_MAX") != nullptr;  // experiments: also under per-kernel timing
    const bool fused = jac && B > 0 && legged_can_fuse(p, dJ, E);
    double* const dJf = fused ? dJ : nullptr;
    // (the fused form has no expansion for the reset map to run beside: one stream measured 2-5 us faster at 256 .. 1024 Digit problems)
    const bool fork = !fused && !no_fork && (p.has_digit || fork_forced) && p.period_rows && jac && B > 0 && E <= fork_max && (between == nullptr || fork_forced);
    if (fork) {
        if (p.side == nullptr) {
            HIP_CHECK(hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&p.ev_loops, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&p.ev_core, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&p.ev_side, hipEventDisableTiming));
        }
        launch_legged_core(p, dZ, B, dG, jac, st, nullptr, p.ev_loops, dJf);
        HIP_CHECK(hipEventRecord(p.ev_core, st));
        HIP_CHECK(hipStreamWaitEvent(p.side, p.ev_loops, 0));
        launch_period(p, dZ, B, dG, dJ, jac, p.side, p.layout.m, p.hper.off, nullptr, nullptr, nullptr, p.ev_core);
        HIP_CHECK(hipEventRecord(p.ev_side, p.side));
    } else {
        // per-kernel timing: the fused form splits loops | core (every row but the reset map's) | reset map, the two-kernel form
        // records | (nothing) | expansion
        launch_legged_core(p, dZ, B, dG, jac, st, nullptr, fused ? split1 : nullptr, dJf);
        if (fused && between) HIP_CHECK(hipEventRecord(between, st));
        if (p.period_rows && B > 0) launch_period(p, dZ, B, dG, dJ, jac, st, p.layout.m, p.hper.off, nullptr, nullptr, nullptr);
    }
    if (!(fused && !fork)) {
        if (split1) HIP_CHECK(hipEventRecord(split1, st));
        if (between) HIP_CHECK(hipEventRecord(between, st));
    }
    if (!jac || B == 0) return;
    if (fused) {
        // every row came out of the core kernel (phase G)
    } else if (p.layout.n % 2 == 0 && ((size_t)dJ & 15) == 0) {
        if (E > 0x7fffffffLL) throw std::invalid_argument("batch too large for one launch");
        const int ndyn = p.hleg.ni * p.hleg.K;
        const int spw = p.htraj.N >= 2 ? 2 : 1;  // two consecutive samples per workgroup: longer contiguous pieces, half the prologues
        const int groups = (p.htraj.N + spw - 1) / spw;
        const int cper = (p.hleg.ncust + groups - 1) / groups;  // customized rows per workgroup: one contiguous run
        const size_t lds_bytes = (size_t)(spw * (p.hleg.c_stride + 3 * ndyn) + cper * p.hleg.ni + p.htraj.N * p.hleg.K) * sizeof(double) +
                                 (size_t)(ndyn + 3 * RAPTOR_MAX_NV + cper + 1) * sizeof(int);
        hipLaunchKernelGGL(legged_expand_kernel, dim3((unsigned)((long long)B * groups)), dim3(LEG_EXP_BLOCK), lds_bytes, st, p.d_traj.p,
                           p.d_leg.p, p.d_basis.p, p.layout, p.s_core.p, p.d_cust.p, spw, cper, 1, dJ);
    } else {
        const long long total = (long long)B * p.layout.m * p.layout.n;
        if (total / RAPTOR_BLOCK > 0x7fffffffLL) throw std::invalid_argument("batch too large for one launch");
        hipLaunchKernelGGL(legged_expand_scalar_kernel, dim3(grid_for(total)), dim3(RAPTOR_BLOCK), 0, st, p.d_traj.p, p.d_leg.p,
                           p.d_basis.p, p.layout, p.s_core.p, p.d_cust.p, total, dJ);
    }
    HIP_CHECK(hipGetLastError());
    if (fork) HIP_CHECK(hipStreamWaitEvent(st, p.ev_side, 0));
}

void eval_device(raptor_problem& p, const double* dZ, int B, unsigned flags, double* dG, double* dJ, hipStream_t st) {
    if (!(flags & (RAPTOR_EVAL_G | RAPTOR_EVAL_JAC))) throw std::invalid_argument("flags: nothing to evaluate");
    const bool jac = (flags & RAPTOR_EVAL_JAC) != 0;
    if (jac && dJ == nullptr) throw std::invalid_argument("JAC buffer is null");
    if ((flags & RAPTOR_EVAL_G) && dG == nullptr) throw std::invalid_argument("G buffer is null");
    if (B > 0 && dZ == nullptr) throw std::invalid_argument("Z is null");
    check_launchable(p, B);
    if (B == 0) return;
    if (!(flags & RAPTOR_EVAL_G)) dG = nullptr;
    // events per evaluation: start | first split | second split | end (the two splits coincide except on the fused legged path)
    const bool prof = p.profiling && p.ev_used + 4 <= p.ev.size();
    if (prof) HIP_CHECK(hipEventRecord(p.ev[p.ev_used], st));
    if (p.kind == 1) {  // Kinova: one fused kernel
        if (jac) launch_kinova<true>(p, dZ, B, dG, dJ, st);
        else launch_kinova<false>(p, dZ, B, dG, nullptr, st);
    } else if (p.kind == 3) {  // Kinova, longer horizon: the fused kernel leaves records, a streaming kernel expands them
        static const bool per_column = std::getenv("RAPTOR_LH_PER_COLUMN") != nullptr;  // the first, per-column form (A/B, tests)
        if (per_column) {
            if (!p.hmodel.serial_chain) throw unsupported_error("inverse dynamics kernel: only serial-chain models are built so far");
            if (jac) launch_id<true>(p, p.layout, dZ, B, dG, dJ, st);
            else launch_id<false>(p, p.layout, dZ, B, dG, nullptr, st);
            launch_sphere_rows(p, dZ, B, dG, jac ? dJ : nullptr, st);
        } else if (!jac) {
            launch_kinova<false>(p, dZ, B, dG, nullptr, st);
        } else {
            const long long E = (long long)B * p.htraj.N;
            if (E > 0x7fffffffLL) throw std::invalid_argument("batch too large for one launch");
            p.s_core.ensure((size_t)E * KREC_STRIDE);
            launch_kinova<true, true>(p, dZ, B, dG, p.s_core.p, st);
            if (prof) {
                HIP_CHECK(hipEventRecord(p.ev[p.ev_used + 1], st));
                HIP_CHECK(hipEventRecord(p.ev[p.ev_used + 2], st));
            }
            int spb = 4;  // samples per workgroup: longer contiguous pieces per store stream
            if (const char* ev = getenv("RAPTOR_LH_SPB")) spb = std::max(1, std::min(8, atoi(ev)));
            const size_t per_sample = (size_t)(KREC_STRIDE + 3 * p.layout.n) * sizeof(double);
            while (spb > 1 && spb * per_sample > 64 * 1024) spb >>= 1;
            if (spb * per_sample > 64 * 1024) throw unsupported_error("Kinova expand kernel: too many decision variables");
            const long long groups = (p.htraj.N + spb - 1) / spb;
            hipLaunchKernelGGL(kinova_expand_kernel, dim3((unsigned)(B * groups)), dim3(RAPTOR_BLOCK), spb * per_sample, st, p.d_traj.p,
                               p.d_basis.p, p.layout, p.hspheres.num_spheres, spb, p.s_core.p, dJ);
            HIP_CHECK(hipGetLastError());
        }
    } else if (p.kind == 2) {  // legged: core kernel, then the streaming expansion (timed separately)
        launch_legged(p, dZ, B, dG, jac ? dJ : nullptr, jac, st, prof ? p.ev[p.ev_used + 2] : nullptr, prof ? p.ev[p.ev_used + 1] : nullptr);
    } else {
        if (jac) launch_id<true>(p, p.layout, dZ, B, dG, dJ, st);
        else launch_id<false>(p, p.layout, dZ, B, dG, nullptr, st);
    }
    if (prof && p.kind != 2 && !(p.kind == 3 && jac)) {
        HIP_CHECK(hipEventRecord(p.ev[p.ev_used + 1], st));
        HIP_CHECK(hipEventRecord(p.ev[p.ev_used + 2], st));
    }
    if (prof) {
        HIP_CHECK(hipEventRecord(p.ev[p.ev_used + 3], st));
        p.ev_used += 4;
    }
}

}  // namespace

// Fixed-size device workspace of raptor_eval_h_batch_noalloc (include/raptor_hip.h): the partial Hessians [B][7][28] of the
// split-form eval_h kernels, B <= RAPTOR_H_NOALLOC_MAX_B.  172 KB, held by every GPU the library runs on until the process ends.
// Host code needs the array's device address, which differs per GPU: it is looked up once per device with
// hipGetSymbolAddress and cached here; kernels receive that address as an ordinary pointer argument.
static __device__ __align__(256) double scratch[RAPTOR_H_NOALLOC_MAX_B * 7 * H_PAIRS];
namespace {
constexpr int kMaxDevices = 64;
double* g_scratch_addr[kMaxDevices] = {nullptr};
double* scratch_device_address() {
    int dev = -1;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) throw std::runtime_error("device index out of range for the eval_h workspace table");
    if (g_scratch_addr[dev] == nullptr) {
        double* addr = nullptr;
        HIP_CHECK(hipGetSymbolAddress((void**)&addr, HIP_SYMBOL(scratch)));
        if (addr == nullptr) throw hip_error("hipGetSymbolAddress(scratch) returned a null address");
        g_scratch_addr[dev] = addr;
    }
    return g_scratch_addr[dev];
}
}  // namespace

// Copy of up to two runs of doubles between HBM and the pinned, device-mapped host buffer of the *_host latency path: full-width
// contiguous accesses (writes to fine-grained host memory are not combined in L2, so the evaluation kernels' own 8-byte row pieces
// would each become a PCIe transaction), and no copy engine (a DMA copy costs ~0.1 ms here).
__global__ __launch_bounds__(256) void copy2_kernel(double* __restrict__ d0, const double* __restrict__ s0, const long long n0,
                                                    double* __restrict__ d1, const double* __restrict__ s1, const long long n1) {
    const long long stride = (long long)gridDim.x * 256, t0 = (long long)blockIdx.x * 256 + threadIdx.x;
    for (long long t = t0; t < n0; t += stride) d0[t] = s0[t];
    for (long long t = t0; t < n1; t += stride) d1[t] = s1[t];
}
namespace {
void launch_copy2(double* d0, const double* s0, size_t n0, double* d1, const double* s1, size_t n1, hipStream_t st) {
    const size_t nmax = std::max(n0, n1);
    if (nmax == 0) return;
    const unsigned grid = (unsigned)std::min<size_t>((nmax + 255) / 256, 2048);
    hipLaunchKernelGGL(copy2_kernel, dim3(grid), dim3(256), 0, st, d0, s0, (long long)n0, d1, s1, (long long)n1);
    HIP_CHECK(hipGetLastError());
}
}  // namespace

// bandwidth probe (scripts/bw_probe.py): every workgroup writes one contiguous chunk with 16-byte stores per lane
// chunk_doubles = -4 (blocks = workgroups, total = blocks * 8192 doubles): grid-stride order -- in every pass workgroup b writes the
// 4 KB at (pass * gridDim + b) * 4 KB, so the workgroups in flight always write one dense window (the order torch's fill uses)
__global__ __launch_bounds__(256) void probe_stride_kernel(double* __restrict__ p, const long long total) {
    for (long long el = ((long long)blockIdx.x * 256 + threadIdx.x) * 2; el < total; el += (long long)gridDim.x * 512)
        *reinterpret_cast<double2*>(p + el) = make_double2(1.0, 2.0);
}
// chunk_doubles = -5: chunked like probe_fill_kernel (64 KB per workgroup) but every lane writes 64 contiguous bytes per pass
// (four 16-byte stores); -6: the same with 32 bytes per lane
template <int W>
__global__ __launch_bounds__(256) void probe_wide_kernel(double* __restrict__ p) {
    double* dst = p + (long long)blockIdx.x * 8192;
    for (int el = threadIdx.x * 2 * W; el < 8192; el += 512 * W)
#pragma unroll
        for (int k = 0; k < W; k++) *reinterpret_cast<double2*>(dst + el + 2 * k) = make_double2(1.0, 2.0);
}
__global__ __launch_bounds__(256) void probe_fill_kernel(double* __restrict__ p, const long long chunk) {
    double* dst = p + (long long)blockIdx.x * chunk;
    for (long long el = 2 * threadIdx.x; el < chunk; el += 512) *reinterpret_cast<double2*>(dst + el) = make_double2(1.0, 2.0);
}

// worst bound violation of each decision vector: max over the rows of max(g_lb - g, g - g_ub, 0); NaN if a row is NaN.
// One workgroup per z, fixed reduction order (deterministic).
__global__ __launch_bounds__(RAPTOR_BLOCK) void violation_kernel(const double* __restrict__ G, const double* __restrict__ lb,
                                                                 const double* __restrict__ ub, const int m,
                                                                 double* __restrict__ out) {
    __shared__ double red[RAPTOR_BLOCK];
    __shared__ int bad[RAPTOR_BLOCK];
    const double* g = G + (size_t)blockIdx.x * m;
    double v = 0.0;
    int nan = 0;
    for (int r = threadIdx.x; r < m; r += RAPTOR_BLOCK) {
        const double x = g[r];
        const double d = fmax(lb[r] - x, x - ub[r]);
        if (x != x) nan = 1;
        v = fmax(v, d);
    }
    red[threadIdx.x] = v;
    bad[threadIdx.x] = nan;
    __syncthreads();
    for (int s = RAPTOR_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
            bad[threadIdx.x] |= bad[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = bad[0] ? __longlong_as_double(0x7ff8000000000000LL) : red[0];
}

// chunk_doubles = -1: the write pattern of legged_expand_kernel on the Digit batch without its arithmetic -- a workgroup per two
// samples of a decision vector (1005 rows x 132 columns, 14 samples): three contiguous pieces in three blocks plus 130
// scattered rows of the customized block.  chunk_doubles = -2: the same bytes, workgroups in address order.
__global__ __launch_bounds__(256) void probe_pattern_kernel(double* __restrict__ p, const int ordered) {
    constexpr int n = 132, M = 1005;
    const long long w = blockIdx.x;
    double* zb = p + (w / 7) * (long long)M * n;
    const int pair = (int)(w % 7);
    if (ordered == 1) {  // 1005 rows in 7 consecutive runs
        const int r0 = pair * 144, r1 = min(M, r0 + 144);
        double* dst = zb + (long long)r0 * n;
        for (int el = 2 * threadIdx.x; el < (r1 - r0) * n; el += 512) *reinterpret_cast<double2*>(dst + el) = make_double2(1.0, 2.0);
        return;
    }
    const int rows[3] = {12, 36, 7}, off[3] = {0, 14 * 12, 14 * 48};
    for (int b = 0; b < 3; b++) {
        double* dst = zb + (long long)(off[b] + pair * 2 * rows[b]) * n;
        for (int el = 2 * threadIdx.x; el < 2 * rows[b] * n; el += 512) *reinterpret_cast<double2*>(dst + el) = make_double2(1.0, 2.0);
    }
    double* cb = zb + (long long)(14 * 55) * n;
    for (int el = 2 * threadIdx.x; el < 130 * n; el += 512) {
        const int r = el / n, c = el - r * n;
        const int row = (ordered == 2) ? pair * 34 + r : (r >> 1) * 14 + pair * 2 + (r & 1);  // quantity-major, sample-minor
        if (ordered == 2 && r >= 34) break;
        if (row < M - 14 * 55) *reinterpret_cast<double2*>(cb + (long long)row * n + c) = make_double2(1.0, 2.0);
    }
}

// chunk_doubles = -7: the pattern of mode -3 (three pieces + one contiguous customized run per (z, sample pair) item) written by a
// PERSISTENT grid: `blocks` workgroups walk the 4096 x 7 items of the Digit bench batch in order, item = blockIdx.x, += gridDim.x.
// What a double-buffered persistent expansion (3-4 workgroups per CU) could reach at best.
__global__ __launch_bounds__(256) void probe_persistent_kernel(double* __restrict__ p, const long long items) {
    constexpr int n = 132, M = 1005;
    for (long long w = blockIdx.x; w < items; w += gridDim.x) {
        double* zb = p + (w / 7) * (long long)M * n;
        const int pair = (int)(w % 7);
        const int rows[3] = {12, 36, 7}, off[3] = {0, 14 * 12, 14 * 48};
        for (int b = 0; b < 3; b++) {
            double* dst = zb + (long long)(off[b] + pair * 2 * rows[b]) * n;
            for (int el = 2 * threadIdx.x; el < 2 * rows[b] * n; el += 512) *reinterpret_cast<double2*>(dst + el) = make_double2(1.0, 2.0);
        }
        double* cb = zb + (long long)(14 * 55 + pair * 21) * n;
        const int nr = min(21, 144 - pair * 21);
        for (int el = 2 * threadIdx.x; el < nr * n; el += 512) *reinterpret_cast<double2*>(cb + el) = make_double2(1.0, 2.0);
    }
}

extern "C" {

int raptor_probe_fill(double* d_p, long long blocks, long long chunk_doubles, void* stream) {
    if (chunk_doubles == -5 || chunk_doubles == -6) {
        if (chunk_doubles == -5) hipLaunchKernelGGL(probe_wide_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_p);
        else hipLaunchKernelGGL(probe_wide_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_p);
        return hipGetLastError() == hipSuccess ? RAPTOR_OK : RAPTOR_ERR_HIP;
    }
    if (chunk_doubles == -7) {
        hipLaunchKernelGGL(probe_persistent_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_p, 4096LL * 7);
        return hipGetLastError() == hipSuccess ? RAPTOR_OK : RAPTOR_ERR_HIP;
    }
    if (chunk_doubles == -4) {
        hipLaunchKernelGGL(probe_stride_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_p, blocks * 8192);
        return hipGetLastError() == hipSuccess ? RAPTOR_OK : RAPTOR_ERR_HIP;
    }
    if (chunk_doubles < 0) {
        hipLaunchKernelGGL(probe_pattern_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_p, (int)(-1 - chunk_doubles));
        return hipGetLastError() == hipSuccess ? RAPTOR_OK : RAPTOR_ERR_HIP;
    }
    hipLaunchKernelGGL(probe_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_p, chunk_doubles);
    return hipGetLastError() == hipSuccess ? RAPTOR_OK : RAPTOR_ERR_HIP;
}

const char* raptor_last_error(void) { return g_err.c_str(); }

int raptor_init(int device) {
    API_TRY
    int count = 0;
    HIP_CHECK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) throw std::invalid_argument("raptor_init: no such device");
    HIP_CHECK(hipSetDevice(device));
    (void)scratch_device_address();  // set-up of the allocation-free eval_h entry point: this device's workspace address
    API_CATCH
}

int raptor_model_create(const raptor_model_desc* desc, raptor_model** out) {
    API_TRY
    if (!out) throw std::invalid_argument("out is null");
    std::unique_ptr<raptor_model> m(new raptor_model());
    fill_model(desc, m->h);
    *out = m.release();
    API_CATCH
}
void raptor_model_destroy(raptor_model* m) { delete m; }

int raptor_id_problem_create(const raptor_model* model, const raptor_traj_desc* traj, raptor_problem** out) {
    API_TRY
    if (!model || !out) throw std::invalid_argument("null argument");
    std::unique_ptr<raptor_problem> p(new raptor_problem());
    p->kind = 0;
    p->hmodel = model->h;
    std::vector<double> basis, wsum;
    fill_traj(traj, p->htraj, p->tspan, basis, wsum);
    if (p->htraj.Nact != p->hmodel.nv) throw std::invalid_argument("InverseDynamics: trajectory Nact must equal model nv");
    std::memset(&p->hspheres, 0, sizeof(p->hspheres));
    const int rows = p->htraj.N * p->htraj.Nact;
    p->layout = DevLayout{rows, p->htraj.varLength, -1, -1, 0, -1};
    p->num_blocks = 1;
    p->block_offsets[0] = 0;
    p->block_offsets[1] = rows;
    p->g_lb.assign(rows, -1e19);
    p->g_ub.assign(rows, 1e19);
    if (p->htraj.type == RAPTOR_TRAJ_PIECEWISE_BEZIER) p->hbasis = basis;  // raptor_cost_hessian_host reads the column weights
    upload_common(*p, basis, wsum);
    finish_problem(*p);
    *out = p.release();
    API_CATCH
}

int raptor_kinova_problem_create(const raptor_model* model, const raptor_kinova_desc* d, raptor_problem** out) {
    API_TRY
    if (!model || !d || !out) throw std::invalid_argument("null argument");
    std::unique_ptr<raptor_problem> p(new raptor_problem());
    p->hmodel = model->h;
    if (d->traj.type != RAPTOR_TRAJ_ARMOUR_BEZIER && d->traj.type != RAPTOR_TRAJ_PIECEWISE_BEZIER)
        throw std::invalid_argument("KinovaOptimizer uses ArmourBezierCurves, KinovaLongerHorizonOptimizer PiecewiseBezierCurves");
    // 1: KinovaOptimizer (fused kernels, diagonal pq_pz); 3: KinovaLongerHorizonOptimizer (per-column kernels)
    p->kind = (d->traj.type == RAPTOR_TRAJ_ARMOUR_BEZIER) ? 1 : 3;
    std::vector<double> basis, wsum;
    fill_traj(&d->traj, p->htraj, p->tspan, basis, wsum);
    if (p->hmodel.nv != 7 || p->htraj.Nact != 7) throw std::invalid_argument("KinovaOptimizer: NUM_JOINTS is 7");
    if (d->num_obstacles < 0) throw std::invalid_argument("num_obstacles < 0");
    if (d->num_obstacles > 0 && (!d->box_center || !d->box_rpy || !d->box_size))
        throw std::invalid_argument("boxCenters, boxOrientation, and boxSize should have the same size");
    const int N = p->htraj.N;

    // sphere tables: KinovaCustomizedConstraints.h:21-54, gripper spheres KinovaCustomizedConstraints.cpp:42-47
    static const int JID[17] = {2, 2, 2, 2, 2, 2, 2, 4, 4, 4, 4, 4, 6, 6, 7, 7, 7};
    static const double OFF[17][3] = {
        {0.0, -0.05, -0.01}, {0.0, -0.11, -0.01}, {0.0, -0.17, -0.01}, {0.0, -0.24, -0.01}, {0.0, -0.30, -0.01},
        {0.0, -0.36, -0.01}, {0.0, -0.43, -0.01}, {0.0, -0.07, 0.0},   {0.0, -0.14, 0.0},   {0.0, -0.21, 0.0},
        {0.0, -0.28, 0.0},   {0.0, -0.32, 0.0},   {0.0, -0.07, 0.0},   {0.0, -0.14, 0.0},   {-0.03, -0.06, -0.055},
        {0.0, -0.06, -0.055}, {0.03, -0.06, -0.055}};
    static const double RAD[17] = {0.06, 0.06, 0.06, 0.06, 0.06, 0.06, 0.07, 0.07, 0.06, 0.06, 0.06, 0.06, 0.05, 0.05, 0.02, 0.02, 0.02};
    DevSpheres& S = p->hspheres;
    std::memset(&S, 0, sizeof(S));
    S.num_spheres = 17;
    for (int s = 0; s < 17; s++) {
        S.joint_id[s] = JID[s];
        for (int k = 0; k < 3; k++) S.offset[s][k] = OFF[s][k];
        S.radius[s] = RAD[s];
    }
    if (d->include_gripper) {
        const double zs[3] = {-0.10, -0.15, -0.20};
        for (int k = 0; k < 3; k++) {
            const int s = S.num_spheres++;
            S.joint_id[s] = 7;
            S.offset[s][0] = 0.0; S.offset[s][1] = 0.0; S.offset[s][2] = zs[k];
            S.radius[s] = 0.05;
        }
    }
    S.num_obstacles = d->num_obstacles;
    {   // links that carry spheres publish their world pose to the 8-lane group (kinova_kernel.h, KL_RO)
        for (int l = 0; l < RAPTOR_MAX_NV; l++) S.ro_slot[l] = -1;
        int nslots = 0;
        for (int s = 0; s < S.num_spheres; s++) {
            const int l = S.joint_id[s] - 1;
            if (S.ro_slot[l] < 0) S.ro_slot[l] = nslots++;
            S.slot[s] = S.ro_slot[l];
        }
        if (nslots > 4 || S.num_spheres > 20) throw unsupported_error("Kinova kernel: at most 4 sphere-carrying links / 20 spheres");
    }

    // boxes: R = Rz(yaw) Ry(pitch) Rx(roll)  (BoxCollisionAvoidance.cpp:59-61)
    std::vector<DevBox> boxes(std::max(1, d->num_obstacles));
    std::memset(boxes.data(), 0, boxes.size() * sizeof(DevBox));
    for (int o = 0; o < d->num_obstacles; o++) {
        const double* rpy = d->box_rpy + 3 * o;
        const double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]);
        const double cp = std::cos(rpy[1]), sp = std::sin(rpy[1]);
        const double cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
        const double Rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr};
        const double Ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp};
        const double Rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1};
        double ZY[9], R[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += Rz[3 * i + k] * Ry[3 * k + j];
                ZY[3 * i + j] = s;
            }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += ZY[3 * i + k] * Rx[3 * k + j];
                R[3 * i + j] = s;
            }
        for (int k = 0; k < 9; k++) boxes[o].R[k] = R[k];
        for (int k = 0; k < 3; k++) {
            boxes[o].center[k] = d->box_center[3 * o + k];
            boxes[o].size[k] = d->box_size[3 * o + k];
            boxes[o].hs[k] = d->box_size[3 * o + k] / 2.0;  // BoxCollisionAvoidance.cpp:101
            boxes[o].cl[k] = R[k] * d->box_center[3 * o] + R[3 + k] * d->box_center[3 * o + 1] + R[6 + k] * d->box_center[3 * o + 2];
        }
    }

    // layout, KinovaOptimizer.cpp:82-109
    const int r7 = N * 7, rs = N * S.num_spheres;
    p->layout = DevLayout{3 * r7 + rs, p->htraj.varLength, 0, r7, 2 * r7, 3 * r7};
    p->num_blocks = 4;
    p->block_offsets[0] = 0;
    p->block_offsets[1] = r7;
    p->block_offsets[2] = 2 * r7;
    p->block_offsets[3] = 3 * r7;
    p->block_offsets[4] = 3 * r7 + rs;

    // bounds, KinovaOptimizer.cpp:52-80 + KinovaConstants.h:11-27 + *Limits::compute_bounds
    static const double JL[7] = {-1e19, -128.9, -1e19, -147.8, -1e19, -120.3, -1e19};
    static const double JU[7] = {1e19, 128.9, 1e19, 147.8, 1e19, 120.3, 1e19};
    static const double VL[7] = {-100, -100, -100, -100, -180, -180, -180};
    static const double VU[7] = {100, 100, 100, 100, 180, 180, 180};
    static const double TL[7] = {-90.0, -90.0, -90.0, -90.0, -45.0, -45.0, -45.0};
    static const double TU[7] = {90.0, 90.0, 90.0, 90.0, 45.0, 45.0, 45.0};
    p->g_lb.assign(p->layout.m, 0.0);
    p->g_ub.assign(p->layout.m, 0.0);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < 7; j++) {
            const double jb = d->joint_limits_buffer ? d->joint_limits_buffer[j] : 0.0;
            const double vb = d->velocity_limits_buffer ? d->velocity_limits_buffer[j] : 0.0;
            const double tb = d->torque_limits_buffer ? d->torque_limits_buffer[j] : 0.0;
            p->g_lb[i * 7 + j] = deg2rad(JL[j]) + jb;
            p->g_ub[i * 7 + j] = deg2rad(JU[j]) - jb;
            p->g_lb[r7 + i * 7 + j] = deg2rad(VL[j]) + vb;
            p->g_ub[r7 + i * 7 + j] = deg2rad(VU[j]) - vb;
            p->g_lb[2 * r7 + i * 7 + j] = TL[j] + tb;
            p->g_ub[2 * r7 + i * 7 + j] = TU[j] - tb;
        }
    for (int r = 0; r < rs; r++) {  // KinovaCustomizedConstraints.cpp:160-164
        p->g_lb[3 * r7 + r] = d->collision_buffer;
        p->g_ub[3 * r7 + r] = 1e19;
    }

    upload_common(*p, basis, wsum);
    if (p->kind == 3) p->hbasis = basis;
    p->d_spheres.upload(&S, 1);
    p->d_boxes.upload(boxes.data(), boxes.size());
    finish_problem(*p);
    *out = p.release();
    API_CATCH
}

// what legged_setup needs beyond the model: the arguments of the single-step optimizers that reach the hot path
struct LeggedSetup {
    const raptor_traj_desc* traj;
    int extra_variables;
    std::vector<int> dep, ind;          // model joint indices, reference order (floating base first)
    int contact_joint_id;               // Pinocchio id (1-based)
    double endT_R[9], endT_p[3];        // stance_foot_endT
    const double *Tdes_R, *Tdes_p;
    const double *tl, *tu, *jl, *ju;
    double mu, gamma, Lx, Ly;
    // closed loops: for a dependent joint that is not part of the floating base, its record and its driving joints
    std::vector<int> loop_rec, loop_drv_a, loop_drv_b;  // indexed by model joint, -1: none
    int nrec = 0;
    // customized block
    bool cust = false, cust_digit = false, stanceL = true;
    int swing_joint_id = 0;             // Pinocchio id
    double sf_R[9], sf_p[3];            // swing-foot end transform
    raptor_gait_desc gait;
    // reset-map block
    bool period = false, period_rows = false;   // tables; rows in this problem (false: link only)
    int period_contact_joint_id = 0;    // Pinocchio id of the stance foot after reinitialize()
    double period_endR[9], period_endp[3];
    std::vector<int> period_id1, period_id2;
    double period_sign = -1.0;
    struct PeriodLoop { int nvl, nr, nx, row0, drv[2], var_joint[8]; };
    std::vector<PeriodLoop> period_loops;   // Digit: index leg * 2 + (0 knee, 1 toe)
};

static void legged_setup(raptor_problem& pr, const raptor_model* model, const LeggedSetup& S) {
    raptor_problem* p = &pr;
    p->kind = 2;
    p->hmodel = model->h;
    const DevModel& m = p->hmodel;
    const int nv = m.nv;
    if (S.traj->type != RAPTOR_TRAJ_BEZIER) throw std::invalid_argument("single-step optimizers use BezierCurves");
    std::vector<double> basis, wsum;
    fill_traj(S.traj, p->htraj, p->tspan, basis, wsum);
    if (S.extra_variables < 0) throw std::invalid_argument("extra_variables < 0");
    p->htraj.varLength += S.extra_variables;  // TalosSingleStepOptimizer.cpp:50-51, DigitSingleStepOptimizer.cpp:50-51
    const int N = p->htraj.N, ni = (int)S.ind.size(), nd = (int)S.dep.size(), K = p->htraj.K;
    if (ni != p->htraj.Nact || nd + ni != nv)
        throw std::invalid_argument("Trajectory and DynamicsConstraints must have the same number of actuated joints");
    if (!S.tl || !S.tu || !S.jl || !S.ju) throw std::invalid_argument("null array");
    if (ni > RAPTOR_LEG_MAX_IND) throw unsupported_error("legged kernel: at most 12 independent joints");
    DevLegged& Q = p->hleg;
    std::memset(&Q, 0, sizeof(Q));
    Q.ni = ni; Q.nd = nd; Q.nv = nv; Q.K = K; Q.nrec = S.nrec;
    for (int j = 0; j < nv; j++) Q.joint_ind[j] = Q.joint_dep[j] = -1;
    for (int k = 0; k < nd; k++) {
        const int id = S.dep[k];
        if (id < 0 || id >= nv || Q.joint_dep[id] >= 0) throw std::invalid_argument("bad dependent joint id");
        Q.joint_dep[id] = k;
        Q.dep_joint[k] = id;
    }
    for (int k = 0; k < ni; k++) {
        const int id = S.ind[k];
        if (id < 0 || id >= nv || Q.joint_dep[id] >= 0 || Q.joint_ind[id] >= 0) throw std::invalid_argument("bad independent joint id");
        Q.joint_ind[id] = k;
        Q.ind_joint[k] = id;
    }
    if (S.contact_joint_id < 1 || S.contact_joint_id > nv) throw std::runtime_error("Can not find joint");
    // the floating base the closed form of TalosDynamicsConstraints.cpp:208-222 assumes: Px,Py,Pz,Rx,Ry,Rz in a chain
    // from the world, identity placements, massless intermediate links, no joint-space extras
    static const int base_types[6] = {4, 5, 6, 1, 2, 3};
    std::vector<int> is_base(nv, 0);
    for (int k = 0; k < 6; k++) {
        const int id = Q.dep_joint[k];
        bool ok = m.jtype[id] == base_types[k] && m.parent[id] == (k == 0 ? -1 : Q.dep_joint[k - 1]);
        for (int t = 0; t < 9; t++) ok = ok && std::fabs(m.plR[id][t] - ((t % 4 == 0) ? 1.0 : 0.0)) < 1e-12;
        for (int t = 0; t < 3; t++) ok = ok && std::fabs(m.plp[id][t]) < 1e-12;
        ok = ok && (k == 5 || m.mass[id] == 0.0) && m.armature[id] == 0.0 && m.damping[id] == 0.0;
        if (!ok) throw unsupported_error("legged kernel: the first six dependent joints must be a Px,Py,Pz,Rx,Ry,Rz floating base");
        is_base[id] = 1;
    }
    const int torso_link = Q.dep_joint[5];
    const int foot_link = S.contact_joint_id - 1;
    if (is_base[foot_link]) throw unsupported_error("legged kernel: the stance foot cannot be a floating-base joint");
    for (int j = 0; j < nv; j++)
        if (!is_base[j] && (m.parent[j] < 0 || (is_base[m.parent[j]] && m.parent[j] != torso_link)))
            throw unsupported_error("legged kernel: every leg must hang from the last floating-base link");

    // re-root at the stance-foot link: DFS pre-order over the links joined by the non-base joints
    struct Edge { int joint, other, rev; };
    std::vector<std::vector<Edge>> adj(nv);
    for (int j = 0; j < nv; j++) {
        if (is_base[j]) continue;
        adj[j].push_back(Edge{j, m.parent[j], 1});
        adj[m.parent[j]].push_back(Edge{j, j, 0});
    }
    int nb = 0;
    std::function<void(int, int, int, int, int)> visit = [&](int link, int via, int rev, int rparent, int depth) {
        if (nb >= RAPTOR_LEG_MAX_BODIES) throw unsupported_error("legged kernel: more than 32 bodies");
        if (depth > RAPTOR_LEG_MAX_DEPTH) throw unsupported_error("legged kernel: tree deeper than 20");
        const int idx = nb++;
        Q.body_link[idx] = link;
        Q.body_joint[idx] = via;
        Q.body_rev[idx] = rev;
        Q.body_rparent[idx] = rparent;
        Q.body_depth[idx] = depth;
        Q.body_traj[idx] = via >= 0 ? Q.joint_ind[via] : -1;
        Q.body_rec[idx] = -1;
        Q.body_col[idx][0] = Q.body_traj[idx];
        Q.body_col[idx][1] = -1;
        if (via >= 0 && Q.joint_ind[via] < 0) {
            const int rc = (int)S.loop_rec.size() == nv ? S.loop_rec[via] : -1;
            if (rc < 0) throw unsupported_error("legged kernel: dependent joint outside the floating base without a closed loop");
            if (m.jtype[via] > 3 || m.jtype[via] < -3) throw unsupported_error("legged kernel: closed-loop joints must be revolute");
            Q.body_rec[idx] = rc;
            Q.body_col[idx][0] = Q.joint_ind[S.loop_drv_a[via]];
            Q.body_col[idx][1] = S.loop_drv_b[via] >= 0 ? Q.joint_ind[S.loop_drv_b[via]] : -1;
        }
        for (int t = 0; t < depth - 1; t++) Q.path[idx][t] = Q.path[rparent][t];
        if (depth > 0) Q.path[idx][depth - 1] = idx;
        if (link == torso_link) Q.torso_body = idx;
        for (const Edge& e : adj[link])
            if (e.joint != via) visit(e.other, e.joint, e.rev, idx, depth + 1);
        Q.body_end[idx] = nb;
    };
    visit(foot_link, -1, 0, -1, 0);
    Q.nb = nb;
    if (nb != nv - 6 + 1) throw unsupported_error("legged kernel: every non-base joint must be reachable from the stance foot");
    for (int t = 0; t < ni; t++) {
        Q.ind_loop_col[t][0] = Q.ind_loop_col[t][1] = -1;
    }
    for (int g = 1; g < nb; g++) {
        if (Q.body_traj[g] >= 0) Q.ind_body[Q.body_traj[g]] = g;
        for (int slot = 0; slot < 2; slot++) {
            const int t = Q.body_col[g][slot];
            if (t < 0) continue;
            if (Q.col_nsupp[t] >= 8) throw unsupported_error("legged kernel: more than 8 joints move with one actuator");
            Q.col_supp_body[t][Q.col_nsupp[t]] = g;
            Q.col_supp_slot[t][Q.col_nsupp[t]] = slot;
            Q.col_nsupp[t]++;
            if (Q.body_rec[g] >= 0) {
                if (Q.ind_ndep[t] >= 6) throw unsupported_error("legged kernel: more than 6 loop joints per actuator");
                Q.ind_dep_body[t][Q.ind_ndep[t]++] = g;
                Q.ind_slot[t] = slot;
                Q.ind_loop_col[t][0] = Q.body_col[g][0];
                Q.ind_loop_col[t][1] = Q.body_col[g][1];
            }
        }
    }

    // packed copies for the core kernel (device_types.h)
    static_assert(RAPTOR_LEG_MAX_DEPTH <= 24 && RAPTOR_LEG_MAX_BODIES <= 32, "packed path / entry layout");
    static_assert(RAPTOR_LEG_MAX_IND <= 15, "flat entry: 4-bit column fields, 15 = none (and the core kernel's packed loop columns)");
    for (int g = 0; g < nb; g++)
        for (int d = 0; d < Q.body_depth[g]; d++) Q.path8[g][d >> 3] |= (unsigned long long)Q.path[g][d] << (8 * (d & 7));
    {   // pointer-doubling tables of phase A
        static_assert(RAPTOR_LEG_MAX_DEPTH < 32, "five doubling rounds cover 32 levels");
        int maxd = 0;
        for (int g = 0; g < RAPTOR_LEG_MAX_BODIES; g++) Q.jump_pk[g] = 0u;
        for (int g = 0; g < nb; g++) {
            const int D = Q.body_depth[g];
            maxd = std::max(maxd, D);
            for (int r = 0; r < 5; r++) {
                const int up = D - (1 << r);  // depth of the ancestor 2^r levels up
                if (up < 0) continue;
                const int anc = (up == 0) ? 0 : Q.path[g][up - 1];
                Q.jump_pk[g] |= ((unsigned)anc | 32u) << (6 * r);
            }
        }
        Q.nrounds = 0;
        while ((1 << Q.nrounds) < maxd + 1) Q.nrounds++;
    }
    for (int t = 0; t < ni; t++)
        for (int r = 0; r < Q.col_nsupp[t]; r++) {
            const int k = Q.col_supp_body[t][r];
            const unsigned ent = (unsigned)k | (unsigned)Q.body_end[k] << 5 | (unsigned)Q.col_supp_slot[t][r] << 11;
            Q.col_pack[t][r >> 2] |= (unsigned long long)(ent | 1u << 12) << (16 * (r & 3));
        }
    {   // the row pass's work list (legged_kernel.h phase F2): columns that share two-driver joints are merged into one group
        Q.flat_n = 0;
        std::vector<int> partner(ni, -1);
        for (int g = 1; g < nb; g++) {
            const int c0 = Q.body_col[g][0], c1 = Q.body_col[g][1];
            if (c0 < 0 || c1 < 0) continue;
            if ((partner[c0] >= 0 && partner[c0] != c1) || (partner[c1] >= 0 && partner[c1] != c0))
                throw unsupported_error("legged kernel: a driver shared between different closed loops");
            partner[c0] = c1;
            partner[c1] = c0;
        }
        auto push = [&](unsigned ent) {
            if (Q.flat_n >= 64) throw unsupported_error("legged kernel: more than 64 (actuator, tree joint) pairs");
            Q.flat[Q.flat_n++] = (int)ent;
        };
        std::vector<char> done(ni, 0);
        for (int t = 0; t < ni; t++) {
            if (done[t]) continue;
            const int t2 = partner[t];
            const int first = Q.flat_n;
            const unsigned cols = (unsigned)t << 15 | (unsigned)(t2 >= 0 ? t2 : 15) << 19;
            for (int g = 1; g < nb; g++) {
                const int c0 = Q.body_col[g][0], c1 = Q.body_col[g][1];
                const unsigned base = (unsigned)g | (unsigned)Q.body_end[g] << 5 | cols;
                if (t2 >= 0 && c0 >= 0 && c1 >= 0 && ((c0 == t && c1 == t2) || (c0 == t2 && c1 == t))) {
                    if (c0 != t) throw unsupported_error("legged kernel: two-driver joints must list their drivers in one order");
                    push(base | 2u << 13);                                  // both sets: slot 0 -> column t, slot 1 -> column t2
                } else {
                    for (int slot = 0; slot < 2; slot++) {
                        const int c = Q.body_col[g][slot];
                        if (c == t) push(base | (unsigned)slot << 11 | 0u << 13);
                        else if (t2 >= 0 && c == t2) push(base | (unsigned)slot << 11 | 1u << 13);
                    }
                }
            }
            if (Q.flat_n == first) throw unsupported_error("legged kernel: an independent joint that moves no tree joint");
            Q.flat[Q.flat_n - 1] |= (int)(1u << 12);
            done[t] = 1;
            if (t2 >= 0) done[t2] = 1;
        }
    }
    for (int t = 0; t < ni; t++)
        if (Q.col_nsupp[t] == 0) throw unsupported_error("legged kernel: an independent joint that moves no tree joint");

    // root body frame = T_des * endT^-1   (TalosDynamicsConstraints.cpp:34-38,214-218; Digit: rotated end transform)
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) {
            double sum = 0;
            for (int k = 0; k < 3; k++) sum += S.Tdes_R[3 * r + k] * S.endT_R[3 * c + k];  // R_des * R_e^T
            Q.R0[3 * r + c] = sum;
        }
        Q.p_foot[r] = S.Tdes_p[r];
    }
    for (int r = 0; r < 3; r++) {
        double sum = 0;
        for (int c = 0; c < 3; c++) sum += Q.R0[3 * r + c] * S.endT_p[c];
        Q.o0[r] = Q.p_foot[r] - sum;
    }
    {   // A_des = W^-1, W = d(rpy)/d(omega) at R_des
        double W[9];
        for (int k = 0; k < 3; k++) {
            double w[3] = {0, 0, 0}, o[3];
            w[k] = 1.0;
            rpy_rates(S.Tdes_R, w, o);
            for (int r = 0; r < 3; r++) W[3 * r + k] = o[r];
        }
        const double det = W[0] * (W[4] * W[8] - W[5] * W[7]) - W[1] * (W[3] * W[8] - W[5] * W[6]) + W[2] * (W[3] * W[7] - W[4] * W[6]);
        if (!(std::fabs(det) > 1e-12) || !std::isfinite(det)) throw std::runtime_error("constraint jacobian is not full rank!");
        const double id = 1.0 / det;
        Q.Ades[0] = (W[4] * W[8] - W[5] * W[7]) * id; Q.Ades[1] = (W[2] * W[7] - W[1] * W[8]) * id; Q.Ades[2] = (W[1] * W[5] - W[2] * W[4]) * id;
        Q.Ades[3] = (W[5] * W[6] - W[3] * W[8]) * id; Q.Ades[4] = (W[0] * W[8] - W[2] * W[6]) * id; Q.Ades[5] = (W[2] * W[3] - W[0] * W[5]) * id;
        Q.Ades[6] = (W[3] * W[7] - W[4] * W[6]) * id; Q.Ades[7] = (W[1] * W[6] - W[0] * W[7]) * id; Q.Ades[8] = (W[0] * W[4] - W[1] * W[3]) * id;
    }
    Q.mu = S.mu; Q.gamma = S.gamma; Q.Lx = S.Lx; Q.Ly = S.Ly;

    // layout, TalosSingleStepOptimizer.cpp:69-92 / DigitSingleStepOptimizer.cpp:74-94: [torque limits | joint limits | contact]
    const int rt = N * ni, rq = N * nv, rcn = N * 7;
    Q.off_tau = 0; Q.off_q = rt; Q.off_contact = rt + rq;
    // customized block: rows in the reference's order, each a linear form of one sample's pose quantities
    std::vector<DevCustRow> crow;
    std::vector<double> clb, cub;
    if (S.cust) {
        if (S.swing_joint_id < 1 || S.swing_joint_id > nv) throw std::runtime_error("Can not find joint");
        Q.sf_body = -1;
        for (int g = 0; g < nb; g++)
            if (Q.body_link[g] == S.swing_joint_id - 1) Q.sf_body = g;
        if (Q.sf_body < 0) throw unsupported_error("legged kernel: the swing foot is not a body of the tree");
        for (int k = 0; k < 9; k++) Q.sf_R[k] = S.sf_R[k];
        for (int k = 0; k < 3; k++) Q.sf_p[k] = S.sf_p[k];
        const raptor_gait_desc& gp = S.gait;
        auto add = [&](int sample, int qa, double ca, int qb, double cb, int wrap, double addv, double lo, double hi) {
            crow.push_back(DevCustRow{sample, qa, qb, wrap, ca, cb, addv});
            clb.push_back(lo);
            cub.push_back(hi);
        };
        const double e = gp.eps_torso_angle, yaw = S.stanceL ? M_PI_2 : -M_PI_2;
        for (int i = 0; i < N; i++)  // g1: swing-foot height (compute_bounds :160-167)
            add(i, 2, 1, -1, 0, 0, 0, (i == N / 2) ? gp.swingfoot_midstep_z_des : 0.0, (i == 0 || i == N - 1) ? 0.0 : 1e19);
        for (int i = 0; i < N; i++) add(i, 3, 1, -1, 0, 1, 0, 0, 0);                                  // g2 roll
        for (int i = 0; i < N; i++) add(i, 4, 1, -1, 0, 1, 0, 0, 0);                                  // g3 pitch
        for (int i = 0; i < N; i++) add(i, 5, 1, -1, 0, 1, S.cust_digit ? yaw : 0.0, 0, 0);           // g4 yaw
        if (S.cust_digit) {  // DigitCustomizedConstraints.cpp:103-124, 176-200
            add(0, 0, 1, -1, 0, 0, 0, gp.swingfoot_begin_x_des, gp.swingfoot_begin_x_des);
            add(0, 1, 1, -1, 0, 0, 0, gp.swingfoot_begin_y_des, gp.swingfoot_begin_y_des);
            add(N - 1, 0, 1, -1, 0, 0, 0, gp.swingfoot_end_x_des, gp.swingfoot_end_x_des);
            add(N - 1, 1, 1, -1, 0, 0, 0, gp.swingfoot_end_y_des, gp.swingfoot_end_y_des);
            for (int i = 0; i < N; i++) add(i, 8, 1, -1, 0, 0, 0, 1.05, 1e19);                        // g6 torso height
            for (int i = 0; i < N; i++) add(i, 6, 1, 0, -0.5, 0, 0, -0.05, 0.05);                     // g7
            for (int i = 0; i < N; i++) add(i, 7, 1, 1, -0.5, 0, 0, -0.10, 0.10);                     // g8
            for (int i = 0; i < N; i++) add(i, 9, 1, -1, 0, 1, 0, -e, e);                             // g9
            for (int i = 0; i < N; i++) add(i, 10, 1, -1, 0, 1, 0, -e, e);                            // g10
            for (int i = 0; i < N; i++) add(i, 11, 1, -1, 0, 1, yaw, -e, e);                          // g11
        } else {  // TalosCustomizedConstraints.cpp:98-124, 176-200
            add(0, 0, 1, -1, 0, 0, -gp.swingfoot_begin_x_des, 0, 0);
            add(N - 1, 0, 1, -1, 0, 0, -gp.swingfoot_end_x_des, 0, 0);
            for (int i = 0; i < N; i++) add(i, 1, 1, -1, 0, 0, -gp.swingfoot_begin_y_des, 0, 0);
            for (int i = 0; i < N; i++) add(i, 8, 1, -1, 0, 0, 0, gp.torso_height_lb, 1e19);          // g6
            for (int i = 0; i < N; i++) add(i, 9, 1, -1, 0, 1, 0, -e, e);                             // g7
            for (int i = 0; i < N; i++) add(i, 10, 1, -1, 0, 1, 0, -e, e);                            // g8
            for (int i = 0; i < N; i++) add(i, 11, 1, -1, 0, 1, 0, -e, e);                            // g9
            for (int i = 0; i < N; i++) add(i, 7, 1, -1, 0, 0, 0, -1e19, 0);                          // g10
            for (int i = 0; i < N; i++) add(i, 7, 1, 1, -1.0, 0, 0, 0, 1e19);                         // g11
        }
    }
    Q.ncust = (int)crow.size();
    Q.off_cust = rt + rq + rcn;
    std::memset(&p->hspheres, 0, sizeof(p->hspheres));
    // reset-map block (*SingleStepPeriodicityConstraints.cpp:12-16): nv + n_dep + Nact + Nact + 7 rows, last
    Q.nperiod = S.period ? nv + nd + 2 * ni + 7 : 0;
    const int prow = S.period_rows ? Q.nperiod : 0;
    Q.off_period = rt + rq + rcn + Q.ncust;
    if (S.period) {
        if (S.extra_variables != nv + nd) throw std::invalid_argument("periodicity constraints need v_reset and lambda_reset in z");
        if (S.period_contact_joint_id < 1 || S.period_contact_joint_id > nv) throw std::runtime_error("Can not find joint");
        if ((int)S.period_id1.size() != ni || (int)S.period_id2.size() != ni) throw std::invalid_argument("JOINT_MAP must have one pair per independent joint");
        if (nd > 24) throw unsupported_error("reset-map kernel: at most 24 dependent joints");
        DevPeriod& P = p->hper;
        std::memset(&P, 0, sizeof(P));
        P.nj = nv;
        P.off = Q.off_period;
        std::vector<std::vector<int>> kids(nv);
        std::vector<int> roots;
        for (int j = 0; j < nv; j++) (m.parent[j] < 0 ? roots : kids[m.parent[j]]).push_back(j);
        int nl = 0;
        std::function<void(int, int, int)> walk = [&](int j, int parent_lane, int depth) {
            if (depth > RAPTOR_LEG_MAX_DEPTH) throw unsupported_error("reset-map kernel: tree deeper than 20");
            const int lane = nl++;
            P.lane_joint[lane] = j;
            P.joint_lane[j] = lane;
            P.depth[lane] = depth;
            for (int t = 0; t < depth - 1; t++) P.path[lane][t] = P.path[parent_lane][t];
            P.path[lane][depth - 1] = lane;
            for (int c : kids[j]) walk(c, lane, depth + 1);
            P.end[lane] = nl;
        };
        for (int r : roots) walk(r, -1, 1);
        static_assert(RAPTOR_LEG_MAX_DEPTH % 4 == 0, "path8 rows are whole 8-byte words");
        for (int lane = 0; lane < nl; lane++) {
            unsigned char* b8 = reinterpret_cast<unsigned char*>(P.path8[lane]);
            for (int t = 0; t < P.depth[lane]; t++) {
                b8[t] = (unsigned char)P.path[lane][t];
                b8[RAPTOR_LEG_MAX_DEPTH + t] = (unsigned char)P.lane_joint[P.path[lane][t]];
            }
        }
        P.old_foot_lane = P.joint_lane[S.contact_joint_id - 1];
        P.new_foot_lane = P.joint_lane[S.period_contact_joint_id - 1];
        for (int k = 0; k < 9; k++) P.endR[k] = S.period_endR[k];
        for (int k = 0; k < 3; k++) P.endp[k] = S.period_endp[k];
        for (int k = 0; k < ni; k++) {
            const int a = S.period_id1[k], c = S.period_id2[k];
            if (a < 0 || a >= nv || c < 0 || c >= nv) throw std::runtime_error("Can not find joint");
            P.id1[k] = a;
            P.id2[k] = c;
        }
        P.sign = S.period_sign;
        for (int j = 0; j < nv; j++) {
            P.joint_loop[j] = P.joint_var[j] = -1;
            P.joint_rec[j] = (int)S.loop_rec.size() == nv ? S.loop_rec[j] : -1;
        }
        P.nloops = (int)S.period_loops.size();
        if (P.nloops > RAPTOR_PERIOD_MAX_LOOPS) throw unsupported_error("reset-map kernel: at most 4 closed loops");
        int off = 0, rows = 0;
        for (int l = 0; l < P.nloops; l++) {
            const LeggedSetup::PeriodLoop& pl = S.period_loops[l];
            P.loop_nvl[l] = pl.nvl; P.loop_nr[l] = pl.nr; P.loop_nx[l] = pl.nx; P.loop_row0[l] = pl.row0;
            P.loop_col[l][0] = Q.joint_ind[pl.drv[0]];
            P.loop_col[l][1] = pl.nx > 1 ? Q.joint_ind[pl.drv[1]] : -1;
            P.loop_off[l] = off;
            off += pl.nr * pl.nvl + pl.nr + pl.nvl + pl.nr * pl.nx + pl.nvl * pl.nx;
            for (int k = 0; k < pl.nvl; k++) {
                P.joint_loop[pl.var_joint[k]] = l;
                P.joint_var[pl.var_joint[k]] = k;
            }
            for (int r = 0; r < pl.nr; r++) {
                if (pl.row0 + r >= 24) throw unsupported_error("reset-map kernel: closure row out of range");
                P.row_loop[pl.row0 + r] = l;
                P.row_local[pl.row0 + r] = r;
            }
            rows += pl.nr;
        }
        if (rows != nd - 6) throw unsupported_error("reset-map kernel: closure rows must match the dependent joints");
        P.loop_stride = (off + 1) & ~1;
        p->d_per.upload(&P, 1);
        p->has_period = true;
        p->period_rows = S.period_rows;
    }
    p->layout = DevLayout{rt + rq + rcn + Q.ncust + prow, p->htraj.varLength, Q.off_q, -1, Q.off_tau, -1};
    p->num_blocks = 3;
    p->block_offsets[0] = 0; p->block_offsets[1] = rt; p->block_offsets[2] = rt + rq; p->block_offsets[3] = rt + rq + rcn;
    if (S.cust) p->block_offsets[++p->num_blocks] = rt + rq + rcn + Q.ncust;
    if (S.period_rows) p->block_offsets[++p->num_blocks] = rt + rq + rcn + Q.ncust + prow;
    p->g_lb.assign(p->layout.m, 0.0);
    p->g_ub.assign(p->layout.m, 0.0);
    if (S.period_rows) {  // *SingleStepPeriodicityConstraints.cpp:217-226: equalities, then the seven contact rows
        const int o5 = Q.off_period + nv + nd + 2 * ni;
        p->g_ub[o5] = 1e19;
        for (int k = 1; k < 7; k++) p->g_lb[o5 + k] = -1e19;
    }
    for (int r = 0; r < Q.ncust; r++) {
        p->g_lb[Q.off_cust + r] = clb[r];
        p->g_ub[Q.off_cust + r] = cub[r];
    }
    {   // per-sample row lists (always uploaded: the kernels take the pointers even when the block is absent)
        std::vector<int> start(N + 1, 0), list(std::max(1, Q.ncust), 0);
        for (const DevCustRow& c : crow) start[c.sample + 1]++;
        for (int i = 0; i < N; i++) start[i + 1] += start[i];
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (int r = 0; r < Q.ncust; r++) list[fill[crow[r].sample]++] = r;
        p->cust_max_per_sample = 0;
        for (int i = 0; i < N; i++) p->cust_max_per_sample = std::max(p->cust_max_per_sample, start[i + 1] - start[i]);
        if (crow.empty()) crow.push_back(DevCustRow{0, 0, -1, 0, 0, 0, 0});
        p->d_cust.upload(crow.data(), crow.size());
        p->d_cust_list.upload(list.data(), list.size());
        p->d_cust_start.upload(start.data(), start.size());
    }
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < ni; j++) {  // TorqueLimits.cpp:45-50
            p->g_lb[i * ni + j] = S.tl[j];
            p->g_ub[i * ni + j] = S.tu[j];
        }
        for (int j = 0; j < nv; j++) {  // ConstrainedJointLimits.cpp:73-78
            p->g_lb[rt + i * nv + j] = S.jl[j];
            p->g_ub[rt + i * nv + j] = S.ju[j];
        }
        p->g_lb[rt + rq + i * 7] = 0;  // RectangleSurfaceContactConstraints.cpp:97-125
        p->g_ub[rt + rq + i * 7] = 1e19;
        for (int k = 1; k < 7; k++) {
            p->g_lb[rt + rq + i * 7 + k] = -1e19;
            p->g_ub[rt + rq + i * 7 + k] = 0;
        }
    }
    // compact per-sample record
    int off = 0;
    Q.c_tau = off; off += ni;
    Q.c_lam = off; off += 6;
    Q.c_q = off; off += nv;
    Q.c_v = off; off += nv;
    Q.c_a = off; off += nv;
    Q.c_D = off; off += ni * 3 * ni;
    Q.c_L = off; off += 6 * 3 * ni;
    Q.c_P = off; off += ni * 6;
    Q.c_Pl = off; off += (nd - 6) * 2;
    Q.c_sf = off; off += 6;
    Q.c_Psf = off; off += ni * 6;
    Q.c_stride = (off + 1) & ~1;
    for (int dd = 6; dd < nd; dd++) {
        Q.dep_col[dd][0] = Q.dep_col[dd][1] = -1;
        Q.dep_body[dd] = 0;
        for (int g = 1; g < nb; g++)
            if (Q.body_joint[g] == Q.dep_joint[dd]) {
                Q.dep_body[dd] = g;
                Q.dep_col[dd][0] = Q.body_col[g][0];
                Q.dep_col[dd][1] = Q.body_col[g][1];
            }
    }

    upload_common(*p, basis, wsum);
    p->d_leg.upload(&Q, 1);
}

int raptor_legged_problem_create(const raptor_model* model, const raptor_legged_desc* d, raptor_problem** out) {
    API_TRY
    if (!model || !d || !out) throw std::invalid_argument("null argument");
    if (d->num_dependent != 6) throw std::invalid_argument("contact constraint: six dependent joints expected");
    if (!d->dependent_ids || !d->independent_ids) throw std::invalid_argument("null array");
    if (d->num_independent < 1 || d->num_independent > RAPTOR_MAX_NV) throw std::invalid_argument("bad num_independent");
    std::unique_ptr<raptor_problem> p(new raptor_problem());
    LeggedSetup S;
    S.traj = &d->traj;
    S.extra_variables = d->extra_variables;
    S.dep.assign(d->dependent_ids, d->dependent_ids + 6);
    S.ind.assign(d->independent_ids, d->independent_ids + d->num_independent);
    S.contact_joint_id = d->contact_joint_id;
    if (d->fk_start_joint != d->dependent_ids[5] + 1)
        throw unsupported_error("legged kernel: fk_start_joint must be the last floating-base joint");
    for (int k = 0; k < 9; k++) S.endT_R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    for (int k = 0; k < 3; k++) S.endT_p[k] = d->stance_foot_endT_p[k];
    S.Tdes_R = d->stance_foot_T_des_R;
    S.Tdes_p = d->stance_foot_T_des_p;
    S.tl = d->torque_limits_lower; S.tu = d->torque_limits_upper;
    S.jl = d->joint_limits_lower; S.ju = d->joint_limits_upper;
    S.mu = d->mu; S.gamma = d->gamma; S.Lx = d->Lx; S.Ly = d->Ly;
    if (d->gait.enabled) {  // TalosSingleStepOptimizer.cpp:95-99
        if (d->stance_leg != 'L' && d->stance_leg != 'R') throw std::runtime_error("Invalid stance leg input");
        S.cust = true;
        S.cust_digit = false;
        S.stanceL = d->stance_leg == 'L';
        S.swing_joint_id = d->gait.swing_joint_id;
        for (int k = 0; k < 9; k++) S.sf_R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        for (int k = 0; k < 3; k++) S.sf_p[k] = d->gait.swingfoot_endT_p[k];
        S.gait = d->gait;
    }
    if (d->period.enabled) {  // TalosSingleStepOptimizer.cpp:101-106
        if (!d->period.joint_id1 || !d->period.joint_id2) throw std::invalid_argument("null argument");
        S.period = true;
        S.period_rows = d->period.enabled != RAPTOR_PERIOD_LINK_ONLY;
        S.period_contact_joint_id = d->period.swing_contact_joint_id;
        for (int k = 0; k < 9; k++) S.period_endR[k] = (k % 4 == 0) ? 1.0 : 0.0;
        for (int k = 0; k < 3; k++) S.period_endp[k] = d->period.swing_foot_endT_p[k];
        S.period_id1.assign(d->period.joint_id1, d->period.joint_id1 + d->num_independent);
        S.period_id2.assign(d->period.joint_id2, d->period.joint_id2 + d->num_independent);
        S.period_sign = d->period.mirror_sign;
    }
    legged_setup(*p, model, S);
    finish_problem(*p);
    *out = p.release();
    API_CATCH
}

static void fill_chain_joint(const DevModel& m, int joint, int var, DevChainJoint& cj) {
    for (int k = 0; k < 9; k++) cj.plR[k] = m.plR[joint][k];
    for (int k = 0; k < 3; k++) cj.plp[k] = m.plp[joint][k];
    cj.jtype = m.jtype[joint];
    cj.var = var;
}

int raptor_digit_problem_create(const raptor_model* model, const raptor_digit_desc* d, raptor_problem** out) {
    API_TRY
    if (!model || !d || !out) throw std::invalid_argument("null argument");
    const DevModel& m = model->h;
    const int nv = m.nv;
    if (nv != 36) throw std::invalid_argument("Digit: NUM_JOINTS is 36");
    if (d->stance_leg != 'L' && d->stance_leg != 'R' && d->stance_leg != 'l' && d->stance_leg != 'r')
        throw std::runtime_error("Invalid stance leg input");
    const int sleg = (d->stance_leg == 'L' || d->stance_leg == 'l') ? 0 : 1;
    std::vector<int> seen(nv, 0);
    auto take = [&](int id) {
        if (id < 0 || id >= nv || seen[id]++) throw std::runtime_error("Can not find joint");
        return id;
    };
    LeggedSetup S;
    S.traj = &d->traj;
    S.extra_variables = d->extra_variables;
    for (int k = 0; k < 6; k++) S.dep.push_back(take(d->base_ids[k]));  // DigitDynamicsConstraints.h:19-57 order
    for (int leg = 0; leg < 2; leg++)
        for (int k = 0; k < 9; k++) S.dep.push_back(take(d->leg_ids[leg][DIGIT_LEG_DEPENDENT[k]]));
    for (int leg = 0; leg < 2; leg++)
        for (int k = 0; k < 6; k++) S.ind.push_back(take(d->leg_ids[leg][DIGIT_LEG_INDEPENDENT[k]]));
    S.contact_joint_id = d->leg_ids[sleg][DG_TOE_ROLL] + 1;
    digit_stance_foot_endT(sleg, S.endT_R, S.endT_p);
    S.Tdes_R = d->stance_foot_T_des_R;
    S.Tdes_p = d->stance_foot_T_des_p;
    S.tl = d->torque_limits_lower; S.tu = d->torque_limits_upper;
    S.jl = d->joint_limits_lower; S.ju = d->joint_limits_upper;
    S.mu = d->mu; S.gamma = d->gamma; S.Lx = d->Lx; S.Ly = d->Ly;

    // loops: per leg a knee loop (records 0..2) and a toe loop (records 3..8); 18 records per sample-eval
    std::unique_ptr<DevDigit> D(new DevDigit());
    std::memset(D.get(), 0, sizeof(DevDigit));
    D->recs_per_sample = 18;
    S.loop_rec.assign(nv, -1);
    S.loop_drv_a.assign(nv, -1);
    S.loop_drv_b.assign(nv, -1);
    S.nrec = 18;
    std::vector<int> ind_of(nv, -1);
    for (size_t t = 0; t < S.ind.size(); t++) ind_of[S.ind[t]] = (int)t;
    for (int leg = 0; leg < 2; leg++) {
        const int* ids = d->leg_ids[leg];
        for (int type = 0; type < 2; type++) {  // 0 knee, 1 toe
            DevLoop& L = D->loop[leg][type];
            const DigitLoop& ref = DIGIT_LOOPS[type == 0 ? 2 : 0];
            L.ny = ref.ndep; L.nx = ref.ndrv; L.ncl = type == 0 ? 1 : 2;
            L.rec0 = leg * 9 + (type == 0 ? 0 : 3);
            // local variables: unknowns first (role order of ref.depj), then the drivers
            auto var_of = [&](int role) {
                for (int r = 0; r < ref.ndep; r++) if (ref.depj[r] == role) return r;
                for (int x = 0; x < ref.ndrv; x++) if (ref.drv[x] == role) return ref.ndep + x;
                throw std::runtime_error("Digit loop: joint outside the loop");
            };
            for (int c = 0; c < L.ncl; c++) {
                const DigitLoop& lp = DIGIT_LOOPS[type == 0 ? 2 : c];
                DevClosure& cl = L.cl[c];
                // chains from lp.start (exclusive) to endA / endB, following the model's parents
                auto chain = [&](int end_role, std::vector<int>& roles) {
                    roles.clear();
                    int j = ids[end_role];
                    const int stop = ids[lp.start];
                    while (j != stop) {
                        int role = -1;
                        for (int r = 0; r < DG_LEG_JOINTS; r++) if (ids[r] == j) role = r;
                        if (role < 0 || roles.size() > 3) throw std::runtime_error("Digit loop: chain leaves the leg");
                        roles.insert(roles.begin(), role);
                        j = m.parent[j];
                        if (j < 0) throw std::runtime_error("Digit loop: chain does not reach its base link");
                    }
                };
                std::vector<int> ra, rb;
                chain(lp.endA, ra);
                chain(lp.endB, rb);
                if ((int)ra.size() != (type == 0 ? 2 : 3) || rb.size() != 2) throw unsupported_error("Digit loop: unexpected chain length");
                cl.nA = (int)ra.size(); cl.nB = (int)rb.size();
                for (size_t k = 0; k < ra.size(); k++) fill_chain_joint(m, ids[ra[k]], var_of(ra[k]), cl.A[k]);
                for (size_t k = 0; k < rb.size(); k++) fill_chain_joint(m, ids[rb[k]], var_of(rb[k]), cl.B[k]);
                // the kernels index their per-lane arrays with the compile-time map of digit_kernel.h: it must be this model's
                for (size_t k = 0; k < ra.size(); k++)
                    if (cl.A[k].var != digit_chain_var(type == 1, c, true, (int)k)) throw unsupported_error("Digit loop: unexpected rod chain");
                for (size_t k = 0; k < rb.size(); k++)
                    if (cl.B[k].var != digit_chain_var(type == 1, c, false, (int)k)) throw unsupported_error("Digit loop: unexpected anchor chain");
                double an[3] = {lp.anchor[leg][0], lp.anchor[leg][1], lp.anchor[leg][2]};
                if (type == 0) digit_knee_anchor(leg, an);
                for (int k = 0; k < 3; k++) { cl.endA[k] = lp.rod[k]; cl.endB[k] = an[k]; }
            }
            if (L.ncl == 2)  // digit_kernel.h evaluates the anchor chain once for both closures of the toe loop
                for (int k = 0; k < 2; k++) {
                    const DevChainJoint &a = L.cl[0].B[k], &b = L.cl[1].B[k];
                    if (a.jtype != b.jtype || a.var != b.var || std::memcmp(a.plR, b.plR, sizeof(a.plR)) != 0 ||
                        std::memcmp(a.plp, b.plp, sizeof(a.plp)) != 0)
                        throw unsupported_error("Digit toe loop: the two closures do not share their anchor chain");
                }
            for (int x = 0; x < ref.ndrv; x++) L.drv_traj[x] = ind_of[ids[ref.drv[x]]];
            // rows of the loop's closures in the compiled reference order: right toe-A, right toe-B, left toe-A, left toe-B,
            // right knee, left knee (DigitDynamicsConstraints.cpp:579-596)
            L.row0 = (type == 1) ? (leg == 1 ? 0 : 6) : (leg == 1 ? 12 : 15);
            for (int r = 0; r < ref.ndep; r++) L.var_joint[r] = ids[ref.depj[r]];
            for (int x = 0; x < ref.ndrv; x++) L.var_joint[ref.ndep + x] = ids[ref.drv[x]];
            for (int r = 0; r < ref.ndep; r++) {
                const int j = ids[ref.depj[r]];
                S.loop_rec[j] = L.rec0 + r;
                S.loop_drv_a[j] = ids[ref.drv[0]];
                S.loop_drv_b[j] = ref.ndrv > 1 ? ids[ref.drv[1]] : -1;
                if (type == 1) {
                    const DigitToeGuess& g = DIGIT_TOE_GUESS[leg][r];
                    L.g_sign[r] = g.sign; L.g_c0[r] = g.c0;
                    for (int k = 0; k < 8; k++) L.g_lin[r][k] = g.lin[k];
                    for (int k = 0; k < 36; k++) L.g_quad[r][k] = 0.0;
                    for (int k = 0; k < DIGIT_TOE_NPAIR; k++) {
                        int a = DIGIT_TOE_PAIR[k][0], bq = DIGIT_TOE_PAIR[k][1];
                        if (a > bq) std::swap(a, bq);
                        L.g_quad[r][a * 8 - a * (a - 1) / 2 + (bq - a)] += g.pair[k];  // row a of the upper triangle starts at 8a - a(a-1)/2
                    }
                } else {
                    const DigitKneeGuess& g = DIGIT_KNEE_GUESS[leg][r];
                    L.g_sign[r] = g.sign; L.g_c0[r] = g.c0;
                    for (int k = 0; k < 4; k++) L.g_lin[r][k] = g.lin[k];
                }
            }
        }
    }
    if (d->gait.enabled) {  // DigitSingleStepOptimizer.cpp:97-101
        S.cust = true;
        S.cust_digit = true;
        S.stanceL = sleg == 0;
        S.swing_joint_id = d->leg_ids[1 - sleg][DG_TOE_ROLL] + 1;
        digit_stance_foot_endT(1 - sleg, S.sf_R, S.sf_p);  // DigitCustomizedConstraints.cpp:16-24
        S.gait = d->gait;
    }
    if (d->period.enabled) {  // DigitSingleStepOptimizer.cpp:102-106, JOINT_MAP of DigitSingleStepPeriodicityConstraints.h:19-32
        S.period = true;
        S.period_rows = d->period.enabled != RAPTOR_PERIOD_LINK_ONLY;
        S.period_contact_joint_id = d->leg_ids[1 - sleg][DG_TOE_ROLL] + 1;
        digit_stance_foot_endT(1 - sleg, S.period_endR, S.period_endp);  // DigitDynamicsConstraints.cpp:81-95 after the swap
        for (int leg = 0; leg < 2; leg++)
            for (int k = 0; k < 6; k++) {
                S.period_id1.push_back(d->leg_ids[leg][DIGIT_LEG_INDEPENDENT[k]]);
                S.period_id2.push_back(d->leg_ids[1 - leg][DIGIT_LEG_INDEPENDENT[k]]);
            }
        S.period_sign = 1.0;
        for (int leg = 0; leg < 2; leg++)
            for (int type = 0; type < 2; type++) {
                const DevLoop& L = D->loop[leg][type];
                LeggedSetup::PeriodLoop pl;
                pl.nvl = L.ny + L.nx; pl.nr = L.ny; pl.nx = L.nx; pl.row0 = L.row0;
                pl.drv[0] = L.var_joint[L.ny];
                pl.drv[1] = L.nx > 1 ? L.var_joint[L.ny + 1] : -1;
                for (int k = 0; k < 8; k++) pl.var_joint[k] = k < pl.nvl ? L.var_joint[k] : -1;
                S.period_loops.push_back(pl);
            }
    }
    std::unique_ptr<raptor_problem> p(new raptor_problem());
    legged_setup(*p, model, S);
    p->d_digit.upload(D.get(), 1);
    p->has_digit = true;
    finish_problem(*p);
    *out = p.release();
    API_CATCH
}

int raptor_constrained_id_eval_host(raptor_problem* p, const double* Z, int B, double* q, double* v, double* a,
                                    double* tau, double* lambda, double* ptau_pz, double* plambda_pz) {
    API_TRY
    if (!p || (B > 0 && !Z)) throw std::invalid_argument("null argument");
    order_after_async(*p, nullptr);  // null stream: after whatever the asynchronous entry points enqueued for this problem
    if (p->kind != 2) throw std::runtime_error("dcPtr_ is not defined yet!");
    if (B < 0) throw std::invalid_argument("B < 0");
    if (B == 0) return RAPTOR_OK;
    const DevLegged& Q = p->hleg;
    const int N = p->htraj.N, n = p->layout.n, nv = Q.nv, ni = Q.ni, nd = 6;  // lambda.tail(6)
    const bool deriv = ptau_pz != nullptr || plambda_pz != nullptr;
    const size_t E = (size_t)B * N;
    p->s_Z.upload(Z, (size_t)B * n);
    launch_legged_core(*p, p->s_Z.p, B, nullptr, deriv, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<double> rec(E * Q.c_stride);
    HIP_CHECK(hipMemcpy(rec.data(), p->s_core.p, rec.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t e = 0; e < E; e++) {
        const double* r = rec.data() + e * Q.c_stride;
        for (int j = 0; j < nv; j++) {
            if (q) q[e * nv + j] = r[Q.c_q + j];
            if (v) v[e * nv + j] = r[Q.c_v + j];
            if (a) a[e * nv + j] = r[Q.c_a + j];
        }
        if (tau) for (int j = 0; j < ni; j++) tau[e * ni + j] = r[Q.c_tau + j];
        if (lambda) for (int j = 0; j < nd; j++) lambda[e * nd + j] = r[Q.c_lam + j];
    }
    for (int pass = 0; pass < 2; pass++) {
        double* dst = pass == 0 ? ptau_pz : plambda_pz;
        if (!dst) continue;
        const int rows = pass == 0 ? ni : nd;
        const long long total = (long long)E * rows * n;
        p->s_J.ensure((size_t)total);
        hipLaunchKernelGGL(legged_expand_id_kernel, dim3(grid_for(total)), dim3(RAPTOR_BLOCK), 0, nullptr, p->d_traj.p,
                           p->d_leg.p, p->d_basis.p, n, p->s_core.p, total, rows, pass, p->s_J.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(dst, p->s_J.p, (size_t)total * sizeof(double), hipMemcpyDeviceToHost));
    }
    API_CATCH
}

int raptor_constrained_id_lambda_host(raptor_problem* p, const double* Z, int B, double* lambda, double* plambda_pz) {
    API_TRY
    if (!p || (B > 0 && !Z)) throw std::invalid_argument("null argument");
    order_after_async(*p, nullptr);  // null stream: after whatever the asynchronous entry points enqueued for this problem
    if (p->kind != 2) throw std::runtime_error("dcPtr_ is not defined yet!");
    if (B < 0) throw std::invalid_argument("B < 0");
    if (B == 0) return RAPTOR_OK;
    const DevLegged& Q = p->hleg;
    const int N = p->htraj.N, n = p->layout.n, ni = Q.ni, nd = Q.nd, nrec = Q.nrec;
    const bool deriv = plambda_pz != nullptr;
    const size_t E = (size_t)B * N;
    p->s_Z.upload(Z, (size_t)B * n);
    DevBuf<double> lamrec, lam, dlam;  // component view: buffers of the call
    if (nrec > 0) {
        lamrec.ensure(E * nrec * (1 + 3 * ni));
        lam.ensure(E * nd);
        if (deriv) dlam.ensure(E * ni * 3 * nd);
    }
    launch_legged_core(*p, p->s_Z.p, B, nullptr, deriv, nullptr, lamrec.p);
    if (nrec > 0) {
        if (!p->has_digit) throw unsupported_error("closure multipliers: Digit's loops only");
        const long long total = (long long)E * 2;
        const unsigned lgrid = (unsigned)((total + 63) / 64);
        hipLaunchKernelGGL((digit_lambda_kernel<false>), dim3(lgrid), dim3(64), 0, nullptr, p->d_traj.p, p->d_digit.p, p->d_basis.p, n,
                           p->s_Z.p, total, p->s_loops.p, lamrec.p, ni, nd, lam.p, deriv ? dlam.p : nullptr);
        hipLaunchKernelGGL((digit_lambda_kernel<true>), dim3(lgrid), dim3(64), 0, nullptr, p->d_traj.p, p->d_digit.p, p->d_basis.p, n,
                           p->s_Z.p, total, p->s_loops.p, lamrec.p, ni, nd, lam.p, deriv ? dlam.p : nullptr);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipDeviceSynchronize());
    if (lambda) {
        std::vector<double> rec(E * Q.c_stride), lh;
        HIP_CHECK(hipMemcpy(rec.data(), p->s_core.p, rec.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (nrec > 0) {
            lh.resize(E * nd);
            HIP_CHECK(hipMemcpy(lh.data(), lam.p, lh.size() * sizeof(double), hipMemcpyDeviceToHost));
        }
        for (size_t e = 0; e < E; e++) {
            for (int j = 0; j < nd - 6; j++) lambda[e * nd + j] = lh[e * nd + j];
            for (int j = 0; j < 6; j++) lambda[e * nd + nd - 6 + j] = rec[e * Q.c_stride + Q.c_lam + j];
        }
    }
    if (deriv) {
        const long long total = (long long)E * nd * n;
        p->s_J.ensure((size_t)total);
        hipLaunchKernelGGL(legged_lambda_expand_kernel, dim3(grid_for(total)), dim3(RAPTOR_BLOCK), 0, nullptr, p->d_traj.p, p->d_leg.p,
                           p->d_basis.p, n, p->s_core.p, dlam.p, total, p->s_J.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(plambda_pz, p->s_J.p, (size_t)total * sizeof(double), hipMemcpyDeviceToHost));
    }
    API_CATCH
}

void raptor_problem_destroy(raptor_problem* p) { delete p; }

int raptor_problem_layout(const raptor_problem* p, int* n, int* m, int* num_blocks, int* block_offsets) {
    API_TRY
    if (!p) throw std::invalid_argument("null problem");
    if (n) *n = p->layout.n;
    if (m) *m = p->layout.m;
    if (num_blocks) *num_blocks = p->num_blocks;
    if (block_offsets)
        for (int i = 0; i <= p->num_blocks; i++) block_offsets[i] = p->block_offsets[i];
    API_CATCH
}

int raptor_problem_bounds(const raptor_problem* p, double* g_lb, double* g_ub) {
    API_TRY
    if (!p) throw std::invalid_argument("null problem");
    if (g_lb) std::memcpy(g_lb, p->g_lb.data(), p->g_lb.size() * sizeof(double));
    if (g_ub) std::memcpy(g_ub, p->g_ub.data(), p->g_ub.size() * sizeof(double));
    API_CATCH
}

int raptor_problem_tspan(const raptor_problem* p, double* tspan) {
    API_TRY
    if (!p || !tspan) throw std::invalid_argument("null argument");
    std::memcpy(tspan, p->tspan.data(), p->tspan.size() * sizeof(double));
    API_CATCH
}

int raptor_eval_batch(raptor_problem* p, const double* d_Z, int B, unsigned flags, double* d_G, double* d_JAC, void* stream) {
    API_TRY
    if (!p) throw std::invalid_argument("null problem");
    eval_device(*p, d_Z, B, flags, d_G, d_JAC, (hipStream_t)stream);
    mark_async(*p, (hipStream_t)stream);
    API_CATCH
}

int raptor_shard_range(long long total, int world, int rank, long long* first, long long* count) {
    API_TRY
    if (total < 0 || world < 1 || rank < 0 || rank >= world || !first || !count) throw std::invalid_argument("bad shard arguments");
    const long long per = (total + world - 1) / world;
    const long long lo = std::min((long long)rank * per, total);
    *first = lo;
    *count = std::min(lo + per, total) - lo;
    API_CATCH
}

int raptor_violation_batch(raptor_problem* p, const double* d_G, int B, double* d_violation, void* stream) {
    API_TRY
    if (!p || (B > 0 && (!d_G || !d_violation))) throw std::invalid_argument("null argument");
    if (B < 0) throw std::invalid_argument("B < 0");
    if (B == 0) return RAPTOR_OK;
    if (!p->d_glb.p) throw std::runtime_error("bounds are not on the device");  // (uploaded by *_problem_create; nothing is allocated here)
    hipLaunchKernelGGL(violation_kernel, dim3((unsigned)B), dim3(RAPTOR_BLOCK), 0, (hipStream_t)stream, d_G, p->d_glb.p, p->d_gub.p,
                       p->layout.m, d_violation);
    HIP_CHECK(hipGetLastError());
    API_CATCH
}

// RCCL is resolved from the process at run time (never linked: a host without multi-GPU use must not need librccl):
// ncclAllGather(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t), ncclDouble == 8 (rccl.h)
namespace {
typedef int (*nccl_all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
nccl_all_gather_fn resolve_nccl_all_gather() {
    static nccl_all_gather_fn fn = nullptr;
    if (fn) return fn;
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
    if (!sym) {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // the copy the caller created its communicator with
            if (!h) continue;
            sym = dlsym(h, "ncclAllGather");
            if (sym) break;
        }
    }
    if (!sym) throw std::runtime_error("raptor_eval_batch_sharded: ncclAllGather not found in this process (load librccl.so.1 and create the communicator first)");
    fn = reinterpret_cast<nccl_all_gather_fn>(sym);
    return fn;
}
}  // namespace

int raptor_eval_batch_sharded(raptor_problem* p, const double* d_Z_all, long long total, int world, int rank, unsigned flags,
                              double* d_G, double* d_JAC, double* d_rec_all, void* nccl_comm, void* stream) {
    API_TRY
    if (!p) throw std::invalid_argument("null problem");
    if (total < 0 || world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("bad shard arguments");
    if (!nccl_comm && world > 1) throw std::invalid_argument("nccl_comm is null with more than one rank");
    if (total > 0 && (!d_Z_all || !d_rec_all)) throw std::invalid_argument("null argument");
    if (!(flags & RAPTOR_EVAL_G) || !d_G) throw std::invalid_argument("the per-problem record needs g: RAPTOR_EVAL_G and d_G");
    if (total == 0) return RAPTOR_OK;
    const long long per = (total + world - 1) / world;
    const long long lo = std::min((long long)rank * per, total), count = std::min(lo + per, total) - lo;
    if (per > 0x7fffffffLL) throw std::invalid_argument("block too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    double* mine = d_rec_all + (size_t)rank * per;
    if (count < per) HIP_CHECK(hipMemsetAsync(mine + count, 0, (size_t)(per - count) * sizeof(double), st));  // padding of a short block
    if (count > 0) {
        eval_device(*p, d_Z_all + (size_t)lo * p->layout.n, (int)count, flags, d_G, d_JAC, st);
        if (!p->d_glb.p) throw std::runtime_error("bounds are not on the device");
        hipLaunchKernelGGL(violation_kernel, dim3((unsigned)count), dim3(RAPTOR_BLOCK), 0, st, d_G, p->d_glb.p, p->d_gub.p,
                           p->layout.m, mine);
        HIP_CHECK(hipGetLastError());
    }
    if (nccl_comm) {  // in place: this rank's block already sits at recvbuff + rank * per
        const int rc = resolve_nccl_all_gather()(mine, d_rec_all, (size_t)per, /*ncclDouble*/ 8, nccl_comm, st);
        if (rc != 0) throw std::runtime_error("ncclAllGather failed with code " + std::to_string(rc));
    }
    mark_async(*p, st);
    API_CATCH
}

int raptor_eval_batch_status(raptor_problem* p, const double* d_Z, int B, unsigned flags, double* d_G, double* d_JAC, int* d_status,
                             int* d_iterations, void* stream) {
    API_TRY
    if (!p) throw std::invalid_argument("null problem");
    if (B < 0) throw std::invalid_argument("B < 0");
    if (B > 0 && !d_status) throw std::invalid_argument("status buffer is null");
    if (B == 0) return RAPTOR_OK;
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipMemsetAsync(d_status, 0, (size_t)B * sizeof(int), st));
    if (d_iterations) HIP_CHECK(hipMemsetAsync(d_iterations, 0, (size_t)B * sizeof(int), st));
    struct Scope {  // the report pointers are valid for this evaluation only
        raptor_problem* p;
        ~Scope() { p->cur_status = nullptr; p->cur_iterations = nullptr; }
    } scope{p};
    p->cur_status = d_status;
    p->cur_iterations = d_iterations;
    eval_device(*p, d_Z, B, flags, d_G, d_JAC, st);
    mark_async(*p, st);
    API_CATCH
}

int raptor_eval_batch_status_host(raptor_problem* p, const double* Z, int B, unsigned flags, double* G, double* JAC, int* status,
                                  int* iterations) {
    API_TRY
    if (!p) throw std::invalid_argument("null problem");
    if (B < 0) throw std::invalid_argument("B < 0");
    if (B > 0 && (!Z || !status)) throw std::invalid_argument("null argument");
    const bool jac = (flags & RAPTOR_EVAL_JAC) != 0, wg = (flags & RAPTOR_EVAL_G) != 0;
    if (jac && !JAC) throw std::invalid_argument("JAC buffer is null");
    if (wg && !G) throw std::invalid_argument("G buffer is null");
    if (B == 0) return RAPTOR_OK;
    const size_t n = p->layout.n, m = p->layout.m;
    hipStream_t st = p->hs.stream();
    order_after_async(*p, st);
    p->s_Z.ensure((size_t)B * n);
    if (wg) p->s_G.ensure((size_t)B * m);
    if (jac) p->s_J.ensure((size_t)B * m * n);
    p->s_status.ensure((size_t)2 * B);
    HIP_CHECK(hipMemcpyAsync(p->s_Z.p, Z, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice, st));
    const int rc = raptor_eval_batch_status(p, p->s_Z.p, B, flags, wg ? p->s_G.p : nullptr, jac ? p->s_J.p : nullptr, p->s_status.p,
                                            p->s_status.p + B, st);
    if (rc != RAPTOR_OK) return rc;
    if (wg) HIP_CHECK(hipMemcpyAsync(G, p->s_G.p, (size_t)B * m * sizeof(double), hipMemcpyDeviceToHost, st));
    if (jac) HIP_CHECK(hipMemcpyAsync(JAC, p->s_J.p, (size_t)B * m * n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(status, p->s_status.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    if (iterations) HIP_CHECK(hipMemcpyAsync(iterations, p->s_status.p + B, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    API_CATCH
}

int raptor_eval_batch_host(raptor_problem* p, const double* Z, int B, unsigned flags, double* G, double* JAC) {
    API_TRY
    if (!p) throw std::invalid_argument("null problem");
    if (B < 0) throw std::invalid_argument("B < 0");
    if (B > 0 && !Z) throw std::invalid_argument("Z is null");
    const bool jac = (flags & RAPTOR_EVAL_JAC) != 0, wg = (flags & RAPTOR_EVAL_G) != 0;
    if (jac && !JAC) throw std::invalid_argument("JAC buffer is null");
    if (wg && !G) throw std::invalid_argument("G buffer is null");
    if (B == 0) return RAPTOR_OK;
    const size_t n = p->layout.n, m = p->layout.m;
    const size_t zb = (size_t)B * n * sizeof(double), gb = wg ? (size_t)B * m * sizeof(double) : 0,
                 jb = jac ? (size_t)B * m * n * sizeof(double) : 0;
    hipStream_t st = p->hs.stream();
    order_after_async(*p, st);
    const bool pin = zb + gb + jb <= HostStage::kZeroCopyMax;
    char* h = pin ? p->hs.pinned(zb + gb + jb) : nullptr;
    if (pin) {
        // latency path (one Ipopt-driven problem): no copy engine (a DMA copy costs ~0.1 ms each on this platform); a small
        // kernel moves z in and the rows out through the pinned, device-mapped buffer with full-width accesses
        std::memcpy(h, Z, zb);
        char* dh = nullptr;
        HIP_CHECK(hipHostGetDevicePointer((void**)&dh, h, 0));
        p->s_Z.ensure((size_t)B * n);
        if (wg) p->s_G.ensure((size_t)B * m);
        if (jac) p->s_J.ensure((size_t)B * m * n);
        launch_copy2(p->s_Z.p, (const double*)dh, (size_t)B * n, nullptr, nullptr, 0, st);
        eval_device(*p, p->s_Z.p, B, flags, wg ? p->s_G.p : nullptr, jac ? p->s_J.p : nullptr, st);
        launch_copy2((double*)(dh + zb), p->s_G.p, wg ? (size_t)B * m : 0, (double*)(dh + zb + gb), p->s_J.p, jac ? (size_t)B * m * n : 0, st);
        HIP_CHECK(hipStreamSynchronize(st));
        if (wg) std::memcpy(G, h + zb, gb);
        if (jac) std::memcpy(JAC, h + zb + gb, jb);
        return RAPTOR_OK;
    }
    p->s_Z.ensure((size_t)B * n);
    if (wg) p->s_G.ensure((size_t)B * m);
    if (jac) p->s_J.ensure((size_t)B * m * n);
    // batches: z up by the copy engine, rows down through the chunked, pipelined download -- all on this call's private stream,
    // which is the only thing waited for (other streams of the process are not)
    HIP_CHECK(hipMemcpyAsync(p->s_Z.p, (const void*)Z, zb, hipMemcpyHostToDevice, st));
    eval_device(*p, p->s_Z.p, B, flags, wg ? p->s_G.p : nullptr, jac ? p->s_J.p : nullptr, st);
    if (wg) p->hs.download(G, p->s_G.p, gb);
    if (jac) p->hs.download(JAC, p->s_J.p, jb);
    API_CATCH
}

int raptor_trajectory_eval_host(raptor_problem* p, const double* Z, int B, double* q, double* q_d, double* q_dd) {
    API_TRY
    if (!p || (B > 0 && !Z)) throw std::invalid_argument("null argument");
    order_after_async(*p, nullptr);  // null stream: after whatever the asynchronous entry points enqueued for this problem
    if (B < 0) throw std::invalid_argument("B < 0");
    if (B == 0) return RAPTOR_OK;
    const size_t tot = (size_t)B * p->htraj.N * p->htraj.Nact;
    p->s_Z.upload(Z, (size_t)B * p->layout.n);
    p->s_a.ensure(tot);
    p->s_b.ensure(tot);
    p->s_c.ensure(tot);
    hipLaunchKernelGGL(traj_kernel, dim3(grid_for((long long)tot)), dim3(RAPTOR_BLOCK), 0, nullptr, p->d_traj.p,
                       p->d_basis.p, p->s_Z.p, (long long)tot, p->s_a.p, p->s_b.p, p->s_c.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    if (q) HIP_CHECK(hipMemcpy(q, p->s_a.p, tot * sizeof(double), hipMemcpyDeviceToHost));
    if (q_d) HIP_CHECK(hipMemcpy(q_d, p->s_b.p, tot * sizeof(double), hipMemcpyDeviceToHost));
    if (q_dd) HIP_CHECK(hipMemcpy(q_dd, p->s_c.p, tot * sizeof(double), hipMemcpyDeviceToHost));
    API_CATCH
}

int raptor_inverse_dynamics_eval_host(raptor_problem* p, const double* Z, int B, double* tau, double* ptau_pz) {
    API_TRY
    if (!p || (B > 0 && !Z)) throw std::invalid_argument("null argument");
    order_after_async(*p, nullptr);  // null stream: after whatever the asynchronous entry points enqueued for this problem
    if (B < 0) throw std::invalid_argument("B < 0");
    if (B == 0) return RAPTOR_OK;
    const int nv = p->hmodel.nv, N = p->htraj.N, n = p->layout.n;
    if (p->htraj.Nact != nv) throw std::invalid_argument("InverseDynamics: trajectory Nact must equal model nv");
    const size_t rows = (size_t)N * nv;
    DevLayout L{(int)rows, n, -1, -1, 0, -1};
    p->s_Z.upload(Z, (size_t)B * n);
    p->s_G.ensure((size_t)B * rows);
    if (ptau_pz) {
        p->s_J.ensure((size_t)B * rows * n);
        launch_id<true>(*p, L, p->s_Z.p, B, p->s_G.p, p->s_J.p, nullptr);
    } else {
        launch_id<false>(*p, L, p->s_Z.p, B, p->s_G.p, nullptr, nullptr);
    }
    HIP_CHECK(hipDeviceSynchronize());
    if (tau) HIP_CHECK(hipMemcpy(tau, p->s_G.p, (size_t)B * rows * sizeof(double), hipMemcpyDeviceToHost));
    if (ptau_pz) HIP_CHECK(hipMemcpy(ptau_pz, p->s_J.p, (size_t)B * rows * n * sizeof(double), hipMemcpyDeviceToHost));
    API_CATCH
}

int raptor_eval_costs(raptor_problem* p, const double* d_Z, int B, int num_costs, const int* kinds, double* d_F, double* d_GRAD,
                      void* stream) {
    API_TRY
    if (!p || !kinds || (B > 0 && (!d_Z || !d_F))) throw std::invalid_argument("null argument");
    if (B < 0 || num_costs < 0) throw std::invalid_argument("negative size");
    int traj_only = 0;
    for (int k = 0; k < num_costs; k++) {
        if (kinds[k] < 0 || kinds[k] > RAPTOR_COST_MINIMIZE_JERK) throw std::invalid_argument("unknown cost kind");
        if (kinds[k] >= RAPTOR_COST_MINIMIZE_PATH_LENGTH) traj_only++;
    }
    if (traj_only != 0 && traj_only != num_costs) throw unsupported_error("costs: path length / jerk cannot share a call with the torque-based costs");
    if (traj_only == 0 && p->kind != 2) throw std::invalid_argument("costs: the problem has no ConstrainedInverseDynamics");
    if (B == 0 || num_costs == 0) return RAPTOR_OK;
    hipStream_t st = (hipStream_t)stream;
    if (traj_only) {  // MinimizePathLength / MinimizeJerk: functions of the trajectory alone
        const size_t lds_bytes = ((size_t)p->htraj.N * p->htraj.Nact + RAPTOR_BLOCK) * sizeof(double);
        if (lds_bytes > 64 * 1024) throw unsupported_error("costs: N * Nact too large for the trajectory-cost kernel");
        if ((long long)B * num_costs > 0x7fffffffLL) throw std::invalid_argument("batch too large for one launch");
        p->d_kinds.ensure(num_costs);
        HIP_CHECK(hipMemcpyAsync(p->d_kinds.p, kinds, num_costs * sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(traj_cost_kernel, dim3((unsigned)(B * num_costs)), dim3(RAPTOR_BLOCK), lds_bytes, st, p->d_traj.p,
                           p->d_basis.p, p->d_wsum.p, d_Z, p->layout.n, num_costs, p->d_kinds.p, d_F, d_GRAD);
        HIP_CHECK(hipGetLastError());
        mark_async(*p, st);
        return RAPTOR_OK;
    }
    launch_legged_core(*p, d_Z, B, nullptr, d_GRAD != nullptr, st);
    p->d_kinds.ensure(num_costs);
    HIP_CHECK(hipMemcpyAsync(p->d_kinds.p, kinds, num_costs * sizeof(int), hipMemcpyHostToDevice, st));
    if ((long long)B * num_costs > 0x7fffffffLL) throw std::invalid_argument("batch too large for one launch");
    hipLaunchKernelGGL(legged_cost_kernel, dim3((unsigned)(B * num_costs)), dim3(RAPTOR_BLOCK), 0, st, p->d_traj.p, p->d_leg.p,
                       p->d_basis.p, p->layout.n, p->s_core.p, num_costs, p->d_kinds.p, d_F, d_GRAD);
    HIP_CHECK(hipGetLastError());
    mark_async(*p, (hipStream_t)stream);
    API_CATCH
}

int raptor_cost_hessian_host(raptor_problem* p, int kind, double* HESS) {
    API_TRY
    if (!p || !HESS) throw std::invalid_argument("null argument");
    if (kind != RAPTOR_COST_MINIMIZE_PATH_LENGTH && kind != RAPTOR_COST_MINIMIZE_JERK)
        throw unsupported_error("cost Hessian: built for MinimizePathLength / MinimizeJerk only");
    if (p->htraj.type != RAPTOR_TRAJ_PIECEWISE_BEZIER) throw unsupported_error("cost Hessian: built for PiecewiseBezierCurves problems only");
    // hess_f = (1/N) sum_i dP_i^T dP_i, dP_i = pq_pz(i+1) - pq_pz(i) (or pq_dd_pz): constant in z, a table like the basis (set-up)
    const int N = p->htraj.N, V = p->htraj.varLength, Nact = p->htraj.Nact;
    if (p->hbasis.size() < (size_t)N * 3 * V) throw unsupported_error("cost Hessian: this problem keeps no host copy of the column weights");
    const int o = (kind == RAPTOR_COST_MINIMIZE_JERK) ? 2 : 0;
    size_t it = 0;
    for (int c1 = 0; c1 < V; c1++)
        for (int c2 = c1; c2 < V; c2++, it++) {
            double s = 0.0;
            if (c1 % Nact == c2 % Nact)
                for (int i = 0; i < N - 1; i++) {
                    const double* W1 = &p->hbasis[(size_t)(i * 3 + o) * V];
                    const double* W2 = &p->hbasis[(size_t)((i + 1) * 3 + o) * V];
                    s += (W2[c1] - W1[c1]) * (W2[c2] - W1[c2]);
                }
            HESS[it] = s / N;
        }
    API_CATCH
}

int raptor_eval_costs_host(raptor_problem* p, const double* Z, int B, int num_costs, const int* kinds, double* F, double* GRAD) {
    API_TRY
    if (!p || !kinds || (B > 0 && (!Z || !F))) throw std::invalid_argument("null argument");
    order_after_async(*p, nullptr);  // null stream: after whatever the asynchronous entry points enqueued for this problem
    if (B < 0 || num_costs < 0) throw std::invalid_argument("negative size");
    if (B == 0 || num_costs == 0) return RAPTOR_OK;
    const size_t n = p->layout.n;
    p->s_Z.upload(Z, (size_t)B * n);
    p->s_G.ensure((size_t)B * num_costs);
    if (GRAD) p->s_J.ensure((size_t)B * num_costs * n);
    const int rc = raptor_eval_costs(p, p->s_Z.p, B, num_costs, kinds, p->s_G.p, GRAD ? p->s_J.p : nullptr, nullptr);
    if (rc != RAPTOR_OK) return rc;
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(F, p->s_G.p, (size_t)B * num_costs * sizeof(double), hipMemcpyDeviceToHost));
    if (GRAD) HIP_CHECK(hipMemcpy(GRAD, p->s_J.p, (size_t)B * num_costs * n * sizeof(double), hipMemcpyDeviceToHost));
    API_CATCH
}

int raptor_kinova_set_objective(raptor_problem* p, const double* q_des, int tplan_n) {
    API_TRY
    if (!p || !q_des) throw std::invalid_argument("null argument");
    if (p->kind != 1) throw std::invalid_argument("objective: not a KinovaOptimizer problem");
    if (tplan_n < 0 || tplan_n >= p->htraj.N) throw std::invalid_argument("tplan_n must be in [0, N)");
    p->d_qdes.upload(q_des, 7);
    p->tplan_n = tplan_n;
    API_CATCH
}

int raptor_kinova_eval_objective(raptor_problem* p, const double* d_Z, int B, double* d_F, double* d_GRAD, void* stream) {
    API_TRY
    if (!p || (B > 0 && (!d_Z || !d_F))) throw std::invalid_argument("null argument");
    if (B < 0) throw std::invalid_argument("B < 0");
    if (p->kind != 1 || p->tplan_n < 0) throw std::invalid_argument("objective: raptor_kinova_set_objective has not been called");
    if (B == 0) return RAPTOR_OK;
    hipLaunchKernelGGL(kinova_objective_kernel, dim3(grid_for(B)), dim3(RAPTOR_BLOCK), 0, (hipStream_t)stream, p->d_traj.p,
                       p->d_basis.p, p->d_wsum.p, d_Z, p->d_qdes.p, p->tplan_n, B, d_F, d_GRAD);
    HIP_CHECK(hipGetLastError());
    mark_async(*p, (hipStream_t)stream);
    API_CATCH
}

int raptor_kinova_eval_objective_host(raptor_problem* p, const double* Z, int B, double* F, double* GRAD) {
    API_TRY
    if (!p || (B > 0 && (!Z || !F))) throw std::invalid_argument("null argument");
    order_after_async(*p, nullptr);  // null stream: after whatever the asynchronous entry points enqueued for this problem
    if (B < 0) throw std::invalid_argument("B < 0");
    if (p->kind != 1 || p->tplan_n < 0) throw std::invalid_argument("objective: raptor_kinova_set_objective has not been called");
    if (B == 0) return RAPTOR_OK;
    p->s_Z.upload(Z, (size_t)B * 7);
    p->s_a.ensure(B);
    if (GRAD) p->s_b.ensure((size_t)B * 7);
    const int rc = raptor_kinova_eval_objective(p, p->s_Z.p, B, p->s_a.p, GRAD ? p->s_b.p : nullptr, nullptr);
    if (rc != RAPTOR_OK) return rc;
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(F, p->s_a.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost));
    if (GRAD) HIP_CHECK(hipMemcpy(GRAD, p->s_b.p, (size_t)B * 7 * sizeof(double), hipMemcpyDeviceToHost));
    API_CATCH
}

// Optimizer::eval_h for KinovaOptimizer on the device: obstacle rows and hess_f (kinova_hessian_kernel), then the torque rows added by
// kinova_hess_torque_kernel.  split: seven workgroups per z into part[B][7][28], then the fixed-order sum (few decision vectors).
static void launch_kinova_hessian(raptor_problem* p, bool split, const double* d_Z, int B, const double* d_obj_factor,
                                  const double* d_lambda, int tplan, double* part, double* d_H, hipStream_t st) {
    if (p->d_hlinks.p == nullptr) throw unsupported_error("eval_h kernel: 7-joint serial chains of revolute joints only");
    static bool attr_set = false;
    const size_t lds_bytes = (size_t)HT_TOTAL * sizeof(double);
    if (!attr_set) {  // > 64 KiB of dynamic LDS needs the opt-in
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&kinova_hess_torque_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&kinova_hess_torque_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    if (split) {
        hipLaunchKernelGGL(kinova_hessian_kernel<true>, dim3((unsigned)B, 7), dim3(RAPTOR_BLOCK), 0, st, p->d_model.p, p->d_traj.p,
                           p->d_basis.p, p->d_wsum.p, p->d_spheres.p, p->d_boxes.p, p->layout, d_Z, d_obj_factor, d_lambda, tplan, part);
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(kinova_hess_torque_kernel<true>, dim3((unsigned)B, 7), dim3(RAPTOR_BLOCK), lds_bytes, st, p->d_hlinks.p,
                           p->d_model.p, p->d_traj.p, p->d_basis.p, p->d_wsum.p, p->layout, d_Z, d_lambda, part);
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(kinova_hessian_reduce_kernel, dim3((unsigned)((B * H_PAIRS + 63) / 64)), dim3(64), 0, st, part, p->d_wsum.p,
                           d_obj_factor, tplan, B, d_H);
    } else {
        hipLaunchKernelGGL(kinova_hessian_kernel<false>, dim3((unsigned)B), dim3(RAPTOR_BLOCK), 0, st, p->d_model.p, p->d_traj.p,
                           p->d_basis.p, p->d_wsum.p, p->d_spheres.p, p->d_boxes.p, p->layout, d_Z, d_obj_factor, d_lambda, tplan, d_H);
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(kinova_hess_torque_kernel<false>, dim3((unsigned)B), dim3(RAPTOR_BLOCK), lds_bytes, st, p->d_hlinks.p,
                           p->d_model.p, p->d_traj.p, p->d_basis.p, p->d_wsum.p, p->layout, d_Z, d_lambda, d_H);
    }
    HIP_CHECK(hipGetLastError());
}

int raptor_eval_h_batch(raptor_problem* p, const double* d_Z, int B, const double* d_obj_factor, const double* d_lambda, double* d_H,
                        void* stream) {
    API_TRY
    if (!p || (B > 0 && (!d_Z || !d_lambda || !d_H))) throw std::invalid_argument("null argument");
    if (B < 0) throw std::invalid_argument("B < 0");
    if (p->kind == 3) {  // KinovaLongerHorizonOptimizer: per-sample joint-space blocks, then the contraction with the column weights
        if (d_obj_factor) throw std::invalid_argument("eval_h: this optimizer's objective is a weighted sum of Costs; add obj_factor * raptor_cost_hessian_host on the caller's side");
        if (B == 0) return RAPTOR_OK;
        hess_blocks_device(*p, d_Z, B, d_lambda, d_H, (hipStream_t)stream);
        return RAPTOR_OK;
    }
    if (p->kind != 1) throw std::runtime_error("eval_h: enable_hessian is false for this optimizer");  // Optimizer.cpp:513-515
    if (d_obj_factor && p->tplan_n < 0) throw std::invalid_argument("eval_h: raptor_kinova_set_objective has not been called");
    if (B == 0) return RAPTOR_OK;
    if (!p->hmodel.serial_chain) throw unsupported_error("eval_h kernel: serial-chain models only");
    for (int l = 0; l < p->hmodel.nv; l++)
        if (std::abs(p->hmodel.jtype[l]) > 3) throw unsupported_error("eval_h kernel: revolute joints only");
    const int tplan = p->tplan_n < 0 ? 0 : p->tplan_n;
    hipStream_t st = (hipStream_t)stream;
    static const int h_split_max = getenv("RAPTOR_H_SPLIT_MAX") ? atoi(getenv("RAPTOR_H_SPLIT_MAX")) : 192;  // measured crossover near B = 210 (round 3 kernels; 128 in round 2)
    if (B <= h_split_max) {  // few decision vectors (the Ipopt-driven case is B = 1): seven workgroups per z, then a fixed-order sum
        p->s_hpart.ensure((size_t)std::max(32, B) * 7 * H_PAIRS);
        launch_kinova_hessian(p, true, d_Z, B, d_obj_factor, d_lambda, tplan, p->s_hpart.p, d_H, st);
    } else {
        launch_kinova_hessian(p, false, d_Z, B, d_obj_factor, d_lambda, tplan, nullptr, d_H, st);
    }
    HIP_CHECK(hipGetLastError());
    mark_async(*p, (hipStream_t)stream);
    API_CATCH
}

// raptor_eval_h_batch restricted to its small-batch (split) form, with the library-owned partial-Hessian buffer replaced by the
// fixed device workspace above: no hipMalloc / hipFree on this path.
int raptor_eval_h_batch_noalloc(raptor_problem* p, const double* d_Z, int B, const double* d_obj_factor, const double* d_lambda,
                                double* d_H, void* stream) {
    API_TRY
    if (!p || (B > 0 && (!d_Z || !d_lambda || !d_H))) throw std::invalid_argument("null argument");
    if (B < 0) throw std::invalid_argument("B < 0");
    if (B > RAPTOR_H_NOALLOC_MAX_B) throw std::invalid_argument("eval_h (allocation-free): B exceeds the workspace limit of 112 decision vectors");
    if (p->kind != 1) throw std::runtime_error("eval_h (allocation-free): KinovaOptimizer (ArmourBezierCurves) problems only");
    if (d_obj_factor && p->tplan_n < 0) throw std::invalid_argument("eval_h: raptor_kinova_set_objective has not been called");
    if (B == 0) return RAPTOR_OK;
    if (!p->hmodel.serial_chain) throw unsupported_error("eval_h kernel: serial-chain models only");
    for (int l = 0; l < p->hmodel.nv; l++)
        if (std::abs(p->hmodel.jtype[l]) > 3) throw unsupported_error("eval_h kernel: revolute joints only");
    double* part = scratch_device_address();  // device address of the workspace on the current GPU (hipGetSymbolAddress, cached)
    const int tplan = p->tplan_n < 0 ? 0 : p->tplan_n;
    hipStream_t st = (hipStream_t)stream;
    // every entry part[b][k][pair], b < B, is written by the first kernel before the second adds to it: nothing has to start at zero
    launch_kinova_hessian(p, true, d_Z, B, d_obj_factor, d_lambda, tplan, part, d_H, st);
    // (no ordering event here: this entry point touches none of the problem's scratch, and a caller under stream capture must not
    // have an event of ours captured into its graph)
    API_CATCH
}

int raptor_eval_h_batch_host(raptor_problem* p, const double* Z, int B, const double* obj_factor, const double* lambda, double* H) {
    API_TRY
    if (!p || (B > 0 && (!Z || !lambda || !H))) throw std::invalid_argument("null argument");
    if (B < 0) throw std::invalid_argument("B < 0");
    if (B == 0) return RAPTOR_OK;
    const size_t n = p->layout.n, m = p->layout.m, nh = n * (n + 1) / 2;
    const size_t zb = (size_t)B * n * sizeof(double), lb = (size_t)B * m * sizeof(double), ob = obj_factor ? (size_t)B * sizeof(double) : 0,
                 hb = (size_t)B * nh * sizeof(double);
    hipStream_t st = p->hs.stream();
    order_after_async(*p, st);
    const bool pin = zb + lb + ob + hb <= HostStage::kPinMax;
    char* h = pin ? p->hs.pinned(zb + lb + ob + hb) : nullptr;
    if (pin) {
        std::memcpy(h, Z, zb);
        std::memcpy(h + zb, lambda, lb);
        if (obj_factor) std::memcpy(h + zb + lb, obj_factor, ob);
    }
    if (pin && zb + lb + ob + hb <= HostStage::kZeroCopyMax) {  // latency path: kernels on the device-mapped pinned buffer
        char* dh = nullptr;
        HIP_CHECK(hipHostGetDevicePointer((void**)&dh, h, 0));
        const int rc = raptor_eval_h_batch(p, (const double*)dh, B, obj_factor ? (const double*)(dh + zb + lb) : nullptr,
                                           (const double*)(dh + zb), (double*)(dh + zb + lb + ob), st);
        if (rc != RAPTOR_OK) return rc;
        HIP_CHECK(hipStreamSynchronize(st));
        std::memcpy(H, h + zb + lb + ob, hb);
        return RAPTOR_OK;
    }
    p->s_Z.ensure((size_t)B * n);
    p->s_G.ensure((size_t)B * m);
    if (obj_factor) p->s_a.ensure(B);
    p->s_b.ensure((size_t)B * nh);
    HIP_CHECK(hipMemcpyAsync(p->s_Z.p, pin ? (const void*)h : (const void*)Z, zb, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(p->s_G.p, pin ? (const void*)(h + zb) : (const void*)lambda, lb, hipMemcpyHostToDevice, st));
    if (obj_factor) HIP_CHECK(hipMemcpyAsync(p->s_a.p, pin ? (const void*)(h + zb + lb) : (const void*)obj_factor, ob, hipMemcpyHostToDevice, st));
    const int rc = raptor_eval_h_batch(p, p->s_Z.p, B, obj_factor ? p->s_a.p : nullptr, p->s_G.p, p->s_b.p, st);
    if (rc != RAPTOR_OK) return rc;
    HIP_CHECK(hipMemcpyAsync(pin ? (void*)(h + zb + lb + ob) : (void*)H, p->s_b.p, hb, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (pin) std::memcpy(H, h + zb + lb + ob, hb);
    API_CATCH
}

int raptor_profile_begin(raptor_problem* p, int max_evaluations) {
    API_TRY
    if (!p || max_evaluations < 0) throw std::invalid_argument("bad argument");
    while (p->ev.size() < (size_t)max_evaluations * 4) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        p->ev.push_back(e);
    }
    p->ev_used = 0;
    p->profiling = true;
    API_CATCH
}

int raptor_profile_end_split(raptor_problem* p, int* evaluations, double* ms_avg) {
    API_TRY
    if (!p) throw std::invalid_argument("null problem");
    p->profiling = false;
    const size_t n = p->ev_used / 4;
    double acc[3] = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        HIP_CHECK(hipEventSynchronize(p->ev[4 * i + 3]));
        for (int k = 0; k < 3; k++) {
            float t = 0;
            HIP_CHECK(hipEventElapsedTime(&t, p->ev[4 * i + k], p->ev[4 * i + k + 1]));
            acc[k] += t;
        }
    }
    if (evaluations) *evaluations = (int)n;
    if (ms_avg)
        for (int k = 0; k < 3; k++) ms_avg[k] = n ? acc[k] / n : 0.0;
    p->ev_used = 0;
    API_CATCH
}

int raptor_profile_end(raptor_problem* p, int* evaluations, double* id_ms_avg, double* obs_ms_avg) {
    double ms[3] = {0, 0, 0};
    const int rc = raptor_profile_end_split(p, evaluations, ms);
    if (id_ms_avg) *id_ms_avg = ms[0] + ms[1];
    if (obs_ms_avg) *obs_ms_avg = ms[2];
    return rc;
}

int raptor_capsule_distance_batch(const double* d_P, const double* d_dP, const double* d_R, int n_pairs, int F,
                                  double* d_dist, double* d_grad, void* stream) {
    API_TRY
    if (n_pairs < 0 || F < 0) throw std::invalid_argument("negative size");
    if (n_pairs == 0) return RAPTOR_OK;
    if (!d_P || !d_R || !d_dist) throw std::invalid_argument("null argument");
    if (d_dP && !d_grad) throw std::invalid_argument("grad buffer is null");
    hipLaunchKernelGGL(capsule_kernel, dim3((unsigned)((n_pairs + CAPS_BLOCK - 1) / CAPS_BLOCK)), dim3(CAPS_BLOCK), 0, (hipStream_t)stream, d_P, d_dP,
                       d_R, n_pairs, F, d_dist, d_grad);
    HIP_CHECK(hipGetLastError());
    API_CATCH
}

// *MultipleStepPeriodicityConstraints (DigitMultipleStepPeriodicityConstraints.cpp:57-230, Talos / G1 twins)
static void step_link_device(raptor_problem& pc, raptor_problem& pn, const double* dZc, const double* dZn, int B, unsigned flags,
                             double* dG, double* dJ, double* dJ2, hipStream_t st) {
    if (pc.kind != 2 || pn.kind != 2 || !pc.has_period)
        throw std::invalid_argument("step link: legged problems, the current step's created with its reset-map block");
    if (pc.layout.n != pn.layout.n || pc.htraj.Nact != pn.htraj.Nact || pc.htraj.K != pn.htraj.K)
        throw std::invalid_argument("currTrajPtr_ and nextTrajPtr_ should have the same varLength");
    if (!(flags & (RAPTOR_EVAL_G | RAPTOR_EVAL_JAC))) throw std::invalid_argument("flags: nothing to evaluate");
    const bool jac = (flags & RAPTOR_EVAL_JAC) != 0;
    if (jac && (dJ == nullptr || dJ2 == nullptr)) throw std::invalid_argument("JAC buffer is null");
    if ((flags & RAPTOR_EVAL_G) && dG == nullptr) throw std::invalid_argument("G buffer is null");
    if (B < 0) throw std::invalid_argument("negative batch size");
    if (B == 0) return;
    if (dZc == nullptr || dZn == nullptr) throw std::invalid_argument("Z is null");
    check_launchable(pc, B);
    launch_legged_core(pc, dZc, B, nullptr, jac, st);  // records only: the step's own rows belong to raptor_eval_batch
    launch_period(pc, dZc, B, (flags & RAPTOR_EVAL_G) ? dG : nullptr, jac ? dJ : nullptr, jac, st, pc.hleg.nperiod, 0, &pn, dZn,
                  jac ? dJ2 : nullptr);
}

// DualKinovaOptimizer.cpp:269-305, 449-510
static void dual_arm_arm_device(raptor_problem& p1, raptor_problem& p2, const double* dZ, int B, unsigned flags, double* dG,
                                double* dJ, hipStream_t st) {
    if ((p1.kind != 1 && p1.kind != 3) || (p2.kind != 1 && p2.kind != 3))
        throw std::invalid_argument("dual Kinova: both problems must be Kinova problems");
    if (p1.htraj.N != p2.htraj.N) throw std::invalid_argument("dual Kinova: the two trajectories must have the same N");
    if (!(flags & (RAPTOR_EVAL_G | RAPTOR_EVAL_JAC))) throw std::invalid_argument("flags: nothing to evaluate");
    const bool jac = (flags & RAPTOR_EVAL_JAC) != 0;
    if (jac && dJ == nullptr) throw std::invalid_argument("JAC buffer is null");
    if ((flags & RAPTOR_EVAL_G) && dG == nullptr) throw std::invalid_argument("G buffer is null");
    if (B < 0) throw std::invalid_argument("negative batch size");
    if (B == 0) return;
    if (dZ == nullptr) throw std::invalid_argument("Z is null");
    DualCapsules caps;
    raptor_problem* pr[2] = {&p1, &p2};
    for (int r = 0; r < 2; r++) {  // TAPERED_CAPSULES[1], [2] (KinovaCustomizedConstraints.h:58-63); the gripper extends the last one (.cpp:46)
        if (pr[r]->hspheres.num_spheres < 17) throw std::invalid_argument("dual Kinova: the sphere table has no tapered capsules");
        caps.ids[r][0][0] = 7;  caps.ids[r][0][1] = 11;
        caps.ids[r][1][0] = 12; caps.ids[r][1][1] = pr[r]->hspheres.num_spheres >= 20 ? 19 : 13;
    }
    const long long total = (long long)B * p1.htraj.N * 4;
    const int F = p1.layout.n + p2.layout.n;
    // pairs per pass: bounds the staging area (12 + 12 F + 4 doubles per pair; 32768 pairs at F = 14, fewer for wider problems)
    const int CH = std::max(256, std::min(32768, (int)((32768LL * 12 * 14) / (12LL * F))));
    p1.s_a.ensure((size_t)CH * 12);
    if (jac) p1.s_b.ensure((size_t)CH * 12 * F);
    p1.s_c.ensure((size_t)CH * 5);
    double* dR = p1.s_c.p;
    double* dDist = p1.s_c.p + (size_t)CH * 4;
    for (long long p0 = 0; p0 < total; p0 += CH) {
        const int np = (int)std::min<long long>(CH, total - p0);
        hipLaunchKernelGGL(dual_kinova_points_kernel, dim3((np + 63) / 64), dim3(64), 0, st, p1.d_model.p, p1.d_traj.p, p1.d_basis.p,
                           p1.d_wsum.p, p1.d_spheres.p, p2.d_model.p, p2.d_traj.p, p2.d_basis.p, p2.d_wsum.p, p2.d_spheres.p, caps, dZ,
                           p0, np, p1.s_a.p, jac ? p1.s_b.p : nullptr, dR);
        hipLaunchKernelGGL(capsule_kernel, dim3((unsigned)((np + CAPS_BLOCK - 1) / CAPS_BLOCK)), dim3(CAPS_BLOCK), 0, st, p1.s_a.p, jac ? p1.s_b.p : nullptr, dR, np, F,
                           (flags & RAPTOR_EVAL_G) ? dG + p0 : dDist, jac ? dJ + (size_t)p0 * F : nullptr);
        HIP_CHECK(hipGetLastError());
    }
}

int raptor_step_link_eval_batch(raptor_problem* curr, raptor_problem* next, const double* d_Zcurr, const double* d_Znext, int B,
                                unsigned flags, double* d_G, double* d_JAC, double* d_JAC2, void* stream) {
    API_TRY
    if (!curr || !next) throw std::invalid_argument("null argument");
    step_link_device(*curr, *next, d_Zcurr, d_Znext, B, flags, d_G, d_JAC, d_JAC2, (hipStream_t)stream);
    mark_async(*curr, (hipStream_t)stream);
    API_CATCH
}

int raptor_step_link_eval_batch_host(raptor_problem* curr, raptor_problem* next, const double* Zcurr, const double* Znext, int B,
                                     unsigned flags, double* G, double* JAC, double* JAC2) {
    API_TRY
    if (!curr || !next) throw std::invalid_argument("null argument");
    order_after_async(*curr, nullptr);
    if (next) order_after_async(*next, nullptr);
    if (B < 0) throw std::invalid_argument("negative batch size");
    if (B > 0 && (!Zcurr || !Znext)) throw std::invalid_argument("Z is null");
    const bool jac = (flags & RAPTOR_EVAL_JAC) != 0;
    if (jac && (!JAC || !JAC2)) throw std::invalid_argument("JAC buffer is null");
    if ((flags & RAPTOR_EVAL_G) && !G) throw std::invalid_argument("G buffer is null");
    if (B == 0) return RAPTOR_OK;
    const size_t n = curr->layout.n, mp = curr->hleg.nperiod, r2 = 2 * (size_t)curr->hleg.ni;
    curr->s_Z.ensure((size_t)B * n * 2);
    curr->s_G.ensure((size_t)B * mp);
    if (jac) curr->s_J.ensure((size_t)B * (mp + r2) * n);
    HIP_CHECK(hipMemcpy(curr->s_Z.p, Zcurr, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(curr->s_Z.p + (size_t)B * n, Znext, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice));
    double* dJ = jac ? curr->s_J.p : nullptr;
    double* dJ2 = jac ? curr->s_J.p + (size_t)B * mp * n : nullptr;
    step_link_device(*curr, *next, curr->s_Z.p, curr->s_Z.p + (size_t)B * n, B, flags, curr->s_G.p, dJ, dJ2, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    if (flags & RAPTOR_EVAL_G) HIP_CHECK(hipMemcpy(G, curr->s_G.p, (size_t)B * mp * sizeof(double), hipMemcpyDeviceToHost));
    if (jac) {
        HIP_CHECK(hipMemcpy(JAC, dJ, (size_t)B * mp * n * sizeof(double), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(JAC2, dJ2, (size_t)B * r2 * n * sizeof(double), hipMemcpyDeviceToHost));
    }
    API_CATCH
}

int raptor_dual_kinova_arm_arm_batch(raptor_problem* p1, raptor_problem* p2, const double* d_Z, int B, unsigned flags,
                                     double* d_G, double* d_JAC, void* stream) {
    API_TRY
    if (!p1 || !p2) throw std::invalid_argument("null argument");
    dual_arm_arm_device(*p1, *p2, d_Z, B, flags, d_G, d_JAC, (hipStream_t)stream);
    mark_async(*p1, (hipStream_t)stream);
    API_CATCH
}

int raptor_dual_kinova_arm_arm_batch_host(raptor_problem* p1, raptor_problem* p2, const double* Z, int B, unsigned flags,
                                          double* G, double* JAC) {
    API_TRY
    if (!p1 || !p2) throw std::invalid_argument("null argument");
    order_after_async(*p1, nullptr);
    if (p2) order_after_async(*p2, nullptr);
    if (B < 0) throw std::invalid_argument("negative batch size");
    if (B > 0 && !Z) throw std::invalid_argument("Z is null");
    const bool jac = (flags & RAPTOR_EVAL_JAC) != 0;
    if (jac && !JAC) throw std::invalid_argument("JAC buffer is null");
    if ((flags & RAPTOR_EVAL_G) && !G) throw std::invalid_argument("G buffer is null");
    if (B == 0) return RAPTOR_OK;
    const size_t rows = (size_t)p1->htraj.N * 4;
    const size_t F = (size_t)p1->layout.n + p2->layout.n;  // numVars1 + numVars2
    p1->s_Z.ensure((size_t)B * F);
    p1->s_G.ensure((size_t)B * rows);
    if (jac) p1->s_J.ensure((size_t)B * rows * F);
    HIP_CHECK(hipMemcpy(p1->s_Z.p, Z, (size_t)B * F * sizeof(double), hipMemcpyHostToDevice));
    dual_arm_arm_device(*p1, *p2, p1->s_Z.p, B, flags, p1->s_G.p, jac ? p1->s_J.p : nullptr, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    if (flags & RAPTOR_EVAL_G) HIP_CHECK(hipMemcpy(G, p1->s_G.p, (size_t)B * rows * sizeof(double), hipMemcpyDeviceToHost));
    if (jac) HIP_CHECK(hipMemcpy(JAC, p1->s_J.p, (size_t)B * rows * F * sizeof(double), hipMemcpyDeviceToHost));
    API_CATCH
}

int raptor_capsule_distance_batch_host(const double* P, const double* dP, const double* R, int n_pairs, int F,
                                       double* dist, double* grad) {
    API_TRY
    if (n_pairs < 0 || F < 0) throw std::invalid_argument("negative size");
    if (n_pairs == 0) return RAPTOR_OK;
    if (!P || !R || !dist) throw std::invalid_argument("null argument");
    if (dP && !grad) throw std::invalid_argument("grad buffer is null");
    DevBuf<double> bP, bdP, bR, bd, bg;
    bP.upload(P, (size_t)n_pairs * 12);
    bR.upload(R, (size_t)n_pairs * 4);
    bd.ensure(n_pairs);
    if (dP) {
        bdP.upload(dP, (size_t)n_pairs * 12 * F);
        bg.ensure((size_t)n_pairs * std::max(F, 1));
    }
    hipLaunchKernelGGL(capsule_kernel, dim3((unsigned)((n_pairs + CAPS_BLOCK - 1) / CAPS_BLOCK)), dim3(CAPS_BLOCK), 0, nullptr, bP.p, dP ? bdP.p : nullptr,
                       bR.p, n_pairs, F, bd.p, dP ? bg.p : nullptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(dist, bd.p, (size_t)n_pairs * sizeof(double), hipMemcpyDeviceToHost));
    if (dP && F > 0) HIP_CHECK(hipMemcpy(grad, bg.p, (size_t)n_pairs * F * sizeof(double), hipMemcpyDeviceToHost));
    API_CATCH
}

}  // extern "C"
