// A compiled C++17 caller of libraptor_hip.so: the adapter of INTEGRATION.md section 2 behind RAPTOR's own plug-in interface,
// driven by the loops of Optimizer::eval_g / eval_jac_g.  No Eigen, no Pinocchio (absent from this image): `Constraints` below has
// the reference's members by name (Constraints/include/Constraints.h:26-57: m, g, pg_pz, g_lb, g_ub, current_z, compute,
// compute_bounds, is_computed, initialize_memory) on std::vector storage, `MiniOptimizer` the reference's loops over
// constraintsPtrVec_ / constraintsNameVec_ (Optimization/src/Optimizer.cpp:361-385, 467-486), feasibility test per block included.
//
// The four blocks of KinovaOptimizer::set_parameters (KinovaOptimizer.cpp:82-109) keep their names: one fused evaluator owns the
// library handle, four thin `Constraints` views ("joint limits", "velocity limits", "torque limits", "obstacle avoidance
// constraints") share it and expose their slice -- so Optimizer's per-block feasibility report (Optimizer.cpp:374-378) and
// summarize_constraints name the same four blocks as the reference.
//
//   usage: kinova_host_test <input.bin> <output.bin>
//   input : written by tests/test_native_host.py (model, trajectory, boxes, z) ; output: g[m], values[m * n], per-block report
// Built by __graft_entry__.build():  g++ -std=c++17 -O2 -I include tests/native/kinova_host_test.cpp -L raptor_amd -lraptor_hip
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "raptor_hip.h"

namespace RAPTOR {

using VecX = std::vector<double>;
// row-major dense matrix with the two members the Optimizer's loops use
struct MatX {
    int rows_ = 0, cols_ = 0;
    std::vector<double> d;
    void resize(int r, int c) { rows_ = r; cols_ = c; d.assign((size_t)r * c, 0.0); }
    double operator()(int i, int j) const { return d[(size_t)i * cols_ + j]; }
    double& operator()(int i, int j) { return d[(size_t)i * cols_ + j]; }
};

// Constraints/include/Constraints.h:11-58
class Constraints {
public:
    virtual ~Constraints() = default;
    virtual void compute(const VecX& z, bool compute_derivatives = true, bool compute_hessian = false) = 0;
    virtual void compute_bounds() = 0;
    virtual void print_violation_info() {}
    // Constraints.cpp:10-36: exact comparison with the last decision vector
    bool is_computed(const VecX& z, bool compute_derivatives, bool compute_hessian) {
        if (compute_hessian && !compute_derivatives)
            throw std::invalid_argument("compute_derivatives needs to be true when compute_hessian is true.");
        const bool same_z = current_z.size() == z.size() && std::equal(z.begin(), z.end(), current_z.begin());  // ifTwoVectorEqual(., ., 0)
        if (!same_z || compute_derivatives != if_compute_derivatives) {
            current_z = z;
            if_compute_derivatives = compute_derivatives;
            if_compute_hessian = compute_hessian;
            return false;
        }
        if_compute_derivatives = compute_derivatives;
        if_compute_hessian = compute_hessian;
        return true;
    }
    void initialize_memory(const int m_input, const int varLength, bool /*initialize_hessian*/ = true) {
        m = m_input;
        g.assign(m, 0.0);
        g_lb.assign(m, 0.0);
        g_ub.assign(m, 0.0);
        pg_pz.resize(m, varLength);
    }
    int m = 0;
    VecX current_z;
    bool if_compute_derivatives = false, if_compute_hessian = false;
    VecX g;
    MatX pg_pz;
    VecX g_lb, g_ub;
};

static void check(int code) {
    if (code == RAPTOR_OK) return;
    if (code == RAPTOR_ERR_INVALID_ARGUMENT) throw std::invalid_argument(raptor_last_error());
    throw std::runtime_error(raptor_last_error());  // Optimizer re-throws as IpoptException (Optimizer.cpp:368-372)
}

// owns the library objects; evaluates all four blocks in one call, memoised on (z, flags) like every reference class
class KinovaHipEvaluator {
public:
    KinovaHipEvaluator(const raptor_model_desc& md, const raptor_kinova_desc& kd) {
        check(raptor_model_create(&md, &model_));
        check(raptor_kinova_problem_create(model_, &kd, &prob_));
        int nb = 0;
        check(raptor_problem_layout(prob_, &n, &m, &nb, offsets_));
        offsets_[nb] = m;
        num_blocks = nb;
        g.assign(m, 0.0);
        jac.assign((size_t)m * n, 0.0);
        g_lb.assign(m, 0.0);
        g_ub.assign(m, 0.0);
        check(raptor_problem_bounds(prob_, g_lb.data(), g_ub.data()));
    }
    ~KinovaHipEvaluator() {
        raptor_problem_destroy(prob_);
        raptor_model_destroy(model_);
    }
    void compute(const VecX& z, bool deriv) {
        if ((int)z.size() != n) throw std::invalid_argument("z has the wrong length");
        if (have_ && z == z_ && (!deriv || deriv_)) return;
        const unsigned flags = RAPTOR_EVAL_G | (deriv ? RAPTOR_EVAL_JAC : 0u);
        check(raptor_eval_batch_host(prob_, z.data(), 1, flags, g.data(), deriv ? jac.data() : nullptr));
        z_ = z;
        deriv_ = deriv;
        have_ = true;
        evaluations++;
    }
    int block_begin(int b) const { return offsets_[b]; }
    int block_end(int b) const { return offsets_[b + 1]; }
    int n = 0, m = 0, num_blocks = 0, evaluations = 0;
    VecX g, jac, g_lb, g_ub;

private:
    raptor_model* model_ = nullptr;
    raptor_problem* prob_ = nullptr;
    int offsets_[9] = {0};
    VecX z_;
    bool deriv_ = false, have_ = false;
};

// one of the reference's four constraint objects, as a view of the fused evaluation
class KinovaHipBlock : public Constraints {
public:
    KinovaHipBlock(std::shared_ptr<KinovaHipEvaluator> ev, int block) : ev_(std::move(ev)), r0_(ev_->block_begin(block)) {
        initialize_memory(ev_->block_end(block) - r0_, ev_->n, false);
    }
    void compute(const VecX& z, bool compute_derivatives = true, bool compute_hessian = false) override {
        if (compute_hessian) throw std::invalid_argument("per-row Hessians are not materialised (INTEGRATION.md 2a)");
        if (is_computed(z, compute_derivatives, compute_hessian)) return;
        ev_->compute(z, compute_derivatives);
        std::memcpy(g.data(), ev_->g.data() + r0_, (size_t)m * sizeof(double));
        if (compute_derivatives) std::memcpy(pg_pz.d.data(), ev_->jac.data() + (size_t)r0_ * ev_->n, (size_t)m * ev_->n * sizeof(double));
    }
    void compute_bounds() override {
        std::memcpy(g_lb.data(), ev_->g_lb.data() + r0_, (size_t)m * sizeof(double));
        std::memcpy(g_ub.data(), ev_->g_ub.data() + r0_, (size_t)m * sizeof(double));
    }

private:
    std::shared_ptr<KinovaHipEvaluator> ev_;
    int r0_;
};

// the loops of Optimizer::eval_g / eval_jac_g (Optimization/src/Optimizer.cpp:341-385, 433-486) and get_bounds_info (:110-152)
class MiniOptimizer {
public:
    std::vector<std::unique_ptr<Constraints>> constraintsPtrVec_;
    std::vector<std::string> constraintsNameVec_;
    double constr_viol_tol = 1e-4;
    bool ifFeasibleCurrIter = true;
    std::vector<int> infeasible_blocks;
    int numCons = 0;

    void get_bounds_info(VecX& g_l, VecX& g_u) {
        numCons = 0;
        for (auto& c : constraintsPtrVec_) numCons += c->m;
        g_l.assign(numCons, 0.0);
        g_u.assign(numCons, 0.0);
        int iter = 0;
        for (auto& c : constraintsPtrVec_) {
            c->compute_bounds();
            for (int i = 0; i < c->m; i++) {
                g_l[iter] = c->g_lb[i];
                g_u[iter] = c->g_ub[i];
                iter++;
            }
        }
    }
    bool eval_g(const VecX& z, VecX& g) {
        int iter = 0;
        ifFeasibleCurrIter = true;
        infeasible_blocks.clear();
        for (size_t c = 0; c < constraintsPtrVec_.size(); c++) {
            try {
                constraintsPtrVec_[c]->compute(z, false);
            } catch (std::exception& e) {
                std::fprintf(stderr, "Error in %s:\n%s\n", constraintsNameVec_[c].c_str(), e.what());
                throw;
            }
            const Constraints& K = *constraintsPtrVec_[c];
            double lo = INFINITY, hi = INFINITY;
            for (int i = 0; i < K.m; i++) {
                lo = std::fmin(lo, K.g[i] - K.g_lb[i]);
                hi = std::fmin(hi, K.g_ub[i] - K.g[i]);
            }
            if (lo < -constr_viol_tol || hi < -constr_viol_tol) {
                ifFeasibleCurrIter = false;
                infeasible_blocks.push_back((int)c);
            }
            for (int i = 0; i < K.m; i++) g[iter++] = K.g[i];
        }
        return true;
    }
    bool eval_jac_g(const VecX& z, VecX& values) {
        int iter = 0;
        for (size_t c = 0; c < constraintsPtrVec_.size(); c++) {
            constraintsPtrVec_[c]->compute(z, true);
            const Constraints& K = *constraintsPtrVec_[c];
            for (int i = 0; i < K.pg_pz.rows_; i++)
                for (int j = 0; j < K.pg_pz.cols_; j++) values[iter++] = K.pg_pz(i, j);
        }
        return true;
    }
};

}  // namespace RAPTOR

namespace {
struct Reader {
    FILE* f;
    template <class T>
    void get(T* p, size_t n) {
        if (n && std::fread(p, sizeof(T), n, f) != n) throw std::runtime_error("short input file");
    }
    int i32() { int v; get(&v, 1); return v; }
    double f64() { double v; get(&v, 1); return v; }
    std::vector<double> vec(size_t n) { std::vector<double> v(n); get(v.data(), n); return v; }
    std::vector<int> ivec(size_t n) { std::vector<int> v(n); get(v.data(), n); return v; }
};
}  // namespace

int main(int argc, char** argv) {
    if (argc != 3) {
        std::fprintf(stderr, "usage: %s <input.bin> <output.bin>\n", argv[0]);
        return 2;
    }
    try {
        FILE* fi = std::fopen(argv[1], "rb");
        if (!fi) throw std::runtime_error("cannot open input");
        Reader R{fi};
        // ---- what KinovaExample.cpp builds: model, ArmourTrajectoryParameters, boxes, z ----
        const int nv = R.i32();
        std::vector<int> parent = R.ivec(nv), jtype = R.ivec(nv);
        std::vector<double> plR = R.vec(9 * nv), plp = R.vec(3 * nv), mass = R.vec(nv), lever = R.vec(3 * nv), inertia = R.vec(9 * nv);
        std::vector<double> gravity = R.vec(3), armature = R.vec(nv), damping = R.vec(nv), friction = R.vec(nv);
        const double T = R.f64();
        const int N = R.i32();
        std::vector<double> q0 = R.vec(nv), qd0 = R.vec(nv), qdd0 = R.vec(nv);
        const int O = R.i32();
        std::vector<double> bc = R.vec(3 * O), br = R.vec(3 * O), bs = R.vec(3 * O);
        const int gripper = R.i32();
        const double buffer = R.f64();
        const int nz = R.i32();
        std::vector<std::vector<double>> Z(nz);
        for (auto& z : Z) z = R.vec(nv);
        std::fclose(fi);

        if (raptor_init(0) != RAPTOR_OK) throw std::runtime_error(raptor_last_error());
        raptor_model_desc md{};
        md.nv = nv; md.parent = parent.data(); md.jtype = jtype.data();
        md.placement_R = plR.data(); md.placement_p = plp.data(); md.mass = mass.data(); md.lever = lever.data(); md.inertia = inertia.data();
        for (int k = 0; k < 3; k++) md.gravity[k] = gravity[k];
        md.armature = armature.data(); md.damping = damping.data(); md.friction = friction.data();
        raptor_kinova_desc kd{};
        kd.traj.type = RAPTOR_TRAJ_ARMOUR_BEZIER; kd.traj.degree = 5; kd.traj.T = T; kd.traj.N = N;
        kd.traj.time_discretization = RAPTOR_TIME_CHEBYSHEV; kd.traj.Nact = nv;
        kd.traj.q0 = q0.data(); kd.traj.q_d0 = qd0.data(); kd.traj.q_dd0 = qdd0.data();
        kd.num_obstacles = O; kd.box_center = bc.data(); kd.box_rpy = br.data(); kd.box_size = bs.data();
        kd.include_gripper = gripper; kd.collision_buffer = buffer;

        // ---- KinovaOptimizer::set_parameters (KinovaOptimizer.cpp:82-109): four named constraint objects ----
        auto ev = std::make_shared<RAPTOR::KinovaHipEvaluator>(md, kd);
        if (ev->num_blocks != 4) throw std::runtime_error("expected the four Kinova blocks");
        RAPTOR::MiniOptimizer opt;
        const char* names[4] = {"joint limits", "velocity limits", "torque limits", "obstacle avoidance constraints"};
        for (int b = 0; b < 4; b++) {
            opt.constraintsPtrVec_.push_back(std::make_unique<RAPTOR::KinovaHipBlock>(ev, b));
            opt.constraintsNameVec_.push_back(names[b]);
        }
        RAPTOR::VecX g_l, g_u;
        opt.get_bounds_info(g_l, g_u);
        const int m = opt.numCons, n = ev->n;

        FILE* fo = std::fopen(argv[2], "wb");
        if (!fo) throw std::runtime_error("cannot open output");
        std::fwrite(&m, sizeof(int), 1, fo);
        std::fwrite(&n, sizeof(int), 1, fo);
        std::fwrite(g_l.data(), sizeof(double), m, fo);
        std::fwrite(g_u.data(), sizeof(double), m, fo);
        RAPTOR::VecX g(m), values((size_t)m * n);
        for (int k = 0; k < nz; k++) {
            // Ipopt's order at one iterate: eval_g (no derivatives), then eval_jac_g at the same x (new_x = false)
            opt.eval_g(Z[k], g);
            const int evals_after_g = ev->evaluations;
            opt.eval_jac_g(Z[k], values);
            const int evals_after_jac = ev->evaluations;
            std::fwrite(g.data(), sizeof(double), m, fo);
            std::fwrite(values.data(), sizeof(double), (size_t)m * n, fo);
            // the report: feasible flag, one flag per named block, and the library calls this iterate cost (1 for g, 1 for jac)
            int rep[7] = {opt.ifFeasibleCurrIter ? 1 : 0, 0, 0, 0, 0, evals_after_g, evals_after_jac};
            for (int b : opt.infeasible_blocks) rep[1 + b] = 1;
            std::fwrite(rep, sizeof(int), 7, fo);
            std::printf("z %d: %s", k, opt.ifFeasibleCurrIter ? "feasible" : "infeasible:");
            for (int b : opt.infeasible_blocks) std::printf(" [%s]", opt.constraintsNameVec_[b].c_str());
            std::printf("\n");
        }
        std::fclose(fo);
        // error behaviour across the boundary: a wrong-sized z is std::invalid_argument, as in the reference's own checks
        bool threw = false;
        try {
            RAPTOR::VecX bad(n + 1, 0.0);
            opt.constraintsPtrVec_[0]->compute(bad, false);
        } catch (const std::invalid_argument&) {
            threw = true;
        } catch (const std::exception&) {
        }
        if (!threw) throw std::runtime_error("a wrong-sized z was not refused with std::invalid_argument");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "kinova_host_test: %s\n", e.what());
        return 1;
    }
    return 0;
}
