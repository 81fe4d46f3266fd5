"""Host Optimizer shell (raptor_amd/optimizer.py) against the reference's TNLP semantics
(Optimization/src/Optimizer.cpp).  CPU: bookkeeping and a short solve with the oracle library behind the callbacks
(the oracle is the checker here: the same shell runs over the HIP library in the GPU test below)."""
import numpy as np
import pytest

from raptor_amd import _capi as A
from raptor_amd import optimizer as O

import helpers as H


def kinova_optimizer(api, kinova, keep, N=8, boxes=None, x0=None, q_des=None):
    mh = api.model_create(kinova)
    keep.append(mh)
    c, rpy, size = boxes if boxes is not None else H.example_boxes()
    z = np.zeros(7)
    opt = O.KinovaOptimizer(api, mh, x0 if x0 is not None else z, 1.0, N, (z, z, z), c, rpy, size,
                            q_des if q_des is not None else np.full(7, 0.3), N // 2)
    return opt


def test_nlp_info_bounds_and_structure(oracle, kinova):
    keep = []
    opt = kinova_optimizer(oracle, kinova, keep)
    n, m, nnz_jac, nnz_h = opt.get_nlp_info()
    assert (n, m, nnz_jac, nnz_h) == (7, 8 * 38, 7 * 8 * 38, 28)  # KinovaOptimizer.cpp:121-147
    with pytest.raises(O.IpoptException):
        opt.get_bounds_info(n + 1, m)
    x_l, x_u, g_l, g_u = opt.get_bounds_info(n, m)
    assert np.all(x_l == -1e19) and np.all(x_u == 1e19) and g_l.shape == (m,)
    assert np.array_equal(opt.get_starting_point(n), np.zeros(7))
    with pytest.raises(O.IpoptException):
        opt.get_starting_point(n, init_lambda=True)
    iRow, jCol = opt.eval_jac_g(n, None, True, m, values=False)  # Optimizer.cpp:451-460: iRow[i * n + j] = i, jCol = j
    assert iRow[9] == 1 and jCol[9] == 2 and iRow.size == nnz_jac
    iRow, jCol = opt.eval_h(n, None, True, 1.0, m, None, True, values=False)  # :531-538
    assert list(zip(iRow[:8], jCol[:8])) == [(0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6), (1, 1)] and iRow.size == 28


def test_callbacks_match_the_library(oracle, kinova):
    keep = []
    opt = kinova_optimizer(oracle, kinova, keep, boxes=H.reach_boxes())
    n, m, _, _ = opt.get_nlp_info()
    opt.get_bounds_info(n, m)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, 7)
    lam = rng.uniform(-1, 1, m)
    G, J = opt.problem.eval_host(x)
    assert np.array_equal(opt.eval_g(n, x, True, m), G[0])
    assert np.array_equal(opt.eval_jac_g(n, x, False, m), J[0].reshape(-1))  # row-major values (:480-485)
    F, GR = opt.problem.objective_host(x)
    assert opt.eval_f(n, x, False) == F[0] and np.array_equal(opt.eval_grad_f(n, x, False), GR[0])
    assert np.array_equal(opt.eval_h(n, x, False, 0.7, m, lam, True), opt.problem.eval_h_host(x, lam, 0.7)[0])
    assert opt.nlp_count == dict(f=1, g=1, grad_f=1, jac_g=1, hess_l=1)
    opt.enable_hessian = False
    assert opt.eval_h(n, x, False, 0.7, m, lam, True) is False  # :513-515


def test_best_feasible_iterate_tracking(oracle, kinova):
    """update_minimal_cost_solution / eval_g bookkeeping and finalize_solution's fallback (Optimizer.cpp:154-196, 387-421, 675-704)."""
    keep = []
    opt = kinova_optimizer(oracle, kinova, keep, q_des=np.full(7, 0.5))
    n, m, _, _ = opt.get_nlp_info()
    opt.get_bounds_info(n, m)
    feas = np.full(7, 0.2)      # slow motion, far boxes: feasible
    infeas = np.full(7, 40.0)   # breaks the velocity limits
    g = opt.eval_g(n, feas, True, m)
    assert opt.ifCurrentIpoptFeasible == O.FEASIBLE and opt.currentIpoptObjValue > 1e300
    f_feas = opt.eval_f(n, feas, False)
    assert opt.optimalIpoptObjValue == f_feas and np.array_equal(opt.optimalIpoptSolution, feas)
    opt.eval_g(n, infeas, True, m)
    f_inf = opt.eval_f(n, infeas, False)
    assert opt.ifCurrentIpoptFeasible == O.INFEASIBLE and opt.optimalIpoptObjValue == f_feas
    # eval_f first on a new x leaves feasibility unknown until eval_g runs on the same x
    better = np.full(7, 0.4)
    f_b = opt.eval_f(n, better, True)
    assert opt.ifCurrentIpoptFeasible == O.UNINITIALIZED and opt.optimalIpoptObjValue == f_feas
    opt.eval_g(n, better, False, m)
    assert f_b < f_feas and opt.optimalIpoptObjValue == f_b
    # the solver "returns" the infeasible point: finalize_solution falls back to the recorded best feasible iterate
    g_inf = opt.eval_g(n, infeas, True, m)
    opt.finalize_solution(n, infeas, m, g_inf, f_inf)
    assert opt.ifFeasible and np.array_equal(opt.solution, better) and opt.obj_value_copy == f_b
    assert opt.final_constr_violation <= opt.constr_viol_tol
    assert opt.ifOptimalIpoptFeasible == O.UNINITIALIZED and opt.nlp_count["g"] == 0  # cleared (:708-766)
    # summarize_constraints names the violated block
    report = opt.summarize_constraints(m, g_inf)
    lb, ub = opt.problem.bounds()
    assert [r[0] for r in report][:2] == ["joint limits", "velocity limits"]  # 40 rad leaves the joint range too
    assert opt.final_constr_violation == pytest.approx(float(np.fmax(lb - g_inf, g_inf - ub).max()))
    assert not opt.ifFeasible


def test_single_step_costs_are_weighted_sums(oracle):
    keep = []
    p = H.legged_problem(oracle, "g1", keep, N=4)[0]
    names = ["torque limits", "joint limits", "contact constraints"]
    opt = O.SingleStepOptimizer(p, np.zeros(p.n), names)
    n, m, _, _ = opt.get_nlp_info()
    opt.get_bounds_info(n, m)
    x = 0.1 * np.random.default_rng(3).uniform(-1, 1, n)
    F, GR = p.costs_host(x, [A.COST_MINIMIZE_POWER, A.COST_MINIMIZE_INITIAL_VELOCITY, A.COST_MINIMIZE_INITIAL_ACCELERATION])
    w = np.array([1.0, 100.0, 20.0])  # DigitSingleStepOptimizer.cpp:109-121
    assert opt.eval_f(n, x, True) == pytest.approx(float(w @ F[0]), rel=1e-14)
    assert np.allclose(opt.eval_grad_f(n, x, False), w @ GR[0], rtol=1e-14, atol=0)
    assert opt.eval_h(n, x, False, 1.0, m, np.zeros(m), True) is False


def solve_kinova(api, kinova, keep):
    rng = np.random.default_rng(5)
    q_des = rng.uniform(-0.6, 0.6, 7)
    opt = kinova_optimizer(api, kinova, keep, N=8, q_des=q_des)
    res = opt.solve(max_iter=30)
    return opt, res, q_des


def test_solve_with_the_stand_in_solver(oracle, kinova):
    """End-to-end through every callback (f, grad_f, g, jac_g, exact h) with scipy trust-constr standing in for Ipopt."""
    keep = []
    opt, res, q_des = solve_kinova(oracle, kinova, keep)
    f0 = opt.problem.objective_host(np.zeros(7), False)[0][0]
    assert opt.ifFeasible and opt.obj_value_copy < 1e-3 * f0
    # unconstrained optimum: q(tplan_n) = q_des, i.e. z = q_des / pq_pz(tplan_n)
    _, J = opt.problem.eval_host(np.zeros(7), A.EVAL_JAC)
    s = J[0, (8 // 2) * 7, 0]
    assert np.abs(opt.solution - q_des / s).max() < 1e-3


def test_solve_follows_ipopts_new_x_protocol_and_restores_a_feasible_iterate(oracle, kinova, monkeypatch):
    """solve() must flag new_x only on the first callback at an x (Ipopt's protocol): then an eval_f + eval_g pair at one x completes
    a record and finalize_solution can fall back to the best feasible iterate (Optimizer.cpp:154-196, 675-701).  A stand-in
    `minimize` visits a feasible point (f, g, grad, jac at the same x) and then returns an infeasible one."""
    import scipy.optimize as so

    keep = []
    opt = kinova_optimizer(oracle, kinova, keep, q_des=np.full(7, 0.5))
    feas, infeas = np.full(7, 0.2), np.full(7, 40.0)
    seen = []
    orig = {k: getattr(opt, k) for k in ("eval_f", "eval_g", "eval_grad_f", "eval_jac_g")}
    for k, fn in orig.items():
        monkeypatch.setattr(opt, k, (lambda name, f: (lambda n, x, new_x, *a: (seen.append((name, bool(new_x))), f(n, x, new_x, *a))[1]))(k, fn))

    class Res:
        pass

    def fake_minimize(fun, x0, jac=None, hess=None, constraints=(), **kw):
        con = constraints[0]
        for x in (feas, infeas):
            fun(x); con.fun(x); jac(x); con.jac(x)
        r = Res()
        r.x, r.fun = infeas, fun(infeas)
        return r

    monkeypatch.setattr(so, "minimize", fake_minimize)
    opt.solve()
    head = [s for s in seen[:8]]
    assert head == [("eval_f", True), ("eval_g", False), ("eval_grad_f", False), ("eval_jac_g", False)] * 2
    assert opt.ifFeasible and np.array_equal(opt.solution, feas)  # the infeasible result was replaced by the recorded feasible iterate
    assert opt.final_constr_violation <= opt.constr_viol_tol


@pytest.mark.gpu
def test_solve_on_the_gpu_matches_the_oracle_run(oracle, hip, kinova):
    keep = []
    oo, ro, _ = solve_kinova(oracle, kinova, keep)
    oh, rh, _ = solve_kinova(hip, kinova, keep)
    assert oh.ifFeasible and abs(oh.obj_value_copy - oo.obj_value_copy) < 1e-8
    assert np.abs(oh.solution - oo.solution).max() < 1e-5


def longer_horizon_optimizer(api, kinova, keep, degree=2, N=12):
    mh = api.model_create(kinova)
    keep.append(mh)
    c, rpy, size = H.example_boxes()
    q0, qT = np.zeros(7), np.full(7, 0.4)
    # straight line from q0 to qT as the initial guess (KinovaLongerHorizonPybindWrapper.cpp:143-149)
    x0 = np.zeros(21 * degree)
    for k in range(degree):
        x0[k * 21:k * 21 + 7] = q0 + (qT - q0) * (k + 1) / (degree + 1)
    return O.KinovaLongerHorizonOptimizer(api, mh, x0, 3.0, N, degree, q0, qT, c, rpy, size), q0, qT


def test_longer_horizon_shell(oracle, kinova):
    """KinovaLongerHorizonOptimizer.cpp:18-122: four blocks over PiecewiseBezierCurves, 20 x path length + 2 x jerk."""
    keep = []
    opt, q0, qT = longer_horizon_optimizer(oracle, kinova, keep)
    n, m, _, _ = opt.get_nlp_info()
    assert (n, m) == (42, 12 * 38)
    opt.get_bounds_info(n, m)
    x = opt.get_starting_point(n)
    F, GR = opt.problem.costs_host(x, [A.COST_MINIMIZE_PATH_LENGTH, A.COST_MINIMIZE_JERK])
    assert opt.eval_f(n, x, True) == pytest.approx(20.0 * F[0, 0] + 2.0 * F[0, 1], rel=1e-14)
    assert np.allclose(opt.eval_grad_f(n, x, False), 20.0 * GR[0, 0] + 2.0 * GR[0, 1], rtol=1e-14, atol=0)
    lam = np.random.default_rng(4).uniform(-1, 1, m)
    want = opt.problem.eval_h_host(x, lam)[0] + 0.7 * (20.0 * opt.problem.cost_hessian_host(A.COST_MINIMIZE_PATH_LENGTH) +
                                                        2.0 * opt.problem.cost_hessian_host(A.COST_MINIMIZE_JERK))
    assert np.allclose(opt.eval_h(n, x, False, 0.7, m, lam, True), want, rtol=1e-14, atol=0)  # Optimizer.cpp:552-563 + 565-619
    f0 = opt.eval_f(n, x, True)
    res = opt.solve(max_iter=25)
    assert opt.ifFeasible and opt.obj_value_copy < f0
    q, _, _ = opt.problem.trajectory_host(opt.solution, 12, 7)
    assert np.abs(q[0, 0] - q0).max() < 1e-13 and np.abs(q[0, -1] - qT).max() < 1e-13  # the ends are pinned by construction


@pytest.mark.gpu
def test_longer_horizon_shell_on_the_gpu(oracle, hip, kinova):
    keep = []
    oo, _, _ = longer_horizon_optimizer(oracle, kinova, keep)
    oh, _, _ = longer_horizon_optimizer(hip, kinova, keep)
    for o in (oo, oh):
        n, m, _, _ = o.get_nlp_info()
        o.get_bounds_info(n, m)
    x = oo.get_starting_point(n) + 0.05 * np.random.default_rng(2).uniform(-1, 1, n)
    assert abs(oh.eval_f(n, x, True) - oo.eval_f(n, x, True)) < 1e-11
    assert np.abs(oh.eval_grad_f(n, x, False) - oo.eval_grad_f(n, x, False)).max() < 1e-11
    assert H.rel_err(oh.eval_g(n, x, False, m), oo.eval_g(n, x, False, m)) < 1e-9
    assert H.rel_err(oh.eval_jac_g(n, x, False, m), oo.eval_jac_g(n, x, False, m)) < 1e-9
    lam = np.random.default_rng(5).uniform(-1, 1, m)
    assert H.rel_err(oh.eval_h(n, x, False, 0.5, m, lam, True), oo.eval_h(n, x, False, 0.5, m, lam, True)) < 1e-9
