"""Host-side mirror of the reference's solver shell, `class Optimizer : Ipopt::TNLP`
(Optimization/include/Optimizer.h:29-259, Optimization/src/Optimizer.cpp), over the C ABI.

Same callback names, argument meaning and bookkeeping as the reference: `get_nlp_info`, `get_bounds_info`,
`get_starting_point`, `eval_f`, `eval_grad_f`, `eval_g`, `eval_jac_g`, `eval_h`, `finalize_solution`,
`summarize_constraints`, the best-feasible-iterate tracking of `update_minimal_cost_solution` and the per-callback wall-clock
counters.  The arithmetic of every callback is one call into the library (`raptor_eval_batch_host`, `raptor_eval_costs_host`,
`raptor_kinova_eval_objective_host`, `raptor_eval_h_batch_host`); Ipopt itself is not in this image, so `solve()` drives the
callbacks from scipy's `trust-constr` as a stand-in NLP solver (tests and demos only: the product is the callbacks).

`KinovaOptimizer`  Examples/Kinova/CollisionAvoidanceTrajectory/src/KinovaOptimizer.cpp:18-250
`SingleStepOptimizer`  Examples/Digit/src/DigitSingleStepOptimizer.cpp:17-124 and the Talos / Unitree-G1 twins
"""
from __future__ import annotations

import time

import numpy as np

from . import _capi as A

INFEASIBLE, FEASIBLE, UNINITIALIZED = 0, 1, 2  # OptimizerConstants::FeasibleState (Optimizer.h:22-26)
_MAX = float(np.finfo(np.float64).max)


class IpoptException(RuntimeError):
    """THROW_EXCEPTION(IpoptException, ...) of the reference."""


class Optimizer:
    def __init__(self, problem: A.ProblemHandle, x0, constraint_names):
        self.problem = problem
        self.x0 = np.array(x0, dtype=np.float64)
        self.constraintsNameVec_ = list(constraint_names)
        self.enable_hessian = False      # Optimizer.h:223
        self.constr_viol_tol = 1e-4      # Optimizer.h:256
        self.display_info = False        # the reference defaults to true; quiet here
        self.numVars = 0
        self.numCons = 0
        self.g_lb = self.g_ub = None
        self.solution = np.zeros(0)
        self.g_copy = np.zeros(0)
        self.obj_value_copy = 0.0
        self.ifFeasible = False
        self.final_constr_violation = 0.0
        self.ifFeasibleCurrIter = False
        self._reset_tracking()
        self._reset_timers()

    # ---- bookkeeping state (Optimizer.cpp:13-32, 708-715, 756-766) ----
    def _reset_tracking(self):
        self.currentIpoptSolution = np.zeros(0)
        self.currentIpoptObjValue = _MAX
        self.ifCurrentIpoptFeasible = UNINITIALIZED
        self.optimalIpoptSolution = np.zeros(0)
        self.optimalIpoptObjValue = _MAX
        self.ifOptimalIpoptFeasible = UNINITIALIZED

    def _reset_timers(self):
        self.nlp_time = dict(f=0.0, g=0.0, grad_f=0.0, jac_g=0.0, hess_l=0.0)   # microseconds
        self.nlp_count = dict(f=0, g=0, grad_f=0, jac_g=0, hess_l=0)

    def _tick(self, key, t0):
        self.nlp_time[key] += (time.perf_counter() - t0) * 1e6
        self.nlp_count[key] += 1

    # ---- TNLP ----
    def get_nlp_info(self):
        """-> n, m, nnz_jac_g, nnz_h_lag (KinovaOptimizer.cpp:121-147: dense Jacobian, upper-triangle Hessian)."""
        self.numVars = self.problem.n
        self.numCons = self.problem.m
        return self.numVars, self.numCons, self.numVars * self.numCons, self.numVars * (self.numVars + 1) // 2

    def get_bounds_info(self, n, m):
        """-> x_l, x_u, g_l, g_u (Optimizer.cpp:36-111)."""
        if n != self.numVars:
            raise IpoptException("*** Error wrong value of n in get_bounds_info!")
        if m != self.numCons:
            raise IpoptException("*** Error wrong value of m in get_bounds_info!")
        if len(self.constraintsNameVec_) != len(self.problem.block_offsets) - 1:
            raise IpoptException("*** Error constraintsPtrVec_ and constraintsNameVec_ have different sizes!")
        self.g_lb, self.g_ub = self.problem.bounds()
        if self.display_info:
            print("Dimension of each constraints and their locations: ")
            off = self.problem.block_offsets
            for c, name in enumerate(self.constraintsNameVec_):
                print(f"{name}: {off[c + 1] - off[c]} [{off[c]} {off[c + 1]}]")
        return np.full(n, -1e19), np.full(n, 1e19), self.g_lb.copy(), self.g_ub.copy()

    def get_starting_point(self, n, init_x=True, init_z=False, init_lambda=False):
        """Optimizer.cpp:116-150."""
        if not init_x or init_z or init_lambda:
            raise IpoptException("*** Error wrong value of init in get_starting_point!")
        if n != self.numVars:
            raise IpoptException("*** Error wrong value of n in get_starting_point!")
        if self.x0.size != self.numVars:
            raise IpoptException("*** Error x0.size() != numVars in get_starting_point!")
        return self.x0.copy()

    def update_minimal_cost_solution(self, n, z, new_x, obj_value):
        """Optimizer.cpp:154-196."""
        if n != self.numVars:
            raise IpoptException("*** Error wrong value of n in update_minimal_cost_solution!")
        if new_x:
            self.currentIpoptSolution = z.copy()
            self.currentIpoptObjValue = obj_value
            self.ifCurrentIpoptFeasible = UNINITIALIZED
        elif self.currentIpoptSolution.shape == z.shape and np.array_equal(self.currentIpoptSolution, z):  # ifTwoVectorEqual(., ., 0)
            if self.ifCurrentIpoptFeasible == UNINITIALIZED:
                raise IpoptException("*** Error ifCurrentIpoptFeasible is not initialized!")
            self.currentIpoptObjValue = obj_value
        else:
            self.currentIpoptSolution = z.copy()
            self.currentIpoptObjValue = obj_value
            self.ifCurrentIpoptFeasible = UNINITIALIZED
        self._update_optimal()

    def _update_optimal(self):
        if self.ifCurrentIpoptFeasible == FEASIBLE and self.currentIpoptObjValue < self.optimalIpoptObjValue:
            self.optimalIpoptSolution = self.currentIpoptSolution.copy()
            self.optimalIpoptObjValue = self.currentIpoptObjValue
            self.ifOptimalIpoptFeasible = self.ifCurrentIpoptFeasible

    # objective: subclasses provide _objective(z, derivatives) -> f, grad_f (Optimizer.cpp:201-295 or an override)
    def _objective(self, z, derivatives):
        raise IpoptException("*** Error costsPtrVec_ is empty!")

    def eval_f(self, n, x, new_x):
        if n != self.numVars:
            raise IpoptException("*** Error wrong value of n in eval_f!")
        t0 = time.perf_counter()
        z = np.array(x, dtype=np.float64)
        obj_value, _ = self._wrap("eval_f", self._objective, z, False)
        self.update_minimal_cost_solution(n, z, new_x, obj_value)
        self._tick("f", t0)
        return obj_value

    def eval_grad_f(self, n, x, new_x):
        if n != self.numVars:
            raise IpoptException("*** Error wrong value of n in eval_grad_f!")
        t0 = time.perf_counter()
        _, grad = self._wrap("eval_grad_f", self._objective, np.array(x, dtype=np.float64), True)
        self._tick("grad_f", t0)
        return grad

    def eval_g(self, n, x, new_x, m):
        """Optimizer.cpp:341-428."""
        if n != self.numVars:
            raise IpoptException("*** Error wrong value of n in eval_g!")
        if m != self.numCons:
            raise IpoptException("*** Error wrong value of m in eval_g!")
        t0 = time.perf_counter()
        z = np.array(x, dtype=np.float64)
        g = self._wrap("eval_g", lambda: self.problem.eval_host(z, A.EVAL_G)[0][0])
        off = self.problem.block_offsets
        self.ifFeasibleCurrIter = True
        for c in range(len(off) - 1):  # :374-378, block by block
            s = slice(off[c], off[c + 1])
            if off[c + 1] > off[c] and ((g[s] - self.g_lb[s]).min() < -self.constr_viol_tol or
                                        (self.g_ub[s] - g[s]).min() < -self.constr_viol_tol):
                self.ifFeasibleCurrIter = False
        state = FEASIBLE if self.ifFeasibleCurrIter else INFEASIBLE
        if new_x:
            self.currentIpoptSolution = z.copy()
            self.currentIpoptObjValue = _MAX
            self.ifCurrentIpoptFeasible = state
        elif self.currentIpoptSolution.shape == z.shape and np.array_equal(self.currentIpoptSolution, z):
            if self.currentIpoptObjValue == _MAX:
                raise IpoptException("*** Error currentIpoptObjValue is not initialized!")
            self.ifCurrentIpoptFeasible = state
        else:
            self.currentIpoptSolution = z.copy()
            self.currentIpoptObjValue = _MAX
            self.ifCurrentIpoptFeasible = state
        self._update_optimal()
        self._tick("g", t0)
        return g

    def eval_jac_g(self, n, x, new_x, m, values=True):
        """values=False -> (iRow, jCol) of the dense structure (Optimizer.cpp:451-460); else `values` row-major (:467-486)."""
        if n != self.numVars:
            raise IpoptException("*** Error wrong value of n in eval_jac_g!")
        if m != self.numCons:
            raise IpoptException("*** Error wrong value of m in eval_jac_g!")
        if not values:
            return np.repeat(np.arange(m), n), np.tile(np.arange(n), m)
        t0 = time.perf_counter()
        z = np.array(x, dtype=np.float64)
        jac = self._wrap("eval_jac_g", lambda: self.problem.eval_host(z, A.EVAL_JAC)[1][0])
        self._tick("jac_g", t0)
        return jac.reshape(-1)

    def _hess_f(self, z, obj_factor):
        """obj_factor * hess_f as the library's packed upper triangle, or None when the objective has no Hessian here."""
        return None

    def eval_h(self, n, x, new_x, obj_factor, m, lam, new_lambda, values=True):
        """Optimizer.cpp:499-635.  Returns False when Hessians are disabled (:513-515)."""
        if not self.enable_hessian:
            return False
        if n != self.numVars:
            raise IpoptException("*** Error wrong value of n in eval_h!")
        if m != self.numCons:
            raise IpoptException("*** Error wrong value of m in eval_h!")
        if not values:
            iu = np.triu_indices(n)
            return iu[0], iu[1]
        t0 = time.perf_counter()
        z = np.array(x, dtype=np.float64)
        H = self._wrap("eval_h", lambda: self.problem.eval_h_host(z, np.asarray(lam, dtype=np.float64),
                                                                   None if obj_factor == 0 else obj_factor)[0])
        self._tick("hess_l", t0)
        return H

    def _wrap(self, where, fn, *args):
        try:
            return fn(*args)
        except (A.RaptorError, A.RaptorInvalidArgument) as e:  # :365-372: std::exception -> IpoptException
            raise IpoptException(f"*** Error in {where}! Check previous error message. ({e})") from e

    def summarize_constraints(self, m, g, verbose=True):
        """Optimizer.cpp:771-829."""
        if m != self.numCons:
            raise IpoptException("*** Error wrong value of m in summarize_constraints!")
        g = np.asarray(g)
        off = self.problem.block_offsets
        self.ifFeasible = True
        self.final_constr_violation = 0.0
        report = []
        for c in range(len(off) - 1):
            s = slice(off[c], off[c + 1])
            if off[c + 1] == off[c]:
                continue
            viol = np.fmax(self.g_lb[s] - g[s], g[s] - self.g_ub[s])
            worst = float(viol.max())
            self.final_constr_violation = max(self.final_constr_violation, worst, 0.0)
            if worst > self.constr_viol_tol:
                i = int(np.argmax(viol))  # first maximum, as the strict `>` of :797 keeps it
                report.append((self.constraintsNameVec_[c], worst, self.g_lb[s][i], self.g_ub[s][i], g[s][i]))
                self.ifFeasible = False
        if self.display_info and verbose:
            print("Constraint violation report:")
            for name, worst, lo, up, val in report:
                print(f"{name}: {worst}\n    range: [{lo}, {up}], value: {val}")
            print(f"Total constraint violation: {self.final_constr_violation}")
        return report

    def finalize_solution(self, n, x, m, g, obj_value):
        """Optimizer.cpp:639-767 (the dual variables and Ipopt's own data are not used by the reference either)."""
        if n != self.numVars:
            raise IpoptException("*** Error wrong value of n in finalize_solution!")
        if m != self.numCons:
            raise IpoptException("*** Error wrong value of m in finalize_solution!")
        self.solution = np.array(x, dtype=np.float64)
        self.obj_value_copy = self.eval_f(n, x, True)
        self.g_copy = self.eval_g(n, x, True, m)
        self.summarize_constraints(m, g, False)
        recorded = (self.optimalIpoptSolution.size == n and self.optimalIpoptObjValue < _MAX and
                    self.ifOptimalIpoptFeasible == FEASIBLE)
        if recorded and (not self.ifFeasible or self.optimalIpoptObjValue < obj_value):
            self.ifFeasible = True
            self.solution = self.optimalIpoptSolution.copy()
            best = self.optimalIpoptObjValue
            self.obj_value_copy = self.eval_f(n, self.solution, True)
            self.g_copy = self.eval_g(n, self.solution, True, m)
            self.summarize_constraints(m, self.g_copy)
            if self.display_info:
                print("Solution is not feasible or optimal but we have found one optimal feasible solution before"
                      f" with cost: {best} and constraint violation: {self.final_constr_violation}")
        else:
            self.summarize_constraints(m, g)
        if self.display_info:
            print(f"Objective value: {self.obj_value_copy}")
            print("  optimizer  :     t_wall      avg     n_eval")
            for key in ("f", "g", "grad_f", "jac_g") + (("hess_l",) if self.enable_hessian else ()):
                c, t = self.nlp_count[key], self.nlp_time[key]
                print(f"  nlp_{key:7s}| {t / 1000.0:8.2f}ms " + (f"{t / c:8.2f}us" if c else f"{'N/A':>8s}") + f" {c:8d}")
        self.ifFeasibleCurrIter = False
        self._reset_tracking()
        timers = (dict(self.nlp_time), dict(self.nlp_count))
        self._reset_timers()
        return timers

    # ---- stand-in for IpoptApplication::OptimizeTNLP (Ipopt is not in this image) ----
    def solve(self, max_iter=50, tol=1e-6, verbose=0):
        """Drives the TNLP callbacks from scipy's trust-constr (exact Hessians when enable_hessian, BFGS otherwise) and ends with
        finalize_solution, like `app->OptimizeTNLP(mynlp)` (KinovaExample.cpp:122).  Returns scipy's result object."""
        from scipy.optimize import BFGS, NonlinearConstraint, minimize

        n, m, _, _ = self.get_nlp_info()
        _, _, g_l, g_u = self.get_bounds_info(n, m)
        x0 = self.get_starting_point(n)
        lb = np.where(g_l <= -1e19, -np.inf, g_l)
        ub = np.where(g_u >= 1e19, np.inf, g_u)
        iu = np.triu_indices(n)

        def unpack(h):
            M = np.zeros((n, n))
            M[iu] = h
            return M + np.triu(M, 1).T

        # Ipopt's protocol for new_x: true on the FIRST callback at an x it has not shown the TNLP before, false on the callbacks
        # that follow at the same x.  The reference's best-feasible-iterate bookkeeping depends on it (an eval_f and an eval_g at
        # the same x, the second with new_x = false, complete one record: Optimizer.cpp:154-196, 380-421).
        last = {"x": None}

        def fresh(x):
            x = np.asarray(x, dtype=np.float64)
            if last["x"] is not None and last["x"].shape == x.shape and np.array_equal(last["x"], x):
                return False
            last["x"] = x.copy()
            return True

        if self.enable_hessian:
            hess_f = lambda x: unpack(self.eval_h(n, x, fresh(x), 1.0, m, np.zeros(m), True))  # noqa: E731
            hess_g = lambda x, v: unpack(self.eval_h(n, x, fresh(x), 0.0, m, v, True))  # noqa: E731
        else:
            hess_f, hess_g = BFGS(), BFGS()
        con = NonlinearConstraint(lambda x: self.eval_g(n, x, fresh(x), m), lb, ub,
                                  jac=lambda x: self.eval_jac_g(n, x, fresh(x), m).reshape(m, n), hess=hess_g)
        res = minimize(lambda x: self.eval_f(n, x, fresh(x)), x0, jac=lambda x: self.eval_grad_f(n, x, fresh(x)), hess=hess_f,
                       constraints=[con], method="trust-constr", options=dict(maxiter=max_iter, gtol=tol, xtol=1e-10, verbose=verbose))
        self.finalize_solution(n, res.x, m, self.eval_g(n, res.x, True, m), float(res.fun))
        return res


class KinovaOptimizer(Optimizer):
    """KinovaOptimizer::set_parameters (KinovaOptimizer.cpp:18-116) + its own eval_f / eval_grad_f / eval_hess_f (:152-250)."""

    NAMES = ["joint limits", "velocity limits", "torque limits", "obstacle avoidance constraints"]  # :86, :92, :99, :109

    def __init__(self, api, model_handle, x0, T, N, atp, boxCenters, boxOrientation, boxSize, q_des, tplan_n,
                 joint_limits_buffer=None, velocity_limits_buffer=None, torque_limits_buffer=None,
                 include_gripper_or_not=False, collision_buffer=0.0):
        self._keep = []
        q0, q_d0, q_dd0 = atp  # ArmourTrajectoryParameters
        traj = A.make_traj_desc(self._keep, A.TRAJ_ARMOUR_BEZIER, T, N, 7, 5, A.TIME_CHEBYSHEV, q0, q_d0, q_dd0)
        prob = A.ProblemHandle.kinova(api, model_handle, traj, boxCenters, boxOrientation, boxSize,
                                      joint_limits_buffer=joint_limits_buffer, velocity_limits_buffer=velocity_limits_buffer,
                                      torque_limits_buffer=torque_limits_buffer, include_gripper=include_gripper_or_not,
                                      collision_buffer=collision_buffer)
        super().__init__(prob, x0, self.NAMES)
        self.enable_hessian = True  # :37
        q_des = np.asarray(q_des, dtype=np.float64)
        if self.x0.size != prob.n or q_des.size != 7 or not (0 <= tplan_n < N):  # the asserts of :111-113
            raise ValueError("KinovaOptimizer: x0 / q_des / tplan_n do not fit the trajectory")
        prob.set_objective(q_des, tplan_n)

    def _objective(self, z, derivatives):
        F, GR = self.problem.objective_host(z, derivatives)
        return float(F[0]), (GR[0] if derivatives else None)


class SingleStepOptimizer(Optimizer):
    """Digit / Talos / Unitree-G1 single-step optimizers: the problem handle comes from ProblemHandle.digit / .legged; the costs
    are MinimizePower (1), MinimizeInitialVelocity (100), MinimizeInitialAcceleration (20)
    (DigitSingleStepOptimizer.cpp:109-121), summed with their weights in eval_f / eval_grad_f (Optimizer.cpp:222-235, 270-288)."""

    def __init__(self, problem, x0, constraint_names,
                 costs=((A.COST_MINIMIZE_POWER, 1.0, "minimize power"),
                        (A.COST_MINIMIZE_INITIAL_VELOCITY, 100.0, "minimize initial velocity"),
                        (A.COST_MINIMIZE_INITIAL_ACCELERATION, 20.0, "minimize initial acceleration"))):
        super().__init__(problem, x0, constraint_names)
        self.costsKinds = [c[0] for c in costs]
        self.costsWeightVec_ = np.array([c[1] for c in costs], dtype=np.float64)
        self.costsNameVec_ = [c[2] for c in costs]

    def _objective(self, z, derivatives):
        if not self.costsKinds:
            raise IpoptException("*** Error costsPtrVec_ is empty!")
        F, GR = self.problem.costs_host(z, self.costsKinds, derivatives)
        obj = 0.0
        for k, w in enumerate(self.costsWeightVec_):
            obj += w * F[0, k]
        grad = None
        if derivatives:
            grad = np.zeros(self.problem.n)
            for k, w in enumerate(self.costsWeightVec_):
                grad += w * GR[0, k]
        return float(obj), grad


class KinovaLongerHorizonOptimizer(SingleStepOptimizer):
    """KinovaLongerHorizonOptimizer::set_parameters (KinovaLongerHorizonOptimizer.cpp:18-122): PiecewiseBezierCurves, the four Kinova
    blocks, MinimizePathLength (20) + MinimizeJerk (2), enable_hessian = true (:36).  eval_h = the library's constraint part plus
    obj_factor * sum_k weight_k * hess_f_k (Optimizer::eval_hess_f, Optimizer.cpp:298-337; both costs are quadratic, their Hessians
    constant)."""

    def __init__(self, api, model_handle, x0, T, N, degree, q0, qT, boxCenters, boxOrientation, boxSize,
                 joint_limits_buffer=None, velocity_limits_buffer=None, torque_limits_buffer=None,
                 include_gripper_or_not=False, collision_buffer=0.0):
        self._keep = []
        traj = A.make_traj_desc(self._keep, A.TRAJ_PIECEWISE_BEZIER, T, N, 7, degree, A.TIME_UNIFORM, q0=q0, qT=qT)
        prob = A.ProblemHandle.kinova(api, model_handle, traj, boxCenters, boxOrientation, boxSize,
                                      joint_limits_buffer=joint_limits_buffer, velocity_limits_buffer=velocity_limits_buffer,
                                      torque_limits_buffer=torque_limits_buffer, include_gripper=include_gripper_or_not,
                                      collision_buffer=collision_buffer)
        super().__init__(prob, x0, KinovaOptimizer.NAMES,
                         costs=((A.COST_MINIMIZE_PATH_LENGTH, 20.0, "minimize path length"),
                                (A.COST_MINIMIZE_JERK, 2.0, "minimize jerk for smoothness")))
        self.enable_hessian = True
        self._hess_f_packed = sum(w * prob.cost_hessian_host(k) for k, w in zip(self.costsKinds, self.costsWeightVec_))

    def eval_h(self, n, x, new_x, obj_factor, m, lam, new_lambda, values=True):
        if not values or not self.enable_hessian:
            return super().eval_h(n, x, new_x, obj_factor, m, lam, new_lambda, values)
        H = super().eval_h(n, x, new_x, 0.0, m, lam, new_lambda, True)  # constraint part (Optimizer.cpp:565-619)
        return H + obj_factor * self._hess_f_packed if obj_factor != 0 else H  # :552-563
