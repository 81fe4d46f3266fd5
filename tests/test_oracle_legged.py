"""Pins the legged / Digit part of the CPU oracle (runs without a GPU).

The reference holds no golden vectors for this path: its own criterion is finite differences (whole-
problem derivative check 1e-4, DigitSingleStep.cpp:119-131; TestInverseDynamics.cpp 1e-5).  On top of that recipe the
oracle -- a literal restatement of DynamicsConstraints / ConstrainedInverseDynamics with FK Hessians, third-order
tensors and dense solves -- is cross-checked against an INDEPENDENT formulation (numpy, scripts/proto_*.py: fixed-base
tree re-rooted at the stance foot, implicit loop derivatives), and against statics (contact force = total weight).
"""
import os
import sys

import numpy as np
import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _fd(p, Z, N, nv, cols, h=1e-6):
    ref = p.constrained_id_host(Z, N, nv, 12)
    fd_t = np.zeros_like(ref["ptau_pz"][0])
    fd_l = np.zeros_like(ref["plambda_pz"][0])
    for c in cols:
        zp, zm = Z.copy(), Z.copy()
        zp[0, c] += h
        zm[0, c] -= h
        a = p.constrained_id_host(zp, N, nv, 12, derivatives=False)
        b = p.constrained_id_host(zm, N, nv, 12, derivatives=False)
        fd_t[:, :, c] = (a["tau"][0] - b["tau"][0]) / (2 * h)
        fd_l[:, :, c] = (a["lam"][0] - b["lam"][0]) / (2 * h)
    et = np.abs(fd_t[:, :, cols] - ref["ptau_pz"][0][:, :, cols]).max() / (1 + np.abs(fd_t).max())
    el = np.abs(fd_l[:, :, cols] - ref["plambda_pz"][0][:, :, cols]).max() / (1 + np.abs(fd_l).max())
    return et, el, ref


@pytest.mark.parametrize("which", ["talos", "g1"])
def test_contact_constrained_id_finite_differences(oracle, which):
    keep = []
    p, m, cfg = H.legged_problem(oracle, which, keep, N=3)
    Z = 0.3 * np.random.default_rng(1).uniform(-1, 1, (1, p.n))
    et, el, ref = _fd(p, Z, 3, m.nv, list(range(0, 84, 5)))
    assert et < 1e-7 and el < 1e-7, (et, el)
    assert np.all(ref["ptau_pz"][..., 84:] == 0)


@pytest.mark.parametrize("stance", ["L", "R"])
def test_digit_constrained_id_finite_differences(oracle, stance):
    keep = []
    p, m, cfg = H.digit_problem(oracle, keep, N=3, stance=stance)
    Z = 0.1 * np.random.default_rng(2).uniform(-1, 1, (1, p.n))
    et, el, ref = _fd(p, Z, 3, 36, list(range(0, 72, 5)))
    assert et < 1e-7 and el < 1e-7, (et, el)


@pytest.mark.parametrize("robot", ["talos", "g1", "digit"])
def test_static_contact_wrench_is_the_weight(oracle, robot):
    """q constant (all Bezier coefficients of a joint equal): v = a = 0, so lambda = (0, 0, m g, mx, my, 0)."""
    keep = []
    if robot == "digit":
        p, m, cfg = H.digit_problem(oracle, keep, N=3)
        K, g = 6, 9.806
    else:
        p, m, cfg = H.legged_problem(oracle, robot, keep, N=3)
        K, g = 7, 9.81
    q0 = 0.1 * np.random.default_rng(3).uniform(-1, 1, 12)
    Z = np.zeros((1, p.n))
    Z[0, : 12 * K] = np.repeat(q0, K)
    r = p.constrained_id_host(Z, 3, m.nv, 12, derivatives=False)
    assert np.abs(r["v"]).max() < 1e-12 and np.abs(r["a"]).max() < 1e-10
    lam = r["lam"][0, 0]
    assert abs(lam[2] - m.mass.sum() * g) < 1e-9 and np.abs(lam[[0, 1, 5]]).max() < 1e-9


@pytest.mark.parametrize("which", ["talos", "g1"])
def test_independent_formulation_contact(oracle, which):
    import proto_rerooted

    worst = proto_rerooted.run(which, N=3)
    assert max(worst.values()) < 1e-11, worst


@pytest.mark.parametrize("stance", ["L", "R"])
def test_independent_formulation_digit(oracle, stance):
    import proto_digit

    worst = proto_digit.run(stance, N=2)
    assert max(worst.values()) < 1e-10, worst


def test_digit_layout_bounds_and_errors(oracle):
    from raptor_amd import _capi as A

    keep = []
    p, m, cfg = H.digit_problem(oracle, keep, N=14)
    assert (p.n, p.m, p.block_offsets) == (132, 770, [0, 168, 672, 770])  # DigitSingleStepOptimizer.cpp:74-94 at N = 14
    lb, ub = p.bounds()
    assert lb[0] == -113.0 and ub[3] == 231.317                            # DigitConstants.h:96-123
    assert np.isclose(lb[168 + 6], np.deg2rad(-60)) and np.isclose(ub[168 + 8], np.deg2rad(90))
    assert np.all(lb[672::7] == 0) and np.all(ub[672::7] == 1e19) and np.all(ub[673::7] == 0)
    cfg.stance_leg = "x"
    mh = oracle.model_create(m)
    keep.append(mh)
    t = A.make_traj_desc(keep, A.TRAJ_BEZIER, cfg.T, 4, 12, cfg.degree, A.TIME_CHEBYSHEV)
    with pytest.raises(A.RaptorError):
        A.ProblemHandle.digit(oracle, mh, t, cfg)


@pytest.mark.parametrize("robot", ["talos", "digit"])
def test_costs_gradient_finite_differences(oracle, robot):
    """TestMinimizePowerCosts.cpp:13-78 / TestMinimizeTorqueCosts.cpp:13-76 recipe (central differences, 1e-4)."""
    from raptor_amd import _capi as A

    keep = []
    if robot == "digit":
        p, m, cfg = H.digit_problem(oracle, keep, N=4)
        ndyn = 72
    else:
        p, m, cfg = H.legged_problem(oracle, robot, keep, N=4)
        ndyn = 84
    Z = 0.1 * np.random.default_rng(4).uniform(-1, 1, (1, p.n))
    kinds = [A.COST_MINIMIZE_POWER, A.COST_MINIMIZE_TORQUE, A.COST_MINIMIZE_INITIAL_VELOCITY, A.COST_MINIMIZE_INITIAL_ACCELERATION]
    F, GR = p.costs_host(Z, kinds)
    assert np.all(F > 0) and np.all(GR[0][:, ndyn:] == 0)
    h = 1e-6
    cols = list(range(0, ndyn, 5))
    for c in cols:
        zp, zm = Z.copy(), Z.copy()
        zp[0, c] += h
        zm[0, c] -= h
        fd = (p.costs_host(zp, kinds, False)[0][0] - p.costs_host(zm, kinds, False)[0][0]) / (2 * h)
        assert np.all(np.abs(fd - GR[0][:, c]) < 1e-4 * (1 + np.abs(GR[0]).max(axis=1))), c
    # ||q_d(0)|| of an all-zero trajectory: below the square-root threshold the gradient stays zero (:26)
    F0, G0 = p.costs_host(np.zeros((1, p.n)), kinds[2:])
    assert np.all(F0 == 0) and np.all(G0 == 0)


@pytest.mark.parametrize("robot", ["digit", "talos"])
def test_customized_block_layout_and_fd(oracle, robot):
    from raptor_amd import _capi as A

    keep = []
    N = 5
    if robot == "digit":
        p, m, cfg = H.digit_problem(oracle, keep, N=N, customized=True)
        rows, ndyn = N * 10 + 4, 72
    else:
        p, m, cfg = H.legged_problem(oracle, robot, keep, N=N, customized=True)
        rows, ndyn = N * 11 + 2, 84
    off = p.block_offsets[3]
    assert p.block_offsets[4] - off == rows == p.m - off
    lb, ub = p.bounds()
    assert lb[off + N // 2] == cfg.gait["swingfoot_midstep_z_des"] and ub[off] == 0 and ub[off + N - 1] == 0 and ub[off + 1] == 1e19
    Z = 0.1 * np.random.default_rng(6).uniform(-1, 1, (1, p.n))
    G, J = p.eval_host(Z)
    h = 1e-6
    for c in range(0, ndyn, 6):
        zp, zm = Z.copy(), Z.copy()
        zp[0, c] += h
        zm[0, c] -= h
        fd = (p.eval_host(zp, A.EVAL_G)[0][0, off:] - p.eval_host(zm, A.EVAL_G)[0][0, off:]) / (2 * h)
        assert np.abs(fd - J[0, off:, c]).max() < 1e-6


@pytest.mark.parametrize("robot", ["talos", "g1", "digit"])
def test_reset_map_block_layout_bounds_and_fd(oracle, robot):
    """*SingleStepPeriodicityConstraints: row count and bounds (.cpp:12-16, 217-226) and the reference's own recipe for
    constraint gradients -- central differences of g against pg_pz over every kind of column (trajectory, v+, lambda)."""
    from raptor_amd import _capi as A

    keep = []
    N = 4
    if robot == "digit":
        p, m, cfg = H.digit_problem(oracle, keep, N=N, periodic=True)
    else:
        p, m, cfg = H.legged_problem(oracle, robot, keep, N=N, periodic=True)
    nv, nd, ni = m.nv, len(cfg.dependent_ids), 12
    rows = nv + nd + 2 * ni + 7
    off = p.block_offsets[-2]
    assert p.block_offsets[-1] - off == rows == p.m - off
    lb, ub = p.bounds()
    o5 = off + nv + nd + 2 * ni
    assert np.all(lb[off:o5] == 0) and np.all(ub[off:o5] == 0)
    assert lb[o5] == 0 and ub[o5] == 1e19 and np.all(lb[o5 + 1:o5 + 7] == -1e19) and np.all(ub[o5 + 1:o5 + 7] == 0)
    rng = np.random.default_rng(8)
    Z = 0.1 * rng.uniform(-1, 1, (1, p.n))
    Z[0, -6:] *= 30
    G, J = p.eval_host(Z)
    h = 1e-6
    cols = list(range(0, p.n - nv - nd, 5)) + list(range(p.n - nv - nd, p.n, 7)) + list(range(p.n - 6, p.n))
    for c in cols:
        zp, zm = Z.copy(), Z.copy()
        zp[0, c] += h
        zm[0, c] -= h
        fd = (p.eval_host(zp, A.EVAL_G)[0][0, off:] - p.eval_host(zm, A.EVAL_G)[0][0, off:]) / (2 * h)
        assert np.abs(fd - J[0, off:, c]).max() < 1e-5 * (1 + np.abs(J[0, off:, c]).max()), c
    # a consistent impact: with v+ = v- and lambda = 0 the first block vanishes, and J v- = 0 would need the OLD stance
    Z2 = Z.copy()
    r = p.constrained_id_host(Z2, N, nv, ni, derivatives=False)
    Z2[0, p.n - nv - nd:p.n - nd] = r["v"][0, N - 1]
    Z2[0, p.n - nd:] = 0
    G2, _ = p.eval_host(Z2, A.EVAL_G)
    assert np.abs(G2[0, off:off + nv]).max() < 1e-10


@pytest.mark.parametrize("robot", ["talos", "digit"])
def test_independent_formulation_reset_map(oracle, robot):
    """numpy model of the directional-derivative formulation the HIP kernel uses (scripts/proto_period.py) against the
    oracle's literal restatement (crba, J tensors, gravity-free RNEA derivatives, dense products)."""
    import proto_period

    eg, ej = proto_period.run(robot, N=3)
    assert eg < 1e-11 and ej < 1e-10


def test_bezier_constrained_initial_state(oracle, kinova):
    """BezierCurves::constrainInitialPosition / constrainInitialVelocity (BezierCurves.cpp:39-69, 84-96, 148-162): q(0) = q0,
    q_d(0) = q_d0 at degree 5 (the T / 5.0 of :86 is the degree-5 end-point rule; at other degrees it scales, as in the
    reference), Nact fewer variables each, pq_pz by central differences, and the ordering rule (position first)."""
    from raptor_amd import _capi as A

    rng = np.random.default_rng(9)
    q0, qd0 = rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 7)
    keep = []
    p1 = H.id_problem(oracle, kinova, keep, ttype=A.TRAJ_BEZIER, degree=5, disc=A.TIME_UNIFORM, q0=q0)
    p2 = H.id_problem(oracle, kinova, keep, ttype=A.TRAJ_BEZIER, degree=5, disc=A.TIME_UNIFORM, q0=q0, q_d0=qd0)
    p3 = H.id_problem(oracle, kinova, keep, ttype=A.TRAJ_BEZIER, degree=6, disc=A.TIME_UNIFORM, q0=q0, q_d0=qd0)
    assert (p1.n, p2.n, p3.n) == (35, 28, 35)
    for p, deg in ((p1, 5), (p2, 5), (p3, 6)):
        Z = rng.uniform(-1, 1, (1, p.n))
        q, qd, qdd = p.trajectory_host(Z, 10, 7)
        assert np.abs(q[0, 0] - q0).max() < 1e-15
        if p is not p1:
            assert np.abs(qd[0, 0] - qd0 * deg / 5.0).max() < 1e-13
        h = 1e-6
        tau, ptau = p.inverse_dynamics_host(Z, 10, 7)
        for c in range(0, p.n, 3):
            zp, zm = Z.copy(), Z.copy()
            zp[0, c] += h
            zm[0, c] -= h
            fd = (p.inverse_dynamics_host(zp, 10, 7, derivatives=False)[0] - p.inverse_dynamics_host(zm, 10, 7, derivatives=False)[0]) / (2 * h)
            assert np.abs(fd[0] - ptau[0, :, :, c]).max() < 1e-5 * (1 + np.abs(ptau[0, :, :, c]).max())
    with pytest.raises(A.RaptorInvalidArgument):  # "initial position must be constrained first"
        H.id_problem(oracle, kinova, keep, ttype=A.TRAJ_BEZIER, degree=5, q_d0=qd0)


@pytest.mark.parametrize("robot", ["talos", "digit"])
def test_step_link_block_fd(oracle, robot):
    """*MultipleStepPeriodicityConstraints (DigitMultipleStepPeriodicityConstraints.cpp:57-230): the link between walking step s
    and s + 1.  Same rows as the single-step block except the reset rows; gradient w.r.t. both steps' variables by central
    differences, and the reset rows vanish when the next step starts where this one ends."""
    from raptor_amd import _capi as A

    keep = []
    N = 4
    if robot == "digit":
        pc, m, cfg = H.digit_problem(oracle, keep, N=N, stance="L", periodic=True)
        pn, _, _ = H.digit_problem(oracle, keep, N=N, stance="R")
    else:
        pc, m, cfg = H.legged_problem(oracle, robot, keep, N=N, stance="L", periodic=True)
        pn, _, _ = H.legged_problem(oracle, robot, keep, N=N, stance="R")
    nv, nd, ni = m.nv, len(cfg.dependent_ids), 12
    mp = nv + nd + 2 * ni + 7
    rng = np.random.default_rng(21)
    Zc = 0.1 * rng.uniform(-1, 1, (1, pc.n))
    Zn = 0.1 * rng.uniform(-1, 1, (1, pc.n))
    G, J, J2 = oracle.step_link_host(pc, pn, Zc, Zn, mp, ni)
    # rows 0 .. nv + nd and the contact rows are the single-step block's
    Gs, Js = pc.eval_host(Zc)
    off = pc.block_offsets[-2]
    keep_rows = list(range(nv + nd)) + list(range(nv + nd + 2 * ni, mp))
    assert np.abs(G[0, keep_rows] - Gs[0, off:][keep_rows]).max() < 1e-12
    assert np.abs(J[0, keep_rows] - Js[0, off:][keep_rows]).max() < 1e-12
    # a multiple-step member (periodic = false in the reference) carries the tables and none of the rows
    stance_swap = "link"
    pl = (H.digit_problem(oracle, keep, N=N, stance="L", periodic=stance_swap) if robot == "digit" else
          H.legged_problem(oracle, robot, keep, N=N, stance="L", periodic=stance_swap))[0]
    assert pl.m == pn.m == pc.m - mp and pl.block_offsets == pc.block_offsets[:-1]
    for x, y in zip(oracle.step_link_host(pl, pn, Zc, Zn, mp, ni), (G, J, J2)):
        assert np.array_equal(x, y)
    h = 1e-6
    for c in list(range(0, pc.n, 9)) + list(range(pc.n - nv - nd, pc.n - nd, 5)):
        zp, zm = Zc.copy(), Zc.copy()
        zp[0, c] += h
        zm[0, c] -= h
        fd = (oracle.step_link_host(pc, pn, zp, Zn, mp, ni, A.EVAL_G)[0] - oracle.step_link_host(pc, pn, zm, Zn, mp, ni, A.EVAL_G)[0]) / (2 * h)
        assert np.abs(fd[0] - J[0, :, c]).max() < 1e-5 * (1 + np.abs(J[0, :, c]).max()), c
    r3 = slice(nv + nd, nv + nd + 2 * ni)
    for c in range(0, pc.n, 5):
        zp, zm = Zn.copy(), Zn.copy()
        zp[0, c] += h
        zm[0, c] -= h
        fd = (oracle.step_link_host(pc, pn, Zc, zp, mp, ni, A.EVAL_G)[0] - oracle.step_link_host(pc, pn, Zc, zm, mp, ni, A.EVAL_G)[0]) / (2 * h)
        assert np.abs(fd[0, r3] - J2[0, :, c]).max() < 1e-6, c
        assert np.abs(np.delete(fd[0], np.arange(nv + nd, nv + nd + 2 * ni))).max() < 1e-9
    # continuity: next step's first Bezier coefficient = q-, its velocity = v+ -> reset rows are zero
    K = cfg.degree + 1
    r = pc.constrained_id_host(Zc, N, nv, ni, derivatives=False)
    ind = cfg.independent_ids
    Zn2 = Zn.copy()
    vplus = Zc[0, pc.n - nv - nd:pc.n - nd]
    for t in range(ni):
        Zn2[0, t * K] = r["q"][0, N - 1, ind[t]]
        Zn2[0, t * K + 1] = Zn2[0, t * K] + vplus[ind[t]] * cfg.T / cfg.degree
    G2, _, _ = oracle.step_link_host(pc, pn, Zc, Zn2, mp, ni, A.EVAL_G)
    assert np.abs(G2[0, r3]).max() < 1e-12


def test_root_finding_status_report(oracle):
    """oracle_eval_batch_status: closable z report 0 and a handful of Newton steps; joint angles no linkage can follow end at the
    50-iteration cap and are flagged, as the reference's GSL loop would end (DigitDynamicsConstraints.cpp:344-367)."""
    keep = []
    p, _, _ = H.digit_problem(oracle, keep, N=3)
    rng = np.random.default_rng(0)
    Z = np.concatenate([0.1 * rng.uniform(-1, 1, (2, p.n)), 3.0 * rng.uniform(-1, 1, (2, p.n))])
    G, J, st, it = p.eval_status_host(Z)
    assert list(st[:2]) == [0, 0] and np.all(it[:2] <= 8) and np.all(it[:2] >= 1)
    assert np.all(st[2:] != 0) and np.all(it[2:] == 50)
    G2, J2 = p.eval_host(Z[:2])
    assert np.array_equal(G2, G[:2]) and np.array_equal(J2, J[:2])
