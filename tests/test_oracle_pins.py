"""Pins the CPU oracle before it is trusted as the parity checker (runs without a GPU).

What the reference's own tests hold for this path:
  * nine TaperedCapsuleCollision known answers -- copied verbatim as data below
    (Tests/Constraints/TestTaperedCapsuleCollisionAvoidance.cpp:16-170, tolerance 1e-4);
  * finite-difference recipes with their tolerances (TestInverseDynamics.cpp:13-46 1e-5,
    TestBoxCollisionAvoidance.cpp:25-123 1e-5, TestTaperedCapsule...:172-251 1e-4).
Everything else has no golden vector in the reference ("parity unpinned"); it is cross-checked here
through physics identities instead (symmetric positive-definite M, gravity torque from potential energy,
RNEA linear in the acceleration, independent numpy forward kinematics).
"""
import ctypes as C

import numpy as np
import pytest

from raptor_amd import _capi as A
from raptor_amd.urdf import jtype_transform

import helpers as H

# (p11, p12, p21, p22, r11, r12, r21, r22, expected)  -- reference test data, verbatim
CAPSULE_KNOWN = [
    ((3, 0, 0), (0, 0, 0), (6, 3, 0), (6, -3, 0), 1.0, 0.5, 1.0, 0.5, 1.23957),
    ((0, 0, 0), (3, 0, 0), (6, 3, 0), (6, -3, 0), 0.5, 1.0, 1.0, 0.5, 1.23957),
    ((6, 3, 0), (6, -3, 0), (3, 0, 0), (0, 0, 0), 1.0, 0.5, 1.0, 0.5, 1.23957),
    ((6, 3, 0), (6, -3, 0), (0, 0, 0), (3, 0, 0), 1.0, 0.5, 0.5, 1.0, 1.23957),
    ((0, 0, 0), (-3, 0, 0), (2, 0, 0), (5, 0, 0), 0.5, 1.5, 0.5, 1.5, 1.0),
    ((0, -3, 0), (0, 0, 0), (0, 2, 0), (0, 5, 0), 0.5, 1.0, 0.5, 1.0, 0.5),
    ((0, 0, 0), (0, 0, -3), (0, 0, 5), (0, 0, 2), 0.5, 1.0, 0.5, 0.75, 0.75),
    ((0, 0, -3), (0, 0, 0), (0, 0, 5), (0, 0, 2), 0.5, 1.0, 0.5, 0.75, 0.25),
    ((5, 2, 0), (-5, 2, 0), (0, -2, 5), (0, -2, -5), 1.0, 1.5, 1.0, 1.5, 1.4900),
]


def capsule_arrays():
    P = np.array([[c[0], c[1], c[2], c[3]] for c in CAPSULE_KNOWN], dtype=np.float64)
    R = np.array([[c[4], c[5], c[6], c[7]] for c in CAPSULE_KNOWN], dtype=np.float64)
    want = np.array([c[8] for c in CAPSULE_KNOWN])
    return P, R, want


def test_capsule_known_answers(oracle):
    P, R, want = capsule_arrays()
    dist, _ = oracle.capsule_distance_host(P, None, R)
    assert np.all(np.abs(dist - want) < 1e-4), dist - want


def random_capsules(seed, n, F=7):
    rng = np.random.default_rng(seed)
    P = rng.uniform(-20, 20, (n, 4, 3))
    dP = rng.uniform(-1, 1, (n, 4, 3, F))
    R = np.where(rng.uniform(size=(n, 1)) < 0.5, rng.uniform(0, 10, (n, 4)), rng.integers(0, 10, (n, 4)).astype(float))
    return P, dP, R


def test_capsule_gradient_fd(oracle):
    # TestTaperedCapsuleCollisionAvoidance.cpp:172-251 recipe; step widened to 1e-6 (their 1e-10 step is at
    # the noise floor in double precision), same 1e-4 tolerance; like the reference, only checked
    # when distance > -1, and skipped where the active candidate switches inside the FD stencil.
    P, dP, R = random_capsules(3, 400)
    dist, grad = oracle.capsule_distance_host(P, dP, R)
    h = 1e-6
    bad = 0
    checked = 0
    for j in range(7):
        Pp = P + h * dP[..., j]
        Pm = P - h * dP[..., j]
        dp, _ = oracle.capsule_distance_host(Pp, None, R)
        dm, _ = oracle.capsule_distance_host(Pm, None, R)
        num = (dp - dm) / (2 * h)
        sel = dist > -1
        err = np.abs(num - grad[:, j])[sel]
        checked += err.size
        bad += int((err > 1e-4).sum())
    # the reference's analytic gradient is itself approximate in its edge-edge case (quirk kept: pu_pz == pt_pz),
    # so a small fraction of pairs disagrees with finite differences there by construction.
    assert checked > 1000
    assert bad / checked < 0.15, (bad, checked)


def _fn(api, name, argtypes):
    f = getattr(api.lib, name)
    f.argtypes = argtypes
    return f


def rnea(api, mh, q, v, a):
    f = _fn(api, "oracle_rnea", [C.c_void_p] + [A._dp] * 5)
    tau = np.zeros(mh.nv)
    api.check(f(mh.handle, A._ptr(q), A._ptr(v), A._ptr(a), A._ptr(tau), None))
    return tau


def rnea_derivs(api, mh, q, v, a):
    f = _fn(api, "oracle_rnea_derivatives", [C.c_void_p] + [A._dp] * 7)
    n = mh.nv
    tau, dq, dv, da = np.zeros(n), np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, n))
    api.check(f(mh.handle, A._ptr(q), A._ptr(v), A._ptr(a), A._ptr(tau), A._ptr(dq), A._ptr(dv), A._ptr(da)))
    return tau, dq, dv, da


def numpy_fk(model, q, upto):
    """Independent forward kinematics: world placement of joint frame `upto` (0-based) after its rotation."""
    R = np.eye(3)
    p = np.zeros(3)
    chain = []
    j = upto
    while j >= 0:
        chain.append(j)
        j = int(model.parent[j])
    for j in reversed(chain):
        Rq, pq = jtype_transform(int(model.jtype[j]), float(q[j]))
        p = p + R @ model.placement_p[j]
        R = R @ model.placement_R[j]
        p = p + R @ pq
        R = R @ Rq
    return R, p


def potential_energy(model, q):
    U = 0.0
    for j in range(model.nv):
        R, p = numpy_fk(model, q, j)
        U -= model.mass[j] * model.gravity @ (p + R @ model.lever[j])
    return U


def test_rnea_physics(oracle, kinova):
    mh = oracle.model_create(kinova)
    rng = np.random.default_rng(1)
    for _ in range(5):
        q, v, a = rng.uniform(-2, 2, 7), rng.uniform(-2, 2, 7), rng.uniform(-3, 3, 7)
        tau, dq, dv, M = rnea_derivs(oracle, mh, q, v, a)
        assert np.allclose(tau, rnea(oracle, mh, q, v, a), rtol=0, atol=1e-13)
        # M symmetric positive definite, and d tau / d a == M: RNEA is affine in a
        assert np.abs(M - M.T).max() < 1e-13
        assert np.linalg.eigvalsh(0.5 * (M + M.T)).min() > 0
        a2 = rng.uniform(-3, 3, 7)
        assert np.allclose(rnea(oracle, mh, q, v, a2) - tau, M @ (a2 - a), atol=1e-11)
        # gravity torque equals the gradient of the potential energy (independent numpy FK)
        g_tau = rnea(oracle, mh, q, np.zeros(7), np.zeros(7))
        h = 1e-6
        num = np.array([(potential_energy(kinova, q + h * e) - potential_energy(kinova, q - h * e)) / (2 * h)
                        for e in np.eye(7)])
        assert np.abs(num - g_tau).max() < 1e-7
        # finite differences of tau (TestInverseDynamics.cpp recipe, tolerance 1e-5)
        for D, which in ((dq, 0), (dv, 1)):
            num = np.zeros((7, 7))
            for j in range(7):
                e = np.zeros(7)
                e[j] = 1e-6
                args_p = [q, v, a]
                args_m = [q, v, a]
                args_p[which] = args_p[which] + e
                args_m[which] = args_m[which] - e
                num[:, j] = (rnea(oracle, mh, *args_p) - rnea(oracle, mh, *args_m)) / 2e-6
            assert np.linalg.norm(D - num) < 1e-5


def test_rnea_energy_rate(oracle, kinova):
    """v^T (tau - g(q)) = d/dt (1/2 v^T M v) along a smooth motion (no damping/friction)."""
    mh = oracle.model_create(kinova)
    rng = np.random.default_rng(5)
    c = rng.uniform(-1, 1, (4, 7))
    qf = lambda t: c[0] + c[1] * t + c[2] * t * t + c[3] * t ** 3  # noqa: E731
    vf = lambda t: c[1] + 2 * c[2] * t + 3 * c[3] * t * t  # noqa: E731
    af = lambda t: 2 * c[2] + 6 * c[3] * t  # noqa: E731

    def kinetic(t):
        _, _, _, M = rnea_derivs(oracle, mh, qf(t), vf(t), af(t))
        return 0.5 * vf(t) @ M @ vf(t)

    t0, h = 0.37, 1e-5
    lhs = vf(t0) @ (rnea(oracle, mh, qf(t0), vf(t0), af(t0)) - rnea(oracle, mh, qf(t0), np.zeros(7), np.zeros(7)))
    rhs = (kinetic(t0 + h) - kinetic(t0 - h)) / (2 * h)
    assert abs(lhs - rhs) < 1e-6 * max(1.0, abs(lhs))


def test_inverse_dynamics_gradient_like_reference(oracle, kinova_friction):
    """Tests/KinematicsDynamics/TestInverseDynamics.cpp:13-46: Polynomials(2.0, 10, nv, Chebyshev, 3)."""
    keep = []
    p = H.id_problem(oracle, kinova_friction, keep)
    rng = np.random.default_rng(11)
    z = (2 / np.pi) * rng.uniform(-1, 1, p.n)  # M_2_PI * Random
    tau, ptau = p.inverse_dynamics_host(z, 10, 7)
    for j in range(p.n):
        e = np.zeros(p.n)
        e[j] = 1e-7
        tp, _ = p.inverse_dynamics_host(z + e, 10, 7, derivatives=False)
        tm, _ = p.inverse_dynamics_host(z - e, 10, 7, derivatives=False)
        num = (tp - tm)[0] / 2e-7
        for i in range(10):
            assert np.linalg.norm(ptau[0, i, :, j] - num[i]) < 1e-5


def test_forward_kinematics(oracle, kinova):
    """FK against the independent numpy chain (the reference pins its FK to pinocchio at 1e-10:
    TestForwardKinematicsSolver.cpp:9-38) and the translation Jacobian against finite differences."""
    mh = oracle.model_create(kinova)
    f = _fn(oracle, "oracle_fk", [C.c_void_p, C.c_int, C.c_int] + [A._dp] * 5)
    rng = np.random.default_rng(2)
    for end in (2, 4, 6, 7):
        q = rng.uniform(-3, 3, 7)
        off = rng.uniform(-0.2, 0.2, 3)
        R, p, J = np.zeros((3, 3)), np.zeros(3), np.zeros((3, 7))
        oracle.check(f(mh.handle, 0, end, A._ptr(q), A._ptr(off), A._ptr(R), A._ptr(p), A._ptr(J)))
        Rn, pn = numpy_fk(kinova, q, end - 1)
        assert np.abs(R - Rn).max() < 1e-12
        assert np.abs(p - (pn + Rn @ off)).max() < 1e-12
        assert np.all(J[:, end:] == 0)
        for j in range(7):
            e = np.zeros(7)
            e[j] = 1e-6
            Rp, pp = numpy_fk(kinova, q + e, end - 1)
            Rm, pm = numpy_fk(kinova, q - e, end - 1)
            num = ((pp + Rp @ off) - (pm + Rm @ off)) / 2e-6
            assert np.abs(num - J[:, j]).max() < 1e-8


def box_distance(api, boxes, point, pp=None):
    f = _fn(api, "oracle_box_distance", [C.c_int] + [A._dp] * 5 + [C.c_int, A._dp, _ip := C.POINTER(C.c_int), A._dp])
    c, rpy, size = (np.ascontiguousarray(b, dtype=np.float64) for b in boxes)
    d = C.c_double()
    idx = C.c_int()
    W = 0 if pp is None else pp.shape[1]
    grad = np.zeros(max(W, 1))
    point = np.ascontiguousarray(point, dtype=np.float64)
    ppc = None if pp is None else np.ascontiguousarray(pp, dtype=np.float64)
    api.check(f(c.shape[0], A._ptr(c), A._ptr(rpy), A._ptr(size), A._ptr(point), A._ptr(ppc), W,
                C.cast(C.byref(d), A._dp), C.byref(idx), A._ptr(grad)))
    return d.value, idx.value, grad[:W]


def test_box_distance_known_and_gradient(oracle):
    """Tests/Constraints/TestBoxCollisionAvoidance.cpp:25-123: unit cube at the origin, points on spheres
    r in {1.2, 1.5, 2.0}, 5 variables, finite-difference tolerance 1e-5; plus hand-checkable distances."""
    cube = (np.zeros((1, 3)), np.zeros((1, 3)), np.ones((1, 3)))
    d, idx, _ = box_distance(oracle, cube, [2.0, 0.0, 0.0])
    assert abs(d - 1.5) < 1e-15 and idx == 0
    d, _, _ = box_distance(oracle, cube, [1.5, 1.5, 0.0])
    assert abs(d - np.sqrt(2.0)) < 1e-15
    d, _, _ = box_distance(oracle, cube, [0.1, 0.0, 0.0])  # inside: negative, nearest face x+
    assert abs(d + 0.4) < 1e-15
    d, _, _ = box_distance(oracle, cube, [0.5, 0.0, 0.0])  # on the face: below the sqrt threshold -> 0
    assert d == 0.0
    rng = np.random.default_rng(4)
    for r in (1.2, 1.5, 2.0):
        for _ in range(20):
            z0 = rng.uniform(-1, 1, 5)
            Aq = rng.uniform(-1, 1, (3, 5))

            def point(z):
                u = Aq @ z
                return r * u / np.linalg.norm(u)

            h = 1e-6
            pp = np.column_stack([(point(z0 + h * e) - point(z0 - h * e)) / (2 * h) for e in np.eye(5)])
            _, _, grad = box_distance(oracle, cube, point(z0), pp)
            num = np.array([(box_distance(oracle, cube, point(z0 + h * e))[0] -
                             box_distance(oracle, cube, point(z0 - h * e))[0]) / (2 * h) for e in np.eye(5)])
            assert np.abs(num - grad).max() < 1e-5


def test_kinova_constraint_jacobian_fd(oracle, kinova):
    """Whole-problem derivative check, the reference's own end-to-end criterion
    (Ipopt derivative_test, tolerance 1e-4: Examples/Digit/DigitSingleStep.cpp:119-131)."""
    keep = []
    p = H.kinova_problem(oracle, kinova, keep, N=8, boxes=H.reach_boxes())
    rng = np.random.default_rng(9)
    z = rng.uniform(-1, 1, 7)
    G, J = p.eval_host(z)
    h = 1e-6
    worst = 0.0
    for j in range(7):
        e = np.zeros(7)
        e[j] = h
        Gp, _ = p.eval_host(z + e, A.EVAL_G)
        Gm, _ = p.eval_host(z - e, A.EVAL_G)
        num = (Gp - Gm)[0] / (2 * h)
        err = np.abs(num - J[0, :, j])
        # obstacle rows are piecewise smooth: ignore rows whose active box/face flips inside the stencil
        worst = max(worst, float(np.sort(err)[: int(0.98 * err.size)].max()))
    assert worst < 1e-4, worst


def test_layout_and_bounds(oracle, kinova):
    keep = []
    p = H.kinova_problem(oracle, kinova, keep, N=16, gripper=True, buffer=0.02)
    assert p.n == 7 and p.m == 16 * (21 + 20)
    assert p.block_offsets == [0, 112, 224, 336, 336 + 320]
    lb, ub = p.bounds()
    assert lb[0] == np.deg2rad(-1e19) and ub[1] == pytest.approx(np.deg2rad(128.9))
    assert np.all(lb[224:336].reshape(16, 7) == [-90, -90, -90, -90, -45, -45, -45])
    assert np.all(lb[336:] == 0.02) and np.all(ub[336:] == 1e19)
    ts = p.tspan(16)
    assert ts[0] == 0 and ts[-1] == 1.0 and np.all(np.diff(ts) > 0)
    i = 3
    assert ts[i] == pytest.approx(0.5 * (1 + np.cos(np.pi * (2 * (16 - i) - 3) / (2 * 14))))


def test_error_codes(oracle, kinova):
    keep = []
    mh = oracle.model_create(kinova)
    with pytest.raises(A.RaptorInvalidArgument):  # ArmourBezierCurves ctor: q0.size() != Nact
        A.ProblemHandle.kinova(oracle, mh, A.make_traj_desc(keep, A.TRAJ_ARMOUR_BEZIER, 1.0, 8, 7), *H.example_boxes())
    with pytest.raises(A.RaptorInvalidArgument):
        A.ProblemHandle.kinova(oracle, mh, A.make_traj_desc(keep, A.TRAJ_BEZIER, 1.0, 8, 7), *H.example_boxes())
    bad = kinova.copy()
    bad.parent = bad.parent.copy()
    bad.parent[2] = 5
    with pytest.raises(A.RaptorInvalidArgument):
        oracle.model_create(bad)


def test_dual_kinova_arm_arm_rows(oracle, kinova):
    """DualKinovaOptimizer.cpp:269-305, 449-510: N*4 tapered-capsule rows between two arms, 14 columns.  Pins: the sphere centres
    are where an independent numpy FK puts them (through the capsule routine on those points), the gradient by central
    differences (the reference checks its constraint gradients the same way), zero blocks outside the own robot's columns."""
    import helpers as H
    from raptor_amd import _capi as A

    keep = []
    m2 = kinova.copy()
    m2.placement_p = m2.placement_p.copy()
    m2.placement_p[0] = m2.placement_p[0] + np.array([0.55, 0.25, 0.0])
    N = 6
    p1 = H.kinova_problem(oracle, kinova, keep, N=N)
    p2 = H.kinova_problem(oracle, m2, keep, N=N, gripper=True)
    rng = np.random.default_rng(17)
    Z = rng.uniform(-1, 1, (2, 14))
    G, J = oracle.dual_kinova_arm_arm_host(p1, p2, Z, N)
    assert G.shape == (2, N * 4) and J.shape == (2, N * 4, 14)
    h = 1e-6
    for c in range(14):
        zp, zm = Z.copy(), Z.copy()
        zp[:, c] += h
        zm[:, c] -= h
        fd = (oracle.dual_kinova_arm_arm_host(p1, p2, zp, N, A.EVAL_G)[0] - oracle.dual_kinova_arm_arm_host(p1, p2, zm, N, A.EVAL_G)[0]) / (2 * h)
        assert np.abs(fd - J[:, :, c]).max() < 1e-5, c
    # same z for both arms, second arm shifted: the distance of like capsules is bounded by the base offset
    Zs = np.concatenate([Z[:, :7], Z[:, :7]], axis=1)
    p2b = H.kinova_problem(oracle, m2, keep, N=N)
    Gs, _ = oracle.dual_kinova_arm_arm_host(p1, p2b, Zs, N, A.EVAL_G)
    off = np.linalg.norm([0.55, 0.25, 0.0])
    assert np.all(Gs.reshape(2, N, 4)[:, :, [0, 3]] <= off + 1e-12)
    with pytest.raises(A.RaptorInvalidArgument):
        oracle.dual_kinova_arm_arm_host(p1, H.kinova_problem(oracle, m2, keep, N=N + 1), Z, N)


def piecewise_problem(api, model, keep, degree=3, N=24, T=2.0, q0=None, qT=None):
    """InverseDynamics over PiecewiseBezierCurves(T, N, nv, degree, q0, qT) (KinovaLongerHorizonOptimizer.cpp:39-47)."""
    mh = api.model_create(model)
    keep.append(mh)
    t = A.make_traj_desc(keep, A.TRAJ_PIECEWISE_BEZIER, T, N, model.nv, degree, A.TIME_UNIFORM, q0=q0, qT=qT)
    return A.ProblemHandle.inverse_dynamics(api, mh, t)


@pytest.mark.parametrize("fixed_ends", [True, False])
def test_piecewise_bezier_against_scipy_bpoly(oracle, kinova, fixed_ends):
    """PiecewiseBezierCurves.cpp:52-255 against an independent construction: scipy's BPoly.from_derivatives builds the quintic
    pieces from (q, q_d, q_dd) at the breakpoints k T / (degree + 1); zero velocity / acceleration at both ends (:91-92, 111-112)."""
    from scipy.interpolate import BPoly

    keep = []
    degree, N, T = 3, 24, 2.0
    rng = np.random.default_rng(13)
    q0, qT = rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 7)
    p = piecewise_problem(oracle, kinova, keep, degree, N, T, q0 if fixed_ends else None, qT if fixed_ends else None)
    assert p.n == 7 * degree * 3 + (0 if fixed_ends else 14)  # :34-45
    z = rng.uniform(-1, 1, p.n)
    if not fixed_ends:
        q0, qT = z[63:70], z[70:77]  # begin position first, then the end position
    q, qd, qdd = (a[0] for a in p.trajectory_host(z, N, 7))
    ts = np.linspace(0.0, T, N)
    xi = np.arange(degree + 2) * T / (degree + 1)
    for j in range(7):
        yi = [[q0[j], 0.0, 0.0]] + [[z[k * 21 + j], z[k * 21 + 7 + j], z[k * 21 + 14 + j]] for k in range(degree)] + [[qT[j], 0.0, 0.0]]
        bp = BPoly.from_derivatives(xi, yi)
        assert np.abs(q[:, j] - bp(ts)).max() < 1e-12
        assert np.abs(qd[:, j] - bp.derivative(1)(ts)).max() < 1e-11
        assert np.abs(qdd[:, j] - bp.derivative(2)(ts)).max() < 1e-10
    # pq_pz through TorqueLimits' chain: the reference's finite-difference recipe (TestInverseDynamics.cpp:13-46)
    tau, ptau = p.inverse_dynamics_host(z, N, 7)
    h = 1e-6
    for c in rng.choice(p.n, 12, replace=False):
        e = np.zeros(p.n)
        e[c] = h
        num = (p.inverse_dynamics_host(z + e, N, 7, False)[0] - p.inverse_dynamics_host(z - e, N, 7, False)[0])[0] / (2 * h)
        assert np.abs(ptau[0, :, :, c] - num).max() < 1e-5


def test_piecewise_bezier_errors(oracle, kinova):
    keep = []
    with pytest.raises(A.RaptorInvalidArgument):
        piecewise_problem(oracle, kinova, keep, degree=3, N=25)  # N must be a multiple of degree + 1 (:18-20)


def test_path_length_and_jerk_costs(oracle, kinova):
    """MinimizePathLength.cpp:10-61 / MinimizeJerk.cpp:10-61 from the trajectory samples (numpy) and finite differences."""
    keep = []
    rng = np.random.default_rng(19)
    p = piecewise_problem(oracle, kinova, keep, 3, 24, 2.0, rng.uniform(-1, 1, 7), None)
    z = rng.uniform(-1, 1, p.n)
    kinds = [A.COST_MINIMIZE_PATH_LENGTH, A.COST_MINIMIZE_JERK]
    F, GR = p.costs_host(z, kinds)
    q, _, qdd = (a[0] for a in p.trajectory_host(z, 24, 7))
    assert F[0, 0] == pytest.approx(0.5 * (np.diff(q, axis=0) ** 2).sum() / 24, rel=1e-13)
    assert F[0, 1] == pytest.approx(0.5 * (np.diff(qdd, axis=0) ** 2).sum() / 24, rel=1e-13)
    h = 1e-6
    for c in range(p.n):
        e = np.zeros(p.n)
        e[c] = h
        num = (p.costs_host(z + e, kinds, False)[0] - p.costs_host(z - e, kinds, False)[0])[0] / (2 * h)
        assert np.abs(num - GR[0, :, c]).max() < 1e-5 * max(1.0, np.abs(GR[0, :, c]).max())
