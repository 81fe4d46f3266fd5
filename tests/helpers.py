"""Shared problem builders for the parity tests (same calls on the oracle and on the HIP library)."""
import numpy as np

from raptor_amd import _capi as A


def armour_traj(keep, N=16, T=1.0, nact=7, q0=None, qd0=None, qdd0=None, disc=A.TIME_CHEBYSHEV):
    z = np.zeros(nact)
    return A.make_traj_desc(keep, A.TRAJ_ARMOUR_BEZIER, T, N, nact, 5, disc,
                            q0 if q0 is not None else z, qd0 if qd0 is not None else z,
                            qdd0 if qdd0 is not None else z)


def example_boxes():
    """KinovaExample.cpp:20-33: five 0.1^3 boxes far away at (10, 0, 0.3 k)."""
    c = np.array([[10.0, 0.0, 0.3 * i] for i in range(5)])
    return c, np.zeros((5, 3)), np.full((5, 3), 0.1)


def reach_boxes(seed=20250524, O=10):
    """ten boxes inside the arm's reach so every distance branch is exercised."""
    rng = np.random.default_rng(seed)
    c = np.column_stack([rng.uniform(-0.8, 0.8, O), rng.uniform(-0.8, 0.8, O), rng.uniform(0.0, 1.0, O)])
    rpy = rng.uniform(-np.pi, np.pi, (O, 3))
    size = rng.uniform(0.05, 0.3, (O, 3))
    return c, rpy, size


def kinova_problem(api, model, keep, N=16, boxes=None, gripper=False, buffer=0.0, **traj_kw):
    mh = api.model_create(model)
    keep.append(mh)
    t = armour_traj(keep, N=N, **traj_kw)
    c, rpy, size = boxes if boxes is not None else example_boxes()
    return A.ProblemHandle.kinova(api, mh, t, c, rpy, size, include_gripper=gripper, collision_buffer=buffer)


def id_problem(api, model, keep, ttype=A.TRAJ_POLYNOMIAL, T=2.0, N=10, degree=3, disc=A.TIME_CHEBYSHEV, **kw):
    mh = api.model_create(model)
    keep.append(mh)
    t = A.make_traj_desc(keep, ttype, T, N, model.nv, degree, disc, **kw)
    return A.ProblemHandle.inverse_dynamics(api, mh, t)


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b)))) if a.size else 0.0


def legged_problem(api, which, keep, N=14, stance="L", T_des=None, degree=None, customized=False, periodic=False,
                   q0=None, q_d0=None):
    """Talos / Unitree-G1 single-step problem (TalosSingleStepOptimizer.cpp:17-124, G1 twin)."""
    import os

    from raptor_amd import robots

    robots_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "robots")
    if which == "talos":
        m = robots.talos_model(os.path.join(robots_dir, "talos_reduced_armfixed_floatingbase.urdf"))
        cfg = robots.talos_config(m, stance)
    else:
        m = robots.g1_model(os.path.join(robots_dir, "g1_22dof_floatingbase.urdf"))
        cfg = robots.g1_config(m, stance)
    if T_des is not None:
        cfg.stance_foot_T_des_R, cfg.stance_foot_T_des_p = T_des
    if degree is not None:
        cfg.degree = degree
    if customized:
        cfg.gait = (robots.talos_gait if which == "talos" else robots.g1_gait)(m, stance)
    if periodic:
        cfg.period = dict((robots.talos_period if which == "talos" else robots.g1_period)(m, stance))
        if periodic == "link":
            cfg.period["enabled"] = 2  # RAPTOR_PERIOD_LINK_ONLY
    mh = api.model_create(m)
    keep.append(mh)
    t = A.make_traj_desc(keep, A.TRAJ_BEZIER, cfg.T, N, len(cfg.independent_ids), cfg.degree, A.TIME_CHEBYSHEV, q0=q0, q_d0=q_d0)
    return A.ProblemHandle.legged(api, mh, t, cfg), m, cfg


def digit_problem(api, keep, N=14, stance="L", T_des=None, degree=None, customized=False, periodic=False, q0=None, q_d0=None):
    """Digit-v3 single-step problem (DigitSingleStepOptimizer.cpp:17-124)."""
    import os

    from raptor_amd import robots

    robots_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "robots")
    m = robots.digit_model(os.path.join(robots_dir, "digit-v3-armfixedspecific-floatingbase-springfixed.urdf"))
    cfg = robots.digit_config(m, stance)
    if T_des is not None:
        cfg.stance_foot_T_des_R, cfg.stance_foot_T_des_p = T_des
    if degree is not None:
        cfg.degree = degree
    if customized:
        cfg.gait = robots.digit_gait()
    if periodic:
        cfg.period = {"enabled": 2} if periodic == "link" else {}
    mh = api.model_create(m)
    keep.append(mh)
    t = A.make_traj_desc(keep, A.TRAJ_BEZIER, cfg.T, N, 12, cfg.degree, A.TIME_CHEBYSHEV, q0=q0, q_d0=q_d0)
    return A.ProblemHandle.digit(api, mh, t, cfg), m, cfg
