"""Parity of the fused eval_h kernel (and the KinovaOptimizer objective) against the oracle, through the C ABI.

Tolerance: 1e-9 relative, |hip - oracle| <= 1e-9 * (1 + |oracle|) on every entry of the 28-entry upper triangle; the
size-independent properties at the bench size are additivity in lambda and exact zeros for the linear blocks.
"""
import numpy as np
import pytest

from raptor_amd import _capi as A

import helpers as H
from test_gpu_parity import both, check

pytestmark = pytest.mark.gpu


def test_objective(oracle, hip, kinova):
    po, ph, keep = both(oracle, hip, lambda api, k: H.kinova_problem(api, kinova, k, N=16))
    rng = np.random.default_rng(5)
    q_des = rng.uniform(-4, 4, 7)
    for p in (po, ph):
        with pytest.raises(A.RaptorInvalidArgument):
            p.objective_host(np.zeros(7))
        p.set_objective(q_des, 9)
    Z = rng.uniform(-3, 3, (33, 7))
    Fo, Go = po.objective_host(Z)
    Fh, Gh = ph.objective_host(Z)
    check(Fh, Fo, 1e-13)
    check(Gh, Go, 1e-13)
    Fh2, none = ph.objective_host(Z, derivatives=False)
    assert none is None and np.array_equal(Fh2, Fh)


@pytest.mark.parametrize("gripper,friction", [(False, False), (True, True)])
def test_eval_h_matches_oracle(oracle, hip, kinova, kinova_friction, gripper, friction):
    model = kinova_friction if friction else kinova
    rng = np.random.default_rng(31 + gripper)
    q0, qd0, qdd0 = rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 7)
    po, ph, keep = both(oracle, hip, lambda api, k: H.kinova_problem(api, model, k, N=11, boxes=H.reach_boxes(), gripper=gripper,
                                                                      q0=q0, qd0=qd0, qdd0=qdd0, T=1.3))
    q_des = rng.uniform(-3, 3, 7)
    po.set_objective(q_des, 4)
    ph.set_objective(q_des, 4)
    B = 5
    Z = rng.uniform(-1, 1, (B, 7))
    lam = rng.uniform(-1, 1, (B, po.m))
    of = rng.uniform(0.5, 2, B)
    check(ph.eval_h_host(Z, lam, of), po.eval_h_host(Z, lam, of))
    check(ph.eval_h_host(Z, lam), po.eval_h_host(Z, lam))  # obj_factor absent: eval_hess_f skipped
    # block by block, and single rows (lambda = e_r isolates pg_pz_pz(r))
    off = po.block_offsets
    for blk in range(4):
        l2 = np.zeros_like(lam)
        l2[:, off[blk]:off[blk + 1]] = lam[:, off[blk]:off[blk + 1]]
        Hh, Ho = ph.eval_h_host(Z, l2), po.eval_h_host(Z, l2)
        check(Hh, Ho)
        if blk < 2:
            assert np.all(Hh == 0.0)
    rows = np.concatenate([rng.integers(off[2], off[3], 4), rng.integers(off[3], off[4], 6)])
    l3 = np.zeros((len(rows), po.m))
    l3[np.arange(len(rows)), rows] = 1.0
    Zr = np.repeat(Z[:1], len(rows), axis=0)
    check(ph.eval_h_host(Zr, l3), po.eval_h_host(Zr, l3))


def test_eval_h_both_launch_forms(oracle, hip, kinova):
    """B <= 192 runs seven workgroups per z plus a fixed-order sum, larger batches one workgroup per z: same numbers, and a z
    evaluated in either form gives bit-identical partial sums order-independently of its neighbours."""
    po, ph, keep = both(oracle, hip, lambda api, k: H.kinova_problem(api, kinova, k, N=70, boxes=H.reach_boxes()))  # two passes, ragged
    rng = np.random.default_rng(41)
    B = 200
    Z = rng.uniform(-1, 1, (B, 7))
    lam = rng.uniform(-1, 1, (B, po.m))
    po.set_objective(np.zeros(7), 3)
    ph.set_objective(np.zeros(7), 3)
    Hb = ph.eval_h_host(Z, lam, 1.0)            # batch form
    check(Hb[:24], po.eval_h_host(Z[:24], lam[:24], 1.0))
    Hs = ph.eval_h_host(Z[:7], lam[:7], 1.0)    # split form
    check(Hs, Hb[:7], 1e-12)
    assert np.array_equal(ph.eval_h_host(Z[3:4], lam[3:4], 1.0), Hs[3:4])  # deterministic, independent of the batch around it


@pytest.mark.parametrize("N", [7, 9, 33, 65, 127])
def test_eval_h_sample_counts(oracle, hip, kinova, N):
    """The torque kernel deals samples in groups of 8 over four wavefronts (and over seven workgroups in the small-batch form), the
    obstacle kernel in passes of 64: sample counts on both sides of every boundary, in both launch forms, against the oracle."""
    po, ph, keep = both(oracle, hip, lambda api, k: H.kinova_problem(api, kinova, k, N=N, boxes=H.reach_boxes()))
    rng = np.random.default_rng(100 + N)
    for p in (po, ph):
        p.set_objective(np.zeros(7), N // 2)
    for B in (2, 200):
        Z = rng.uniform(-1, 1, (B, 7))
        lam = rng.uniform(-1, 1, (B, po.m))
        of = rng.uniform(0.5, 1.5, B)
        Hh = ph.eval_h_host(Z, lam, of)
        check(Hh[:2], po.eval_h_host(Z[:2], lam[:2], of[:2]), 1e-11)
        if B > 192:  # the rest of the batch form against the split form of the same z
            Hs = np.concatenate([ph.eval_h_host(Z[k:k + 50], lam[k:k + 50], of[k:k + 50]) for k in range(0, B, 50)])
            check(Hh, Hs, 1e-12)


def test_eval_h_inside_boxes(oracle, hip, kinova):
    """Every sphere inside an obstacle: the nearest-face branch (BoxCollisionAvoidance.cpp:252-274), negative distances, one clamped
    axis; the first-minimum face / box rules must pick the same face as the oracle for the values to agree."""
    big = (np.array([[0.0, 0.0, 0.5], [0.1, 0.0, 0.4]]), np.array([[0.3, -0.2, 0.5], [0.0, 0.0, 0.0]]), np.array([[3.0, 3.0, 3.0], [2.5, 2.6, 2.7]]))
    po, ph, keep = both(oracle, hip, lambda api, k: H.kinova_problem(api, kinova, k, N=5, boxes=big, gripper=True))
    rng = np.random.default_rng(51)
    Z = rng.uniform(-1, 1, (3, 7))
    lam = np.zeros((3, po.m))
    lam[:, po.block_offsets[3]:] = rng.uniform(-1, 1, (3, po.m - po.block_offsets[3]))
    Go, _ = po.eval_host(Z, A.EVAL_G)
    assert np.all(Go[:, po.block_offsets[3]:] < 0.0)  # inside for every sphere and sample
    Gh, Jh = ph.eval_host(Z)
    check(Gh, Go)
    check(Jh, po.eval_host(Z)[1])
    check(ph.eval_h_host(Z, lam), po.eval_h_host(Z, lam))


def test_eval_h_edge_cases(oracle, hip, kinova):
    none = (np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)))
    po, ph, keep = both(oracle, hip, lambda api, k: H.kinova_problem(api, kinova, k, N=3, boxes=none))
    rng = np.random.default_rng(8)
    Z = rng.uniform(-1, 1, (2, 7))
    lam = rng.uniform(-1, 1, (2, po.m))
    check(ph.eval_h_host(Z, lam), po.eval_h_host(Z, lam))  # no obstacles: the obstacle rows keep zero Hessians
    assert ph.eval_h_host(np.zeros((0, 7)), np.zeros((0, po.m))).shape == (0, 28)
    with pytest.raises(A.RaptorInvalidArgument):
        ph.eval_h_host(Z, lam, 1.0)  # objective not set
    # far-away boxes of the reference's example, one sample
    po, ph, keep = both(oracle, hip, lambda api, k: H.kinova_problem(api, kinova, k, N=1))
    lam = rng.uniform(-1, 1, (2, po.m))
    check(ph.eval_h_host(Z, lam), po.eval_h_host(Z, lam))
    pid = H.id_problem(hip, kinova, keep)
    with pytest.raises(A.RaptorError):
        pid.eval_h_host(np.zeros(pid.n), np.zeros(pid.m))


def test_eval_h_full_size_properties(hip, kinova):
    """configs[1] size (N = 128, B = 512 here): H is linear in (obj_factor, lambda); the linear blocks contribute exact zeros."""
    keep = []
    p = H.kinova_problem(hip, kinova, keep, N=128, boxes=H.reach_boxes())
    p.set_objective(np.zeros(7), 64)
    rng = np.random.default_rng(77)
    B = 512
    Z = rng.uniform(-1, 1, (B, 7))
    l1, l2 = rng.uniform(-1, 1, (B, p.m)), rng.uniform(-1, 1, (B, p.m))
    H1, H2 = p.eval_h_host(Z, l1, 1.0), p.eval_h_host(Z, l2, 0.5)
    H12 = p.eval_h_host(Z, l1 + l2, 1.5)
    assert np.all(np.isfinite(H12))
    assert H.rel_err(H12, H1 + H2) < 1e-10
    off = p.block_offsets
    l0 = np.zeros((B, p.m))
    l0[:, :off[2]] = l1[:, :off[2]]
    assert np.all(p.eval_h_host(Z, l0) == 0.0)
