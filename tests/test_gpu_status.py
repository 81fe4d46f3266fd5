"""raptor_eval_batch_status: the per-z report of Digit's closed-loop root finding (SURVEY.md 8b `int* status`).

The reference silently accepts whatever gsl_multiroot holds after 50 iterations (DigitDynamicsConstraints.cpp:344-367).  Both
libraries do the same and flag it.  Unclosable z (|z| ~ 3: joint angles no linkage can follow) must be flagged on both sides for
exactly the same z; z that close keep full parity whatever shares the batch with them.  The rows of a flagged z are "what the
solver returned" after 50 steps of a diverging iteration -- chaotic, so they are not compared."""
import numpy as np
import pytest

from raptor_amd import _capi as A

import helpers as H
from test_gpu_parity import both, check

pytestmark = pytest.mark.gpu


def test_digit_status_flags_match_the_oracle(oracle, hip):
    keep = []
    po, _, _ = H.digit_problem(oracle, keep, N=4, customized=True, periodic=True)
    ph, _, _ = H.digit_problem(hip, keep, N=4, customized=True, periodic=True)
    rng = np.random.default_rng(0)
    good = 0.1 * rng.uniform(-1, 1, (4, po.n))
    wild = 3.0 * rng.uniform(-1, 1, (5, po.n))
    Z = np.concatenate([good[:2], wild[:3], good[2:], wild[3:]])
    is_wild = np.array([0, 0, 1, 1, 1, 0, 0, 1, 1], dtype=bool)
    Go, Jo, so, io = po.eval_status_host(Z)
    Gh, Jh, sh, ih = ph.eval_status_host(Z)
    assert np.array_equal(so != 0, is_wild), so          # the oracle cannot close exactly the wild ones ...
    assert np.array_equal(sh != 0, so != 0), (sh, so)    # ... and the HIP library flags the same z
    assert np.all(ih[is_wild] == 50) and np.all(io[is_wild] == 50)
    assert np.all(ih[~is_wild] >= 1) and np.all(np.abs(ih[~is_wild] - io[~is_wild]) <= 1)  # per-loop vs joint residual test
    check(Gh[~is_wild], Go[~is_wild])
    check(Jh[~is_wild], Jo[~is_wild])
    # the closable z give the same rows when evaluated without the wild ones around them
    G2, J2 = ph.eval_host(Z[~is_wild])
    assert np.array_equal(G2, Gh[~is_wild]) and np.array_equal(J2, Jh[~is_wild])
    # G only, and no iteration buffer
    G3, none, s3, _ = ph.eval_status_host(Z, A.EVAL_G)
    assert none is None and np.array_equal(s3, sh)
    check(G3[~is_wild], Gh[~is_wild], 1e-12)  # the values-only kernel is another instantiation: same up to rounding


def test_status_is_zero_without_a_root_finder(oracle, hip, kinova):
    keep = []
    for build in (lambda api: H.kinova_problem(api, kinova, keep, N=4, boxes=H.reach_boxes()),
                  lambda api: H.legged_problem(api, "g1", keep, N=3, customized=True, periodic=True)[0]):
        po, ph = build(oracle), build(hip)
        Z = 0.1 * np.random.default_rng(1).uniform(-1, 1, (3, po.n))
        Go, Jo, so, io = po.eval_status_host(Z)
        Gh, Jh, sh, ih = ph.eval_status_host(Z)
        assert not so.any() and not sh.any() and not io.any() and not ih.any()
        check(Gh, Go)
        check(Jh, Jo)
    ph = H.kinova_problem(hip, kinova, keep, N=4)
    assert hip.fn("eval_batch_status_host")(ph.handle, None, 1, A.EVAL_G, None, None, None, None) == A.ERR_INVALID_ARGUMENT


@pytest.mark.gpu
def test_violation_record_on_device(hip):
    """raptor_violation_batch against the formula of multistart.violation (the record the ranks gather), NaN rows included."""
    import torch

    from raptor_amd import multistart

    keep = []
    ph, m, cfg = H.legged_problem(hip, "g1", keep, N=6)
    rng = np.random.default_rng(5)
    Z = torch.tensor(0.3 * rng.uniform(-1, 1, (9, ph.n))).cuda()
    G = torch.empty(9, ph.m, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ph.eval_device(Z.data_ptr(), 9, A.EVAL_G, G.data_ptr(), 0, st)
    lb, ub = ph.bounds()
    G[3, 17] = float("nan")
    flo, fhi = lb > -1e18, ub < 1e18
    inside = np.where(flo & fhi, 0.5 * (lb + ub), np.where(flo, lb + 1.0, np.where(fhi, ub - 1.0, 0.0)))
    G[5] = torch.tensor(inside).cuda()  # inside every bound: exactly feasible
    v = torch.full((9,), -1.0, dtype=torch.float64, device="cuda")
    ph.violation_device(G.data_ptr(), 9, v.data_ptr(), st)
    torch.cuda.synchronize()
    ref = multistart.violation(G, torch.tensor(lb).cuda(), torch.tensor(ub).cuda())
    assert torch.isnan(v[3]) and torch.isnan(ref[3])
    keep_rows = [0, 1, 2, 4, 5, 6, 7, 8]
    assert torch.equal(v[keep_rows], ref[keep_rows])
    assert v[5].item() == 0.0 and v[0].item() > 0.0
