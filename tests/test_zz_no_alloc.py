"""raptor_eval_h_batch_noalloc (include/raptor_hip.h): the allocation-free variant of raptor_eval_h_batch.

Called through the C ABI and compared with raptor_eval_h_batch on the same inputs: same kernels, same summation order, so the
28 doubles per z must be bit-identical (the bar for this row is 1e-9 relative; equality is asserted because nothing may differ).
Two calls in a row on the library's one workspace, the second with a smaller batch: the contents the first call leaves behind
must not leak into the second.
"""
import numpy as np
import pytest
import torch

from raptor_amd import _capi as A

import helpers as H

pytestmark = pytest.mark.gpu


def _run(p, fn, Z, lam, of):
    B = Z.shape[0]
    dZ, dl = torch.from_numpy(Z).cuda(), torch.from_numpy(lam).cuda()
    dof = torch.from_numpy(of).cuda() if of is not None else None
    dH = torch.full((B, 28), float("nan"), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    fn(dZ.data_ptr(), B, dof.data_ptr() if dof is not None else 0, dl.data_ptr(), dH.data_ptr(), st)
    torch.cuda.synchronize()
    return dH.cpu().numpy()


def test_noalloc_matches_eval_h_batch(oracle, hip, kinova):
    keep = []
    p = H.kinova_problem(hip, kinova, keep, N=70, boxes=H.reach_boxes(), gripper=True)  # two 64-sample passes, ragged
    p.set_objective(np.linspace(-1, 1, 7), 11)
    rng = np.random.default_rng(2025)
    for B, with_of in ((112, True), (5, False), (1, True)):  # the limit, then smaller batches on the same workspace
        Z = rng.uniform(-1, 1, (B, 7))
        lam = rng.uniform(-1, 1, (B, p.m))
        of = rng.uniform(0.5, 2.0, B) if with_of else None
        want = _run(p, p.eval_h_device, Z, lam, of)
        got = _run(p, p.eval_h_device_noalloc, Z, lam, of)
        assert np.all(np.isfinite(got))
        assert H.rel_err(got, want) <= 1e-9
        assert np.array_equal(got, want)
    # against the oracle too, on the last (single-problem) call
    po = H.kinova_problem(oracle, kinova, keep, N=70, boxes=H.reach_boxes(), gripper=True)
    po.set_objective(np.linspace(-1, 1, 7), 11)
    assert H.rel_err(got, po.eval_h_host(Z, lam, of)) <= 1e-9


def test_noalloc_limits(hip, kinova):
    keep = []
    p = H.kinova_problem(hip, kinova, keep, N=8)
    z = torch.zeros(113, 7, dtype=torch.float64, device="cuda")
    lam = torch.zeros(113, p.m, dtype=torch.float64, device="cuda")
    out = torch.zeros(113, 28, dtype=torch.float64, device="cuda")
    with pytest.raises(A.RaptorInvalidArgument):  # beyond the workspace: refused, nothing launched
        p.eval_h_device_noalloc(z.data_ptr(), 113, 0, lam.data_ptr(), out.data_ptr())
    p.eval_h_device_noalloc(z.data_ptr(), 0, 0, lam.data_ptr(), out.data_ptr())  # empty batch is fine
    pid = H.id_problem(hip, kinova, keep)
    with pytest.raises(A.RaptorError):
        pid.eval_h_device_noalloc(z.data_ptr(), 1, 0, lam.data_ptr(), out.data_ptr())


def test_noalloc_allocates_no_device_memory(hip, kinova):
    """The device's free memory (hipMemGetInfo) is the same before and after the call -- first call on a fresh problem included: the
    workspace address was resolved by raptor_kinova_problem_create, the kernels' code object by an earlier launch of the library."""
    keep = []
    warm = H.kinova_problem(hip, kinova, keep, N=8)
    rng = np.random.default_rng(7)
    B = 16
    _run(warm, warm.eval_h_device, rng.uniform(-1, 1, (B, 7)), rng.uniform(-1, 1, (B, warm.m)), None)  # loads the code object
    p = H.kinova_problem(hip, kinova, keep, N=8)  # a fresh problem: nothing of it has run yet
    dZ = torch.from_numpy(rng.uniform(-1, 1, (B, 7))).cuda()
    dl = torch.from_numpy(rng.uniform(-1, 1, (B, p.m))).cuda()
    dH = torch.empty(B, 28, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        p.eval_h_device_noalloc(dZ.data_ptr(), B, 0, dl.data_ptr(), dH.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert torch.isfinite(dH).all()


def test_violation_batch_allocates_no_device_memory(hip, kinova):
    """raptor_violation_batch is documented asynchronous: its bounds are on the device since the problem was created."""
    keep = []
    warm = H.kinova_problem(hip, kinova, keep, N=8, boxes=H.reach_boxes())
    B = 8
    G = torch.zeros(B, warm.m, dtype=torch.float64, device="cuda")
    viol = torch.empty(B, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    warm.violation_device(G.data_ptr(), B, viol.data_ptr(), st)  # (the library's code object is loaded by its first launch)
    p = H.kinova_problem(hip, kinova, keep, N=8, boxes=H.reach_boxes())  # a fresh problem: this is its first call
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    p.violation_device(G.data_ptr(), B, viol.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert torch.isfinite(viol).all()
