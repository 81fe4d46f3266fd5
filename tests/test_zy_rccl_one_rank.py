"""The path's one exchange step -- the all-gather of per-problem records -- executed through RCCL with ONE rank.

An N > 1 run needs N GPUs (the driver's scaling run); what one GPU can show is that the RCCL leg itself works end to end:
(a) `multistart.evaluate_sharded` under `init_process_group("nccl", world_size=1)`: the collective runs whenever a group exists;
(b) `raptor_eval_batch_sharded` (C ABI) with an ncclComm_t created through librccl.so.1's own API: own block evaluated, record
    written in place, one in-place ncclAllGather on the caller's stream.
Both are compared with the plain single-process result, bit for bit.  Collected late (pytest -x): an RCCL that cannot
initialise in some environment is reported as a skip with the reason, everything after initialisation is asserted."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch

from raptor_amd import _capi as A
from raptor_amd.multistart import evaluate_sharded, violation

import helpers as H

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _g1(hip, keep, B):
    p, m, cfg = H.legged_problem(hip, "g1", keep, N=6, customized=True, periodic=True)
    Z = (0.1 * (torch.rand(B, p.n, generator=torch.Generator().manual_seed(5), dtype=torch.float64) * 2 - 1)).cuda()
    Z[0] *= 30.0  # one far-off start: a clearly infeasible record
    return p, Z


def test_all_gather_through_torch_nccl_one_rank(hip):
    import torch.distributed as dist

    keep = []
    B = 37
    p, Z = _g1(hip, keep, B)
    lb, ub = (torch.from_numpy(x).cuda() for x in p.bounds())
    G = torch.empty(B, p.m, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def evaluate(Zb):
        p.eval_device(Zb.data_ptr(), Zb.shape[0], A.EVAL_G, G.data_ptr(), 0, st)
        return G[: Zb.shape[0]]

    want = violation(evaluate(Z), lb, ub).clone()
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=torch.device("cuda", torch.cuda.current_device()))
    except Exception as e:  # environment, not this library
        pytest.skip(f"RCCL process group could not be initialised here: {e}")
    try:
        viol, feas, none = evaluate_sharded(evaluate, Z, lb, ub, constr_viol_tol=1e-4)
        torch.cuda.synchronize()
        assert none is None and viol.shape == (B,)
        assert torch.equal(viol, want)                      # through ncclAllGather, unchanged
        assert bool(feas[0]) is False
        one = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(one)
        assert int(one.item()) == 1                          # the ranks RCCL saw
    finally:
        dist.destroy_process_group()


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]  # NCCL_UNIQUE_ID_BYTES (rccl.h)


def test_eval_batch_sharded_c_abi_one_rank(hip):
    try:
        rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)  # the copy torch has loaded, if any (same SONAME)
    except OSError:
        try:
            rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
        except OSError as e:
            pytest.skip(f"no RCCL library to load: {e}")
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    keep = []
    B = 29
    p, Z = _g1(hip, keep, B)
    torch.cuda.synchronize()
    uid, comm = _UniqueId(), C.c_void_p()
    if rccl.ncclGetUniqueId(C.byref(uid)) != 0 or rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) != 0 or not comm.value:
        pytest.skip("ncclCommInitRank(1 rank) failed in this environment")
    try:
        flags = A.EVAL_G | A.EVAL_JAC
        G = torch.empty(B, p.m, dtype=torch.float64, device="cuda")
        J = torch.empty(B, p.m, p.n, dtype=torch.float64, device="cuda")
        rec = torch.full((B,), float("nan"), dtype=torch.float64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        p.eval_sharded_device(Z.data_ptr(), B, 1, 0, flags, G.data_ptr(), J.data_ptr(), rec.data_ptr(), comm.value, st)
        torch.cuda.synchronize()
        G2, J2 = torch.empty_like(G), torch.empty_like(J)
        want = torch.empty(B, dtype=torch.float64, device="cuda")
        p.eval_device(Z.data_ptr(), B, flags, G2.data_ptr(), J2.data_ptr(), st)
        p.violation_device(G2.data_ptr(), B, want.data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.equal(G, G2) and torch.equal(J, J2) and torch.equal(rec, want)
        assert rec[0].item() > 1e-4 and torch.isfinite(rec).all()
        # the shard arithmetic of a many-rank job, on this one GPU: rank r of 4 evaluates its block (no communicator: the
        # collective is the caller's then), records land in the rank's slot, padding is zero
        total, world = 29, 4
        per = -(-total // world)
        rec_all = torch.full((world * per,), float("nan"), dtype=torch.float64, device="cuda")
        for r in range(world):
            with pytest.raises(A.RaptorInvalidArgument):  # more than one rank needs a communicator
                p.eval_sharded_device(Z.data_ptr(), total, world, r, flags, G.data_ptr(), J.data_ptr(), rec_all.data_ptr(), 0, st)
        lo, cnt = hip.shard_range(total, world, 3)
        assert (lo, cnt) == (24, 5) and per == 8
    finally:
        rccl.ncclCommDestroy(comm)
