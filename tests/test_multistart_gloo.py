"""World-size-2 rehearsal of the multi-start path on CPU (gloo).  The evaluator here is the CPU oracle -- the
sharding, padding and gather logic under test is host code and identical for RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raptor_amd import _capi as A  # noqa: E402
from raptor_amd.multistart import evaluate_sharded, select_sharded, shard_range, violation  # noqa: E402
from raptor_amd.urdf import load_urdf  # noqa: E402


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 8, 9, 8192):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) == -(-total // world) if total else True
    assert [shard_range(8192, 8, r) for r in (0, 7)] == [(0, 1024), (7168, 8192)]
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_violation_rule():
    lb = torch.tensor([0.0, -1.0])
    ub = torch.tensor([1.0, 1.0])
    G = torch.tensor([[0.5, 0.0], [-0.2, 0.0], [0.5, 3.0]], dtype=torch.float64)
    assert torch.allclose(violation(G, lb, ub), torch.tensor([0.0, 0.2, 2.0], dtype=torch.float64))


def _worker(rank, world, port, B, out_dir):
    import helpers as H

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    model = load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
    keep = []
    p = H.kinova_problem(oracle, model, keep, N=4, boxes=H.reach_boxes())
    lb, ub = (torch.from_numpy(x) for x in p.bounds())
    Z = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (B, 7)))

    def evaluate(Zb):
        G, _ = p.eval_host(Zb.numpy(), A.EVAL_G)
        return torch.from_numpy(G)

    viol, feas, G = evaluate_sharded(evaluate, Z, lb, ub, gather_g=True)
    p.set_objective(np.full(7, 0.2), 2)
    sel = select_sharded(evaluate, lambda Zb: torch.from_numpy(p.objective_host(Zb.numpy(), False)[0]), 0.05 * Z, lb, ub)
    torch.save({"viol": viol, "feas": feas, "G": G, "sel": sel}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 8])
def test_two_ranks_match_single_process(tmp_path, B):
    import helpers as H

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    model = load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
    keep = []
    p = H.kinova_problem(oracle, model, keep, N=4, boxes=H.reach_boxes())
    Z = np.random.default_rng(5).uniform(-1, 1, (B, 7))
    G, _ = p.eval_host(Z, A.EVAL_G)
    lb, ub = (torch.from_numpy(x) for x in p.bounds())
    want = violation(torch.from_numpy(G), lb, ub)
    p.set_objective(np.full(7, 0.2), 2)
    Gs, _ = p.eval_host(0.05 * Z, A.EVAL_G)
    vs = violation(torch.from_numpy(Gs), lb, ub)
    fs = torch.from_numpy(p.objective_host(0.05 * Z, False)[0])
    ok = vs <= 1e-4
    want_best = int(torch.argmin(torch.where(ok, fs, torch.full_like(fs, float("inf"))))) if bool(ok.any()) else -1
    for r in (r0, r1):
        assert torch.equal(r["sel"]["cost"], fs) and torch.equal(r["sel"]["violation"], vs) and r["sel"]["best"] == want_best
        assert r["G"].shape == (B, p.m)
        assert torch.equal(r["G"], torch.from_numpy(G))
        assert torch.equal(r["viol"], want)
        assert torch.equal(r["feas"], want <= 1e-4)
