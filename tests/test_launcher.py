"""bench.py's self-launch path and the measured multi-start sweep, rehearsed on the CPU (gloo, world size 2)."""
import json
import os
import socket
import subprocess
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
from raptor_amd.launch import dist_env, launcher_command, run_ranks  # noqa: E402
from raptor_amd.multistart import timed_sweep, violation  # noqa: E402
from raptor_amd.urdf import load_urdf  # noqa: E402


def test_run_ranks_starts_one_process_per_rank(tmp_path):
    rc = run_ranks(os.path.join(ROOT, "tests", "_rank_probe.py"), [str(tmp_path)], 2, timeout=300)
    assert rc == 0
    recs = [json.load(open(tmp_path / f"probe{r}.json")) for r in range(2)]
    assert [r["rank"] for r in recs] == [0, 1] and [r["local_rank"] for r in recs] == [0, 1]
    assert all(r["world"] == 2 and r["ranks_seen"] == 2 and r["master"] == "127.0.0.1" for r in recs)


def test_run_ranks_reports_a_failed_rank(tmp_path):
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    assert run_ranks(str(bad), [], 2, timeout=300) != 0


def test_launcher_command_shape_and_env():
    cmd = launcher_command("bench.py", ["--gpus", "4"], 4, 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and cmd[-3:] == ["bench.py", "--gpus", "4"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert dist_env() == (0, 0, 1) or "WORLD_SIZE" in os.environ


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than two GPUs")
def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` with no launcher and fewer than 2 GPUs must fail loudly, not print an n_gpus = 1 line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and "n_gpus" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in r.stderr


def _sweep_worker(rank, world, port, B, out_dir):
    import helpers as H

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    model = load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
    keep = []
    p = H.kinova_problem(oracle, model, keep, N=4, boxes=H.reach_boxes())
    lb, ub = (torch.from_numpy(x) for x in p.bounds())
    Z = torch.from_numpy(np.random.default_rng(11).uniform(-1, 1, (B, 7)))
    calls = []

    def evaluate(Zb):
        calls.append(Zb.shape[0])
        return torch.from_numpy(p.eval_host(Zb.numpy(), A.EVAL_G)[0])

    r = timed_sweep(evaluate, Z, lb, ub, steps=3, warmup=1)
    r["calls"] = calls
    torch.save(r, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_timed_sweep_two_ranks(tmp_path):
    import helpers as H

    B = 7  # ragged: blocks of 4 and 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sweep_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    model = load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
    keep = []
    p = H.kinova_problem(oracle, model, keep, N=4, boxes=H.reach_boxes())
    Z = np.random.default_rng(11).uniform(-1, 1, (B, 7))
    lb, ub = (torch.from_numpy(x) for x in p.bounds())
    want = violation(torch.from_numpy(p.eval_host(Z, A.EVAL_G)[0]), lb, ub)
    for r, block in ((r0, 4), (r1, 3)):
        assert r["ranks_seen"] == 2 and len(r["per_rank_s"]) == 2 and r["elapsed_s"] == max(r["per_rank_s"]) > 0
        assert torch.equal(r["violation"], want) and torch.equal(r["feasible"], want <= 1e-4)
        assert r["calls"] == [block] * 4  # one warm-up + three timed sweeps, own block only
    assert r0["per_rank_s"] == r1["per_rank_s"]


def test_shard_range_matches_c_abi():
    """raptor_shard_range (the C-ABI partition a C++ host uses) and multistart.shard_range are the same function."""
    import os

    from raptor_amd import _capi, multistart

    api = _capi.CApi(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raptor_amd", "libraptor_hip.so"), "raptor_")
    for total in (0, 1, 7, 8, 9, 1000, 8192):
        for world in (1, 2, 3, 8):
            covered = 0
            for rank in range(world):
                first, count = api.shard_range(total, world, rank)
                lo, hi = multistart.shard_range(total, world, rank)
                assert (first, first + count) == (lo, hi)
                covered += count
            assert covered == total
    import pytest

    with pytest.raises(_capi.RaptorInvalidArgument):
        api.shard_range(5, 2, 2)


def test_visible_gpu_count_reads_topology_without_hip(tmp_path, monkeypatch):
    """The parent of a self-launched run counts GPUs from the KFD topology (sysfs), never through the HIP runtime."""
    from raptor_amd.launch import visible_gpu_count

    nodes = tmp_path / "topology" / "nodes"
    for k, simd in enumerate((0, 1024, 1024, 0, 1024)):  # CPU nodes have no SIMDs
        d = nodes / str(k)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert visible_gpu_count(str(nodes)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert visible_gpu_count(str(nodes)) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert visible_gpu_count(str(nodes)) == 1
    assert visible_gpu_count(str(tmp_path / "absent" / "nodes")) == 0
