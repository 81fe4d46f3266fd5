"""One process per GPU: start the ranks of a single-node job.

`run_ranks` starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N <script> <argv>` as a CHILD process and
returns its exit code.  The caller must not have touched the GPU (no HIP call, no `torch.cuda.is_available()`).  To count the
GPUs before launching use `visible_gpu_count()` below, which reads the KFD topology from sysfs and never loads the HIP runtime
(`torch.cuda.device_count()` falls back to `hipGetDeviceCount` when its amdsmi query fails, which does initialise HIP).  The ranks
read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (`dist_env`).  The reference has no counterpart: it is a single-process library.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Dict, Optional, Sequence, Tuple


def visible_gpu_count(topology: str = "/sys/class/kfd/kfd/topology/nodes") -> Optional[int]:
    """GPUs this process would see, without touching the HIP runtime: KFD topology nodes with SIMDs, narrowed by the
    ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES lists.  None when the topology cannot be read (then the
    per-rank check in the ranks themselves is what reports a missing GPU)."""
    if not os.path.exists(os.path.dirname(topology)):
        return 0  # no KFD device at all: no AMD GPU driver in this environment
    try:
        nodes = sorted(os.listdir(topology), key=lambda x: int(x) if x.isdigit() else 1 << 30)
    except OSError:
        return None
    gpus = 0
    for n in nodes:
        try:
            props = dict(l.split(None, 1) for l in open(os.path.join(topology, n, "properties")).read().splitlines() if " " in l)
        except OSError:
            return None
        if int(props.get("simd_count", "0").strip() or 0) > 0:
            gpus += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is None:
            continue
        listed = [x for x in val.split(",") if x.strip() != ""]
        gpus = min(gpus, len(listed))
    return gpus


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(script: str, argv: Sequence[str], nproc: int, port: int) -> list:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]


def run_ranks(script: str, argv: Sequence[str], nproc: int, extra_env: Optional[Dict[str, str]] = None, timeout: Optional[float] = None) -> int:
    """Run `script argv` as `nproc` ranks on this node; returns the launcher's exit code (non-zero if any rank failed)."""
    if nproc < 1:
        raise ValueError("nproc must be positive")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    if extra_env:
        env.update(extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    return subprocess.call(launcher_command(script, argv, nproc, free_port()), env=env, timeout=timeout)


def dist_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) of this process; (0, 0, 1) outside a launcher."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
