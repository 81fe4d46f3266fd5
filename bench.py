#!/usr/bin/env python3
"""Headline benchmark: RNEA + d(tau)/dz sample-evals/s (fp64) on the Kinova batched configuration.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

A *step* is one pass of the hot path (eval_g + eval_jac_g, i.e. raptor_eval_batch with G|JAC) over one
batch of B = 4096 decision vectors x N = 128 time samples = 524 288 sample-evals, inputs and outputs
resident in HBM (BASELINE.json configs[1]).  With N GPUs every rank evaluates its own batch of 4096
(multi-start problems are independent: weak scaling, no data-path collective).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fused Kinova evaluation kernel),
timed live with HIP events recorded on the evaluation stream inside the timed region;
`cpu_baseline` is the CPU oracle (a restatement of the reference: kind "port") timed on one host core
over a bounded sample of the same batch.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak

N_SAMPLES = 128
BATCH = 4096
# algorithmic bytes per sample-eval = rows written once x (7 Jacobian columns + 1 value) x 8 B
ROWS = 21 + 17  # JointLimits 7 + VelocityLimits 7 + TorqueLimits 7 rows + one row per collision sphere, per sample
BYTES_ID = ROWS * 8 * 8   # 2432 B per sample-eval, the whole evaluation is one fused kernel
# algorithmic flops per sample-eval: RNEA 550*nv + analytic derivatives ~7e3 + contraction 2058
# + sphere FK / box distances ~25e3
FLOPS_ID = 550 * 7 + 7000 + 2058 + 25000


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import raptor_amd
    from raptor_amd import _capi as A
    import helpers as H

    hip = raptor_amd.api()  # raises if the HIP library is missing: there is no fallback
    hip.init(local_rank)
    model = raptor_amd.load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
    keep = []
    boxes = H.reach_boxes()
    prob = H.kinova_problem(hip, model, keep, N=N_SAMPLES, boxes=boxes)
    B = args.batch
    gen = torch.Generator(device="cpu").manual_seed(20250523 + 1 + rank)
    Z = (torch.rand(B, prob.n, generator=gen, dtype=torch.float64) * 2 - 1).cuda()
    G = torch.empty(B, prob.m, dtype=torch.float64, device="cuda")
    J = torch.empty(B, prob.m, prob.n, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    flags = A.EVAL_G | A.EVAL_JAC

    def step():
        prob.eval_device(Z.data_ptr(), B, flags, G.data_ptr(), J.data_ptr(), stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    prob.profile_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    n_prof, id_ms, obs_ms = prob.profile_end()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    sample_evals = B * N_SAMPLES
    value = world * sample_evals * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    if rank == 0:
        assert bool(torch.isfinite(G).all()) and bool(torch.isfinite(J).all())
        id_s = id_ms * 1e-3
        out = {
            "metric": "RNEA+dtau/dz sample-evals/sec (fp64)",
            "value": value,
            "unit": "sample-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "Kinova-gen3 7-DoF ArmourBezier, 128 Chebyshev samples x 4096 random z in U(-1,1)^7 per GPU, "
                            "eval_g+eval_jac_g (joint/velocity/torque limits + 17 spheres vs 10 boxes in reach), no solver",
                "batch_per_gpu": B, "samples": N_SAMPLES, "n": prob.n, "m": prob.m,
                "flags": "G|JAC", "parallelism": f"multi-start x{world} (independent batches, no collective)",
            },
            "eval_jac_g_latency_ms_batch": ms_per_step,
            "roofline": {
                "kernel": "kinova_fused_kernel<true> (trajectory + RNEA + d tau/dz + limit rows + sphere/box rows)",
                "bound": "hbm",
                "achieved": sample_evals * BYTES_ID / id_s / 1e9 if id_s > 0 else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (sample_evals * BYTES_ID / id_s / 1e9) / HBM_PEAK_GBS if id_s > 0 else None,
                "traffic": pmc_traffic(sample_evals * BYTES_ID),
                "kernel_ms": id_ms,
                "launches_timed": n_prof,
                "algorithmic_bytes_per_launch": sample_evals * BYTES_ID,
            },
            "roofline_fp64": {
                "kernel": "kinova_fused_kernel<true>",
                "bound": "fp64-valu",
                "achieved": sample_evals * FLOPS_ID / id_s / 1e12 if id_s > 0 else None,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (sample_evals * FLOPS_ID / id_s / 1e12) / FP64_PEAK_TFLOPS if id_s > 0 else None,
                "algorithmic_flops_per_sample_eval": FLOPS_ID,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, boxes, Z.cpu().numpy(), args.cpu_seconds, G, J, prob)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(algorithmic_bytes):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/pmc_latest.json,
    collected with this same command in separate WRITE_SIZE / FETCH_SIZE runs); None when no such file is present."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        k = d["kinova_fused_kernel<true>"]
        return k["WRITE_SIZE_bytes"] + k["FETCH_SIZE_bytes"]
    except Exception:
        return None


def cpu_baseline(model, boxes, Zs, budget_s, G, J, prob):
    """The CPU oracle (restatement of the reference's serial per-sample loop), one host core, on the first few z
    of the same batch; also used as a last check of the GPU results on exactly those z."""
    import numpy as np
    from raptor_amd import _capi as A
    import helpers as H

    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    keep = []
    po = H.kinova_problem(oracle, model, keep, N=N_SAMPLES, boxes=boxes)
    t0 = time.perf_counter()
    po.eval_host(Zs[:1])
    per_z = time.perf_counter() - t0
    nz = int(max(1, min(len(Zs), budget_s / max(per_z, 1e-6))))
    t0 = time.perf_counter()
    Go, Jo = po.eval_host(Zs[:nz])
    dt = time.perf_counter() - t0
    eg = H.rel_err(G[:nz].cpu().numpy(), Go)
    ej = H.rel_err(J[:nz].cpu().numpy(), Jo)
    if not (eg < 1e-9 and ej < 1e-9):
        raise SystemExit(f"GPU result differs from the oracle: g {eg:.2e} pg_pz {ej:.2e}")
    return {
        "value": nz * N_SAMPLES / dt,
        "unit": "sample-evals/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {nz} of the 4096 z of the same batch (x128 samples), eval_g+eval_jac_g, oracle/liboracle.so "
                  f"(-O2, serial like the reference's per-sample loop); GPU rows for those z checked against it: "
                  f"rel err g {eg:.1e}, pg_pz {ej:.1e}",
        "seconds": dt,
    }


if __name__ == "__main__":
    main()
