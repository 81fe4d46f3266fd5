#!/usr/bin/env python3
"""Headline benchmark: RNEA + d(tau)/dz sample-evals/s (fp64).

  python bench.py --gpus N --steps K --warmup W [--workload kinova|talos|g1|digit]
  (N > 1: launched by torch.distributed.run, one rank per GPU)

Default workload = BASELINE.json configs[1] (Kinova): a *step* is one pass of the hot path (eval_g + eval_jac_g, i.e.
raptor_eval_batch with G|JAC) over one batch of B = 4096 decision vectors x N = 128 time samples = 524 288
sample-evals, inputs and outputs resident in HBM.  With N GPUs every rank evaluates its own batch (multi-start
problems are independent: weak scaling, no data-path collective).  `--workload talos|g1` runs the legged single-step
problem (configs[2] / configs[4] shard: 1024 problems x 14 samples per GPU) the same way.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed live with HIP events recorded on the
evaluation stream inside the timed region; `cpu_baseline` is the CPU oracle (a restatement of the reference: kind
"port") timed on one host core over a bounded sample of the same batch, which doubles as a last parity check.
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


class Kinova:
    """BASELINE configs[1]."""
    name = "kinova"
    samples = 128
    batch = 4096
    # algorithmic bytes per sample-eval = rows written once x (7 Jacobian columns + 1 value) x 8 B
    rows = 21 + 17  # JointLimits 7 + VelocityLimits 7 + TorqueLimits 7 rows + one row per collision sphere
    bytes_per_se = rows * 8 * 8
    # algorithmic flops per sample-eval: RNEA 550*nv + analytic derivatives ~7e3 + contraction 2058
    # + sphere FK / box distances ~25e3
    flops_per_se = 550 * 7 + 7000 + 2058 + 25000
    dominant = 0  # index of the dominant kernel among the timed pair
    kernels = ["kinova_fused_kernel<true> (trajectory + RNEA + d tau/dz + limit rows + sphere/box rows)", None]
    pmc_key = "kinova_fused_kernel<true>"
    z_scale = 1.0
    workload = ("Kinova-gen3 7-DoF ArmourBezier, 128 Chebyshev samples x 4096 random z in U(-1,1)^7 per GPU, "
                "eval_g+eval_jac_g (joint/velocity/torque limits + 17 spheres vs 10 boxes in reach), no solver")

    def build(self, api, keep):
        import helpers as H
        import raptor_amd

        model = raptor_amd.load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
        return H.kinova_problem(api, model, keep, N=self.samples, boxes=H.reach_boxes())


class KinovaLongerHorizon:
    """KinovaLongerHorizonOptimizer: PiecewiseBezierCurves, degree 4 -> 84 variables, N = 20 (degree + 1) samples
    as KinovaLongerHorizonPybindWrapper.cpp:151 picks them.  Not a BASELINE config: measured so that the per-column kernels have numbers."""
    name = "kinova_lh"
    samples = 100
    batch = 1024
    degree = 4
    rows = 21 + 17
    bytes_per_se = rows * (84 + 1) * 8
    flops_per_se = 550 * 7 + 3 * 2 * 7 * 7 * 84 + 7000 + 25000  # RNEA + the reference's three 7x7 . 7x84 products + derivatives + spheres
    dominant = 1
    kernels = ["kinova_fused_kernel<true,RECORD> (trajectory + RNEA + d tau/d(q,v,a) blocks + sphere gradients per joint + g rows)",
               "kinova_expand_kernel (pg_pz rows in Optimizer order from the per-sample records and the column weights)"]
    pmc_key = "kinova_expand_kernel"
    z_scale = 1.0
    workload = ("Kinova-gen3 7-DoF PiecewiseBezierCurves degree 4 (84 variables), 100 uniform samples x 1024 random z in U(-1,1)^84 per GPU, "
                "eval_g+eval_jac_g (joint/velocity/torque limits + 17 spheres vs 10 boxes in reach), no solver")

    def build(self, api, keep):
        import helpers as H
        import numpy as np
        import raptor_amd
        from raptor_amd import _capi as A

        model = raptor_amd.load_urdf(os.path.join(ROOT, "tests", "robots", "kinova.urdf"))
        mh = api.model_create(model)
        keep.append(mh)
        t = A.make_traj_desc(keep, A.TRAJ_PIECEWISE_BEZIER, 3.0, self.samples, 7, self.degree, A.TIME_UNIFORM,
                             q0=np.zeros(7), qT=np.full(7, 0.5))
        c, rpy, size = H.reach_boxes()
        return A.ProblemHandle.kinova(api, mh, t, c, rpy, size)


class Legged:
    """BASELINE configs[2] (Talos) / configs[4] (G1, one GPU's shard of the 8192-problem sweep)."""
    samples = 14
    batch = 1024
    bytes_per_se = None                   # 8 (rows / N) (n + 1): every row the expand kernel writes, once, g and pg_pz
    flops_per_se = 230000                 # SURVEY.md 8d total for Talos / G1 as the reference does it
    dominant = 1
    kernels = ["legged_core_kernel<16,true> (trajectory + re-rooted RNEA + analytic derivatives + g rows) + legged_period_kernel (reset-map rows)",
               "legged_expand_kernel (pg_pz rows in Optimizer order from the per-sample records)"]
    pmc_key = "legged_expand_kernel"
    z_scale = 0.1  # TalosSingleStep.cpp:61

    def __init__(self, which):
        self.name = which
        self.workload = (f"{'Talos' if which == 'talos' else 'Unitree-G1'} single-step gait, nv=18 (6 floating + 12), Bezier degree 6, "
                         f"14 Chebyshev samples x {self.batch} random z in 0.1*U(-1,1)^108 per GPU, eval_g+eval_jac_g "
                         "(torque limits + constrained joint limits + rectangle contact + customized swing-foot / torso rows + reset map: "
                         "all 729 rows of the reference's single-step problem), no solver")

    def build(self, api, keep):
        import helpers as H

        return H.legged_problem(api, self.name, keep, N=self.samples, customized=True, periodic=True)[0]


class Digit:
    """BASELINE configs[3] (north_star target: >= 1e7 sample-evals/s on one MI355X)."""
    name = "digit"
    samples = 14
    batch = 4096
    bytes_per_se = None                   # 8 (rows / N) (n + 1) over the expand kernel's rows (torque, joint limits, contact, customized)
    flops_per_se = 1300000                # SURVEY.md 8d total for Digit as the reference does it
    dominant = 1
    kernels = ["digit_loops_kernel<knee,toe> + legged_core_kernel<32,true> (closed loops, re-rooted RNEA, analytic derivatives, g rows) "
               "+ digit_period_loops_kernel + legged_period_kernel (reset-map rows)",
               "legged_expand_kernel (pg_pz rows in Optimizer order from the per-sample records)"]
    pmc_key = "legged_expand_kernel"
    z_scale = 0.1  # DigitSingleStep.cpp:75
    workload = ("Digit-v3 single-step gait, nv=36 (6 floating + 12 actuated + 18 closed-loop), Bezier degree 5, "
                "14 Chebyshev samples x 4096 random z in 0.1*U(-1,1)^132 per GPU, eval_g+eval_jac_g "
                "(torque limits + constrained joint limits + rectangle contact + customized swing-foot / torso rows + reset map: "
                "all 1005 rows of the reference's single-step problem), no solver")

    def build(self, api, keep):
        import helpers as H

        return H.digit_problem(api, keep, N=self.samples, customized=True, periodic=True)[0]


WORKLOADS = {"kinova": Kinova, "kinova_lh": KinovaLongerHorizon, "talos": lambda: Legged("talos"), "g1": lambda: Legged("g1"), "digit": Digit}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="kinova", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Digit record appended to the default (Kinova) line")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-problem latency loop (profiling runs: keeps the kernel statistics to the batch launches)")
    ap.add_argument("--no-eval-h", action="store_true", help="skip the eval_h record of the Kinova line")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

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

    hip = raptor_amd.api()  # raises if the HIP library is missing: there is no fallback
    hip.init(local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(W, batch, cpu_seconds):
        """W over `args.steps` timed steps; returns the JSON record (rank 0) or None."""
        keep = []
        prob = W.build(hip, keep)
        B = batch or W.batch
        gen = torch.Generator(device="cpu").manual_seed(20250523 + 1 + rank)
        Z = (W.z_scale * (torch.rand(B, prob.n, generator=gen, dtype=torch.float64) * 2 - 1)).cuda()
        G = torch.empty(B, prob.m, dtype=torch.float64, device="cuda")
        J = torch.empty(B, prob.m, prob.n, dtype=torch.float64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        flags = A.EVAL_G | A.EVAL_JAC

        def step():
            prob.eval_device(Z.data_ptr(), B, flags, G.data_ptr(), J.data_ptr(), stream)

        for _ in range(args.warmup):
            step()
        barrier()
        prob.profile_begin(args.steps)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        n_prof, k0_ms, k1_ms = prob.profile_end()
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if rank != 0:
            return None
        sample_evals = B * W.samples
        assert bool(torch.isfinite(G).all()) and bool(torch.isfinite(J).all())
        kms = [k0_ms, k1_ms]
        dom_ms = kms[W.dominant]
        dom_s = dom_ms * 1e-3
        tot_s = (k0_ms + k1_ms) * 1e-3
        # legged workloads: the dominant (expand) kernel writes every block but the reset map, which legged_period_kernel
        # writes once per z inside the first kernel group
        exp_rows = prob.m - (prob.block_offsets[-1] - prob.block_offsets[-2] if not getattr(W, "name", "").startswith("kinova") else 0)
        bytes_per_se = W.bytes_per_se or 8.0 * exp_rows / W.samples * (prob.n + 1)
        alg_bytes = sample_evals * bytes_per_se
        ms_per_step = 1e3 * elapsed / args.steps
        out = {
            "metric": "RNEA+dtau/dz sample-evals/sec (fp64)",
            "value": world * sample_evals * args.steps / elapsed,
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
                "workload": W.workload,
                "batch_per_gpu": B, "samples": W.samples, "n": prob.n, "m": prob.m,
                "flags": "G|JAC", "parallelism": f"multi-start x{world} (independent batches, no collective)",
            },
            "eval_jac_g_latency_ms_batch": ms_per_step,
            "roofline": {
                "kernel": W.kernels[W.dominant],
                "bound": "hbm",
                "achieved": alg_bytes / dom_s / 1e9 if dom_s > 0 else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (alg_bytes / dom_s / 1e9) / HBM_PEAK_GBS if dom_s > 0 else None,
                "traffic": pmc_traffic(W.pmc_key, W.name),
                "kernel_ms": dom_ms,
                "launches_timed": n_prof,
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_per_sample_eval": bytes_per_se,
            },
            "roofline_fp64": {
                "kernel": "all kernels of one evaluation",
                "bound": "fp64-valu",
                "achieved": sample_evals * W.flops_per_se / tot_s / 1e12 if tot_s > 0 else None,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (sample_evals * W.flops_per_se / tot_s / 1e12) / FP64_PEAK_TFLOPS if tot_s > 0 else None,
                "algorithmic_flops_per_sample_eval": W.flops_per_se,
            },
            "kernels_ms": {k: v for k, v in zip(W.kernels, kms) if k},
        }
        # single-problem latency (the Ipopt-driven use: one z, eval_g + eval_jac_g, results left in HBM), host-synchronised per call
        if not args.no_latency:
            for _ in range(5):
                prob.eval_device(Z.data_ptr(), 1, flags, G.data_ptr(), J.data_ptr(), stream)
            torch.cuda.synchronize()
            reps = 200
            t1 = time.perf_counter()
            for _ in range(reps):
                prob.eval_device(Z.data_ptr(), 1, flags, G.data_ptr(), J.data_ptr(), stream)
                torch.cuda.synchronize()
            out["single_problem_eval_jac_g_latency_ms"] = 1e3 * (time.perf_counter() - t1) / reps
        if world == 1 and cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(W, Z.cpu().numpy(), cpu_seconds, G, J)
        if W.name.startswith("kinova") and world == 1 and not args.no_eval_h:
            out["eval_h"] = eval_h_record(W, prob, Z, B, cpu_seconds > 0)
        return out

    def eval_h_record(W, prob, Z, B, with_cpu):
        """Optimizer::eval_h for the same batch (KinovaOptimizer runs with exact Hessians): one fused kernel per call,
        28 doubles out per z.  Timed with events on the evaluation stream; checked against the oracle on the first z."""
        import numpy as np

        armour = W.name == "kinova"  # the longer-horizon optimizer's objective is a sum of Costs: its Hessian stays with the caller
        if armour:
            prob.set_objective(np.zeros(prob.n), W.samples // 2)
        gen = torch.Generator(device="cpu").manual_seed(7)
        lam = (torch.rand(B, prob.m, generator=gen, dtype=torch.float64) * 2 - 1).cuda()
        of = torch.ones(B, dtype=torch.float64, device="cuda") if armour else None
        Hh = torch.empty(B, prob.n * (prob.n + 1) // 2, dtype=torch.float64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream

        def call():
            prob.eval_h_device(Z.data_ptr(), B, of.data_ptr() if armour else 0, lam.data_ptr(), Hh.data_ptr(), stream)

        for _ in range(2):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rec = {
            "kernel": "kinova_hessian_kernel (hyper-dual forward recursion, Lagrangian Hessian fused)" if armour else
                      "kinova_hess_blocks_kernel (per-sample second-order blocks: hyper-dual qq, reduced vv / qv / qa passes) + kinova_hess_contract_kernel (column weights)",
            "ms_per_call": ms,
            "lagrangian_hessians_per_s": B / (ms * 1e-3),
            "sample_evals_per_s": B * W.samples / (ms * 1e-3),
            "batch": B, "samples": W.samples,
        }
        if armour:
            # as the reference does it: second-order tensors + contraction + sphere Hessians ~ 1.2e5 flops / sample-eval
            tf = B * W.samples * 1.2e5 / (ms * 1e-3) / 1e12
            rec["roofline_fp64"] = {"bound": "fp64-valu", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": tf / FP64_PEAK_TFLOPS, "algorithmic_flops_per_sample_eval": 1.2e5}
        if not args.no_latency:  # the Ipopt-driven use: one z per call, host-synchronised
            reps = 100
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                prob.eval_h_device(Z.data_ptr(), 1, of.data_ptr() if armour else 0, lam.data_ptr(), Hh.data_ptr(), stream)
                torch.cuda.synchronize()
            rec["single_problem_eval_h_latency_ms"] = 1e3 * (time.perf_counter() - t1) / reps
            call()  # restore the batch results for the check below
            torch.cuda.synchronize()
        if with_cpu:
            import helpers as H

            oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
            keep2 = []
            po = W.build(oracle, keep2)
            if armour:
                po.set_objective(np.zeros(prob.n), W.samples // 2)
            nz = 8 if armour else 2
            t0 = time.perf_counter()
            Ho = po.eval_h_host(Z[:nz].cpu().numpy(), lam[:nz].cpu().numpy(), 1.0 if armour else None)
            dt = time.perf_counter() - t0
            err = H.rel_err(Hh[:nz].cpu().numpy(), Ho)
            if not err < 1e-9:
                raise SystemExit(f"GPU eval_h differs from the oracle: {err:.2e}")
            rec["cpu_oracle_ms_per_z"] = 1e3 * dt / nz
            rec["rel_err_vs_oracle"] = err
        return rec

    out = run(WORKLOADS[args.workload](), args.batch, 0.0 if args.no_cpu_baseline else args.cpu_seconds)
    if args.workload == "kinova" and not args.no_secondary:
        # the north-star target is quoted on Digit-v3 (BASELINE.json): measure it in the same run, as a secondary record
        sec = run(WORKLOADS["digit"](), 0, 0.0 if args.no_cpu_baseline else min(args.cpu_seconds, 6.0))
        if out is not None:
            out["secondary"] = {"digit": {k: sec[k] for k in ("value", "unit", "ms_per_step", "config", "roofline", "kernels_ms", "single_problem_eval_jac_g_latency_ms", "cpu_baseline") if k in sec}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(key, workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/pmc_latest.json,
    collected with this same command in separate WRITE_SIZE / FETCH_SIZE runs); None when no such entry is present."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        k = d.get(workload + ":" + key) or d[key]
        return k["WRITE_SIZE_bytes"] + k["FETCH_SIZE_bytes"]
    except Exception:
        return None


def cpu_baseline(W, Zs, budget_s, G, J):
    """The CPU oracle (restatement of the reference's serial per-sample loop), one host core, on the first few z
    of the same batch; also used as a last check of the GPU results on exactly those z."""
    from raptor_amd import _capi as A
    import helpers as H

    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    keep = []
    po = W.build(oracle, keep)
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
        "value": nz * W.samples / dt,
        "unit": "sample-evals/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {nz} of the {len(Zs)} z of the same batch (x{W.samples} samples), eval_g+eval_jac_g, "
                  f"oracle/liboracle.so (-O2, serial like the reference's per-sample loop); GPU rows for those z checked "
                  f"against it: rel err g {eg:.1e}, pg_pz {ej:.1e}",
        "seconds": dt,
    }


if __name__ == "__main__":
    main()
