#!/usr/bin/env python3
"""Headline benchmark: RNEA + d(tau)/dz sample-evals/s (fp64).

  python bench.py --gpus N --steps K --warmup W [--workload kinova|kinova_lh|talos|g1|digit]

N > 1 without a launcher (WORLD_SIZE unset): this process starts N ranks itself (`python -m torch.distributed.run`, as a child
process, before anything here touches the GPU) and exits with their code; it refuses to run when fewer than N GPUs are visible.
Under an external launcher (WORLD_SIZE set) it is one rank: one process per GPU, RCCL through torch.distributed ("nccl").

Default workload = BASELINE.json configs[1] (Kinova): a *step* is one pass of the hot path (eval_g + eval_jac_g, i.e.
raptor_eval_batch with G|JAC) over one batch of B = 4096 decision vectors x N = 128 time samples = 524 288 sample-evals,
inputs and outputs resident in HBM.  With N GPUs every rank evaluates its own batch (multi-start problems are independent:
weak scaling, no data-path collective in the evaluation).  The path's one exchange step -- the gather of per-problem records
after a multi-start sweep (BASELINE.json configs[4]: Unitree-G1, 8192 problems over the GPUs of the node) -- is measured in the
same run as the record `multistart` (evaluation + record + ONE all_gather_into_tensor inside the timed region).

Prints ONE JSON line (rank 0).  `roofline` is for the longest kernel, timed live with HIP events recorded on the evaluation
stream inside the timed region; `cpu_baseline` is the CPU oracle (a restatement of the reference: kind "port") timed on the
host cores over a bounded sample of the same batch (1 core like the reference's serial loop, and all cores as independent
processes over z), which doubles as a last parity check.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak
SPINUP_SECONDS = 0.25     # untimed passes before the W warm-up steps: the GPU needs ~50 ms of load to reach its clocks


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
    kernels = ["kinova_fused_kernel<true> (trajectory + RNEA + d tau/dz + limit rows + sphere/box rows)", None]
    pmc_keys = ["kinova_fused_kernel<true, false>", None]
    bound = "fp64-valu issue (VALU-busy from the PMC pass beside it); the HBM fraction is reported because the kernel's only traffic is its output"
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
    kernels = ["kinova_fused_kernel<true,RECORD> (trajectory + RNEA + d tau/d(q,v,a) blocks + sphere gradients per joint + g rows)",
               "kinova_expand_kernel (pg_pz rows in Optimizer order from the per-sample records and the column weights)"]
    pmc_keys = ["kinova_fused_kernel<true, true>", "kinova_expand_kernel"]
    bound = "hbm"
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
    # fused form (round 3): the core kernel writes every pg_pz row of the torque / joint-limit / contact / customized blocks itself
    kernels = [None,
               "legged_core_kernel<16,true,fused> (trajectory + re-rooted RNEA + analytic derivatives + g rows + records, then the pg_pz rows of "
               "the torque / joint-limit / contact / customized blocks in Optimizer order)",
               "legged_period_kernel (reset-map rows)"]
    rows_kernel = 1
    pmc_keys = [None, "legged_core_kernel<16, true, true, false>", "legged_period_kernel<18, 6, true>"]
    kernels_unfused = ["legged_core_kernel<16,true> (trajectory + re-rooted RNEA + analytic derivatives + g rows) + legged_period_kernel (reset-map rows)",
                       "legged_expand_kernel (pg_pz rows in Optimizer order from the per-sample records)"]
    pmc_keys_unfused = ["legged_core_kernel<16, true>", "legged_expand_kernel"]
    bound = "hbm"
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
    # fused form (round 3): the core kernel writes every pg_pz row of the torque / joint-limit / contact / customized blocks itself
    kernels = ["digit_loops_both_kernel (closed loops: knee + toe, Newton + implicit 1st-3rd derivatives)",
               "legged_core_kernel<32,true,fused> (re-rooted RNEA, analytic derivatives, g rows, records, then the pg_pz rows of the "
               "torque / joint-limit / contact / customized blocks in Optimizer order)",
               "digit_period_loops_kernel + legged_period_kernel (reset-map rows)"]
    rows_kernel = 1
    pmc_keys = ["digit_loops_both_kernel", "legged_core_kernel<32, true, true, true>", "legged_period_kernel<36, 24, true>"]
    kernels_unfused = ["digit_loops_both_kernel (knee + toe) + legged_core_kernel<32,true> (closed loops, re-rooted RNEA, analytic derivatives, g rows) "
                       "+ digit_period_loops_kernel + legged_period_kernel (reset-map rows)",
                       "legged_expand_kernel (pg_pz rows in Optimizer order from the per-sample records)"]
    pmc_keys_unfused = ["legged_core_kernel<32, true>", "legged_expand_kernel"]
    bound = "hbm"
    z_scale = 0.1  # DigitSingleStep.cpp:75
    workload = ("Digit-v3 single-step gait, nv=36 (6 floating + 12 actuated + 18 closed-loop), Bezier degree 5, "
                "14 Chebyshev samples x 4096 random z in 0.1*U(-1,1)^132 per GPU, eval_g+eval_jac_g "
                "(torque limits + constrained joint limits + rectangle contact + customized swing-foot / torso rows + reset map: "
                "all 1005 rows of the reference's single-step problem), no solver")

    def build(self, api, keep):
        import helpers as H

        return H.digit_problem(api, keep, N=self.samples, customized=True, periodic=True)[0]


WORKLOADS = {"kinova": Kinova, "kinova_lh": KinovaLongerHorizon, "talos": lambda: Legged("talos"), "g1": lambda: Legged("g1"), "digit": Digit}
MULTISTART_TOTAL = 8192   # BASELINE.json configs[4]
MULTISTART_WEAK = 1024    # ... = 1024 problems per GPU on the 8-GPU node


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="kinova", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Digit and Talos records appended to the default (Kinova) line")
    ap.add_argument("--no-multistart", action="store_true", help="skip the G1 multi-start sweep record (configs[4])")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-problem latency loops (profiling runs: keeps the kernel statistics to the batch launches)")
    ap.add_argument("--no-eval-h", action="store_true", help="skip the eval_h record of the Kinova line")
    ap.add_argument("--no-spinup", action="store_true", help="skip the untimed clock spin-up before the warm-up steps")
    ap.add_argument("--no-check", action="store_true", help=argparse.SUPPRESS)  # phase-stub builds (scripts/*_phases.sh) leave garbage rows
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-worker", nargs=3, metavar=("WORKLOAD", "ZFILE", "OUT"), help=argparse.SUPPRESS)
    return ap.parse_args()


def main() -> None:
    args = parse_args()
    if args.cpu_worker:
        return cpu_worker(*args.cpu_worker)
    if args.gpus < 1:
        raise SystemExit("--gpus must be positive")
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        # no launcher: start the ranks ourselves, as a child process, before this process has touched the GPU
        from raptor_amd.launch import run_ranks, visible_gpu_count

        visible = visible_gpu_count()  # KFD topology in sysfs: the HIP runtime stays unloaded in this (parent) process
        if visible is not None and visible < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} asked for, {visible} GPU(s) visible: refusing to report a smaller run")
        # (topology unreadable: every rank checks for its own GPU below and the launcher's exit code reports a missing one)

        sys.exit(run_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world = int(env_world) if env_world is not None else 1
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
    run_rank(args)


def run_rank(args) -> None:
    import torch

    from raptor_amd.launch import dist_env
    from raptor_amd.multistart import shard_range, timed_sweep

    rank, local_rank, world = dist_env()
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to file descriptor 1 when its first communicator
    # is created, so everything written to fd 1 from here on goes to stderr and the line itself is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no GPU {local_rank}")
    dist = None
    torch.cuda.set_device(local_rank)
    rccl_note = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import raptor_amd
    from raptor_amd import _capi as A

    hip = raptor_amd.api()  # raises if the HIP library is missing: there is no fallback
    hip.init(local_rank)

    def barrier():
        if dist is not None and world > 1:  # (a one-rank group exists for the multi-start record's collective only)
            dist.barrier()
        torch.cuda.synchronize()

    def spinup(step):
        """Untimed passes until the GPU has been busy for SPINUP_SECONDS (clock ramp); returns how many ran."""
        if args.no_spinup:
            return 0
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < SPINUP_SECONDS:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            n += 10
        return n

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

        spun = spinup(step)
        for _ in range(args.warmup):
            step()
        barrier()
        prob.profile_begin(args.steps)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        n_prof, kms3 = prob.profile_end_split()
        per_rank_ms, ranks_seen = [1e3 * elapsed / args.steps], 1
        if dist is not None and world > 1:
            mine = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            allt = torch.empty(world, dtype=torch.float64, device="cuda")
            dist.all_gather_into_tensor(allt, mine)            # RCCL: every rank's own time
            one = torch.ones(1, dtype=torch.float64, device="cuda")
            dist.all_reduce(one, op=dist.ReduceOp.SUM)          # RCCL: how many ranks took part
            per_rank_ms = [1e3 * float(x) / args.steps for x in allt.cpu()]
            ranks_seen = int(round(float(one.item())))
            elapsed = float(allt.max().item())
        if rank != 0:
            return None
        sample_evals = B * W.samples
        assert args.no_check or (bool(torch.isfinite(G).all()) and bool(torch.isfinite(J).all()))
        # three timed intervals per evaluation (raptor_profile_end_split): kernels before the split | 0 | kernels after it, except on the
        # fused legged path (loops | core kernel writing the rows | reset map)
        if len(W.kernels) == 3 and kms3[1] < 0.02 * sum(kms3):  # RAPTOR_NO_FUSE=1 (A/B runs): the two-kernel form ran
            W.kernels, W.pmc_keys, W.rows_kernel = W.kernels_unfused, W.pmc_keys_unfused, 1
        kms = list(kms3) if len(W.kernels) == 3 else [kms3[0] + kms3[1], kms3[2]]
        dom = max(range(len(kms)), key=lambda k: kms[k])   # the longest kernel (group) of the evaluation
        tot_s = sum(kms) * 1e-3
        # legged workloads: the expand kernel writes every block but the reset map, which legged_period_kernel
        # writes once per z inside the first kernel group
        exp_rows = prob.m - (prob.block_offsets[-1] - prob.block_offsets[-2] if not W.name.startswith("kinova") else 0)
        bytes_per_se = W.bytes_per_se or 8.0 * exp_rows / W.samples * (prob.n + 1)
        alg_bytes = sample_evals * bytes_per_se
        ms_per_step = 1e3 * elapsed / args.steps

        def hbm_line(k):
            """HBM roofline entry of timed kernel (group) k: the algorithmic output bytes belong to the kernel that writes them."""
            ms = kms[k]
            writes_rows = k == rows_k
            if not writes_rows or ms <= 0:
                return {"kernel": W.kernels[k], "kernel_ms": ms, "achieved": None, "frac": None,
                        "note": "this kernel group leaves compact per-sample records, not the rows: its roof is fp64 issue, see roofline_fp64 / valu"}
            gbs = alg_bytes / (ms * 1e-3) / 1e9
            return {"kernel": W.kernels[k], "kernel_ms": ms, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS}

        rows_k = getattr(W, "rows_kernel", 0 if W.kernels[1] is None else 1)
        lines_by_k = {k: hbm_line(k) for k in range(len(kms)) if W.kernels[k]}
        lines = list(lines_by_k.values())
        traffic, pmc_note = pmc_traffic(W.pmc_keys[rows_k], W.name)
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
                "flags": "G|JAC", "parallelism": f"multi-start x{world} (independent batches, no collective in the evaluation; the gather step is the `multistart` record)",
                "untimed_spinup_steps": spun,
            },
            "ranks_seen_by_rccl": ranks_seen,
            "per_rank_ms_per_step": per_rank_ms,
            "eval_jac_g_latency_ms_batch": ms_per_step,
            "roofline": {
                "kernel": W.kernels[rows_k],
                "bound": "hbm",
                "binding_roof": W.bound,
                "achieved": lines_by_k[rows_k]["achieved"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": lines_by_k[rows_k]["frac"],
                "traffic": traffic,
                "traffic_source": pmc_note,
                "kernel_ms": kms[rows_k],
                "longest_kernel": W.kernels[dom],
                "launches_timed": n_prof,
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_per_sample_eval": bytes_per_se,
                "per_kernel": lines,
                "mfma": "none: no matrix-core instruction anywhere in the library (the d tau/dz contraction has rank-1 column structure, DESIGN.md 3.1)",
            },
            "roofline_fp64": {
                "kernel": "all kernels of one evaluation",
                "bound": "fp64-valu",
                "kind": "reference-equivalent: flops the reference's algorithm spends on the same work (SURVEY.md 8d), not flops executed; "
                        "the kernels execute far fewer (analytic derivatives, no dense products)",
                "achieved": sample_evals * W.flops_per_se / tot_s / 1e12 if tot_s > 0 else None,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (sample_evals * W.flops_per_se / tot_s / 1e12) / FP64_PEAK_TFLOPS if tot_s > 0 else None,
                "algorithmic_flops_per_sample_eval": W.flops_per_se,
                "valu": pmc_valu(W, kms),
            },
            "kernels_ms": {k: v for k, v in zip(W.kernels, kms) if k},
        }
        if W.bound.startswith("fp64"):
            # The headline object carries the BINDING roof ("report both fractions and name the larger as binding"):
            # for the fused Kinova kernel that is fp64 VALU issue.  achieved = executed VALU wave-instructions of one launch (the
            # committed SQ_INSTS_VALU pass of this same command) x 128 flops -- every VALU wave-instruction priced as one 64-lane fp64
            # FMA, the most a SIMD can retire per issue slot -- over the launch time measured live; peak = the fp64 vector peak.
            # frac is therefore VALU-busy at the nominal 4 cycles per instruction and 2.4 GHz.  The HBM line moves to `hbm`.
            v = (out["roofline_fp64"]["valu"] or {}).get("per_kernel") or []
            ent = next((e for e in v if e["kernel"] == W.pmc_keys[rows_k]), None)
            hbm = {k: out["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                   "algorithmic_bytes_per_launch", "algorithmic_bytes_per_sample_eval")}
            if ent is not None:
                ach = ent["executed_fp64_tflops_upper_bound"]
                out["roofline"].update({
                    "bound": "fp64-valu", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                    "achieved_definition": "VALU wave-instructions per launch (profiles/pmc_latest.json, SQ_INSTS_VALU) x 64 lanes x 2 flops / "
                                           "kernel time measured in this run = VALU-busy at the nominal 4 cycles and 2.4 GHz",
                    "frac_at_measured_issue_rate": ent["valu_busy_at_measured_issue_rate"],
                    "measured_issue_rate_source": "profiles/r3_valu_rate_probe.txt: an fp64 wave-instruction costs its SIMD 5.7 / 5.0 / 4.7 "
                                                  "shader cycles at 2 / 3 / 4 wavefronts per SIMD with the shader clock itself measured at "
                                                  "2.2-2.3 GHz under sustained fp64 FMA issue (clock64 against s_memrealtime in the same launch)",
                    "hbm": hbm,
                })
            else:
                out["roofline"]["note"] = "no PMC record of this kernel beside the run: the fp64-valu fraction cannot be formed, the HBM line stands in"
        if not args.no_latency:
            out.update(latency_records(prob, Z, G, J, flags, stream))
        if world == 1 and cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(W, Z.cpu().numpy(), cpu_seconds, G, J)
        if W.name.startswith("kinova") and world == 1 and not args.no_eval_h:
            out["eval_h"] = eval_h_record(W, prob, Z, B, cpu_seconds > 0)
        return out

    def latency_records(prob, Z, G, J, flags, stream):
        """The Ipopt-driven use: one z per call, host-synchronised.  (a) results left in HBM (raptor_eval_batch);
        (b) the adapter's real call, raptor_eval_batch_host: host buffers in and out, PCIe included."""
        rec = {}
        for _ in range(5):
            prob.eval_device(Z.data_ptr(), 1, flags, G.data_ptr(), J.data_ptr(), stream)
        torch.cuda.synchronize()
        reps = 200
        t1 = time.perf_counter()
        for _ in range(reps):
            prob.eval_device(Z.data_ptr(), 1, flags, G.data_ptr(), J.data_ptr(), stream)
            torch.cuda.synchronize()
        rec["single_problem_eval_jac_g_latency_ms"] = 1e3 * (time.perf_counter() - t1) / reps
        import ctypes
        import numpy as np

        z1 = np.ascontiguousarray(Z[:1].cpu().numpy())
        g1, j1 = np.empty(prob.m), np.empty((prob.m, prob.n))  # caller-owned buffers, reused like the adapter's members
        dp = ctypes.POINTER(ctypes.c_double)
        fn, args_ = prob.api.fn("eval_batch_host"), (prob.handle, z1.ctypes.data_as(dp), 1, flags, g1.ctypes.data_as(dp), j1.ctypes.data_as(dp))
        for _ in range(5):
            prob.api.check(fn(*args_))
        reps = 200
        t1 = time.perf_counter()
        for _ in range(reps):
            fn(*args_)
        rec["single_problem_eval_jac_g_latency_ms_host"] = 1e3 * (time.perf_counter() - t1) / reps
        assert np.array_equal(g1, G[0].cpu().numpy()) or float(np.max(np.abs(g1 - G[0].cpu().numpy()))) < 1e-9
        rec["single_problem_host_bytes"] = 8 * (prob.n + prob.m * (prob.n + 1))
        return rec

    def eval_h_record(W, prob, Z, B, with_cpu):
        """Optimizer::eval_h for the same batch (KinovaOptimizer runs with exact Hessians): two kernels per call (obstacle rows,
        torque rows), 28 doubles out per z, the per-row Hessians never formed.  Timed with events on the evaluation stream; checked against the oracle on the first z."""
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

        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rec = {
            "kernel": "kinova_hessian_kernel (obstacle rows + hess_f) + kinova_hess_torque_kernel (value and first-order chains walked once per sample, mixed parts as cross products)" if armour else
                      "kinova_hess_blocks_kernel (per-sample second-order blocks: hyper-dual qq, reduced vv / qv / qa passes) + kinova_hess_contract_kernel (column weights)",
            "ms_per_call": ms,
            "lagrangian_hessians_per_s": B / (ms * 1e-3),
            "sample_evals_per_s": B * W.samples / (ms * 1e-3),
            "batch": B, "samples": W.samples,
        }
        if armour:
            # as the reference does it: second-order tensors + contraction + sphere Hessians ~ 1.2e5 flops / sample-eval
            tf = B * W.samples * 1.2e5 / (ms * 1e-3) / 1e12
            rec["roofline_fp64"] = {"bound": "fp64-valu", "kind": "reference-equivalent", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
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

    def multistart_record():
        """BASELINE.json configs[4]: the Unitree-G1 multi-start sweep.  MULTISTART_TOTAL problems (the same set on every rank,
        seeded) block-partitioned over the ranks; one sweep = this rank's block through eval_g + eval_jac_g, the per-problem
        record {worst violation, feasible} (Optimizer.cpp:374-378) and ONE all-gather of the records (RCCL), all inside the
        timed region.  Strong scaling: the total is fixed.  The weak figure (MULTISTART_WEAK problems per GPU) is measured too."""
        W = Legged("g1")
        keep = []
        prob = W.build(hip, keep)
        lb, ub = (torch.from_numpy(x).cuda() for x in prob.bounds())
        stream = torch.cuda.current_stream().cuda_stream
        flags = A.EVAL_G | A.EVAL_JAC
        rec = {"kernels": W.kernels}
        for key, total in (("strong", MULTISTART_TOTAL), ("weak", MULTISTART_WEAK * world)):
            gen = torch.Generator(device="cpu").manual_seed(20250523 + 5)
            Z_all = (W.z_scale * (torch.rand(total, prob.n, generator=gen, dtype=torch.float64) * 2 - 1)).cuda()
            lo, hi = shard_range(total, world, rank)
            per = -(-total // world)
            G = torch.empty(per, prob.m, dtype=torch.float64, device="cuda")
            J = torch.empty(per, prob.m, prob.n, dtype=torch.float64, device="cuda")

            def evaluate(Zb):
                prob.eval_device(Zb.data_ptr(), Zb.shape[0], flags, G.data_ptr(), J.data_ptr(), stream)
                return G[:Zb.shape[0]]

            viol = torch.empty(per, dtype=torch.float64, device="cuda")

            def violation_native(Gb):  # the record through the library's own kernel (raptor_violation_batch)
                prob.violation_device(Gb.data_ptr(), Gb.shape[0], viol.data_ptr(), stream)
                return viol[:Gb.shape[0]]

            spinup(lambda: evaluate(Z_all[lo:hi]))
            def violation_into(Gb, out):  # ... written straight into the record that is gathered
                prob.violation_device(Gb.data_ptr(), Gb.shape[0], out.data_ptr(), stream)

            r = timed_sweep(evaluate, Z_all, lb, ub, args.steps, args.warmup, sync=torch.cuda.synchronize, violation_fn=violation_native,
                            violation_into=violation_into)
            check_v = violation_native(evaluate(Z_all[lo:hi])).clone()
            from raptor_amd.multistart import violation as violation_torch
            assert torch.equal(check_v, violation_torch(G[:hi - lo], lb, ub)), "raptor_violation_batch differs from the torch formula"
            # the evaluation alone (no record, no collective), for the split of the sweep time
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                evaluate(Z_all[lo:hi])
            barrier()
            ev = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            if dist is not None:
                dist.all_reduce(ev, op=dist.ReduceOp.MAX)
            sweep_ms = 1e3 * r["elapsed_s"] / args.steps
            eval_ms = 1e3 * float(ev.item()) / args.steps
            split = r.get("split_ms") or {}
            rec[key] = {
                "problems_total": total, "problems_per_gpu": per, "samples": W.samples, "n": prob.n, "m": prob.m,
                "value": total * W.samples * args.steps / r["elapsed_s"], "unit": "sample-evals/s",
                "ms_per_sweep": sweep_ms,
                # HIP events inside the timed sweeps (this rank's stream): evaluation | record | collective
                "ms_evaluation": split.get("evaluation_ms"), "ms_record": split.get("record_ms"), "ms_gather": split.get("gather_ms"),
                "ms_record_and_gather": (split.get("record_ms", 0.0) + split.get("gather_ms", 0.0)) if split else None,
                "ms_evaluation_only_separate_run": eval_ms,
                "per_rank_ms_per_sweep": [1e3 * x / args.steps for x in r["per_rank_s"]],
                "ranks_seen_by_rccl": r["ranks_seen"],
                "gathered_records": int(r["violation"].numel()), "feasible": int(r["feasible"].sum().item()),
                "collective": ("all_gather_into_tensor of one double per problem (RCCL" + (")" if world > 1 else ", ONE rank: the leg runs, it moves nothing)")) if dist is not None else "none (no process group)",
                "scaling": key,
            }
            del Z_all, G, J
        rec["workload"] = ("Unitree-G1 single-step gait, 729 rows x 108 variables, 14 Chebyshev samples; "
                           f"{MULTISTART_TOTAL} problems over {world} GPU(s) (strong) and {MULTISTART_WEAK} per GPU (weak), eval_g+eval_jac_g, "
                           "Jacobians stay on the GPU that produced them")
        return rec if rank == 0 else None

    cpu_s = 0.0 if args.no_cpu_baseline else args.cpu_seconds
    out = run(WORKLOADS[args.workload](), args.batch, cpu_s)
    if args.workload == "kinova" and not args.no_secondary and world == 1:
        # the north-star target is quoted on Digit-v3 (BASELINE.json): measure it in the same run, as a secondary record
        sec = run(WORKLOADS["digit"](), 0, min(cpu_s, 6.0))
        out["secondary"] = {"digit": {k: sec[k] for k in ("value", "unit", "ms_per_step", "config", "roofline", "roofline_fp64", "kernels_ms",
                                                          "single_problem_eval_jac_g_latency_ms", "single_problem_eval_jac_g_latency_ms_host",
                                                          "cpu_baseline") if k in sec}}
        # BASELINE.json configs[2] (Talos): the same kernels as G1 on the other 18-DoF robot, recorded so that every config has a line
        sec = run(WORKLOADS["talos"](), 0, 0.0)
        out["secondary"]["talos"] = {k: sec[k] for k in ("value", "unit", "ms_per_step", "config", "roofline", "kernels_ms") if k in sec}
    if args.workload == "kinova" and not args.no_multistart:
        if world == 1:
            # ONE rank: the RCCL leg of the multi-start record still runs (a one-rank group), so that the exchange step is executed code
            # on every bench run, not only on the driver's multi-GPU node.  A group that cannot be created here is reported, not fatal.
            # Created HERE, after every other record: RCCL's helper threads, once alive, add ~30 us to the host-synchronised
            # single-problem latency loops (Digit 167 -> 196 us when the group was created at the start of the run).
            try:
                import torch.distributed as dist  # noqa: F811
                from raptor_amd.launch import free_port

                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1,
                                        device_id=torch.device("cuda", local_rank))
            except Exception as e:  # noqa: BLE001
                rccl_note = f"one-rank RCCL group not created: {e}"
                dist = None
        ms = multistart_record()
        if out is not None:
            if ms is not None and rccl_note:
                ms["rccl_note"] = rccl_note
            out["multistart"] = ms
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _pmc_file():
    return os.path.join(ROOT, "profiles", "pmc_latest.json")


def _pmc_stale():
    """True when a source of the profiled kernels differs from what the committed counter file was collected on (by content:
    the file lists the SHA-256 of those sources; modification times do not survive a checkout)."""
    import hashlib
    try:
        want = json.load(open(_pmc_file())).get("sources_sha256")
        if not want:
            return True
        src = os.path.join(ROOT, "raptor_amd", "csrc")
        return any(hashlib.sha256(open(os.path.join(src, f), "rb").read()).hexdigest() != h for f, h in want.items())
    except (OSError, ValueError):
        return True


def pmc_traffic(key, workload):
    """HBM bytes per launch of the row-writing kernel from the committed rocprofv3 --pmc passes (profiles/pmc_latest.json,
    collected with this same command in separate WRITE_SIZE / FETCH_SIZE runs); None when no such entry is present."""
    try:
        d = json.load(open(_pmc_file()))
        k = d.get(workload + ":" + key) or d[key]
        note = "profiles/pmc_latest.json (separate rocprofv3 --pmc passes of this command)"
        if _pmc_stale():
            note += "; WARNING: a kernel source is newer than that file, the counters may describe older code"
        return k["WRITE_SIZE_bytes"] + k["FETCH_SIZE_bytes"], note
    except Exception:
        return None, "no counter entry for this kernel in profiles/pmc_latest.json"


def pmc_valu(W, kms):
    """Executed-VALU utilisation of the two kernel groups from the committed SQ_INSTS_VALU pass: wave-instructions x 4 cycles
    (a wave64 fp64 / 32-bit VALU instruction occupies its SIMD for 4 cycles) over the SIMD-cycles of the measured launch."""
    try:
        d = json.load(open(_pmc_file()))
    except Exception:
        return None
    out = []
    for key, ms in zip(W.pmc_keys, kms):
        ent = (d.get(W.name + ":" + key) or d.get(key)) if key else None
        if not ent or "SQ_INSTS_VALU" not in ent or ms <= 0:
            continue
        simd_cycles = 1024 * 2.4e9 * ms * 1e-3  # 256 CUs x 4 SIMDs at 2.4 GHz
        # measured issue interval of independent fp64 VALU instructions on this part (scripts/valu_rate_probe.hip,
        # profiles/r2f_valu_rate_probe.txt), in 2.4 GHz cycles per wave-instruction and SIMD: 8.3 / 5.9 / 5.3 / 5.0 at 1 / 2 / 3 / >= 4
        # wavefronts per SIMD
        out.append({"kernel": key, "valu_wave_instructions": ent["SQ_INSTS_VALU"], "valu_busy": ent["SQ_INSTS_VALU"] * 4 / simd_cycles,
                    # r3 probe: 2.40 / 2.23 / 2.15 ns per fp64 wave-instruction and SIMD at 2 / 3 / 4 wavefronts per SIMD (wall time)
                    "valu_busy_at_measured_issue_rate": {"2_waves_per_simd": ent["SQ_INSTS_VALU"] * 2.40e-9 / (1024 * ms * 1e-3),
                                                         "3_waves_per_simd": ent["SQ_INSTS_VALU"] * 2.23e-9 / (1024 * ms * 1e-3),
                                                         "4_or_more": ent["SQ_INSTS_VALU"] * 2.15e-9 / (1024 * ms * 1e-3)},
                    "executed_fp64_tflops_upper_bound": ent["SQ_INSTS_VALU"] * 64 * 2 / (ms * 1e-3) / 1e12})
    return {"per_kernel": out, "stale": _pmc_stale(),
            "note": "valu_busy prices a wave-instruction at the nominal 4 cycles at 2.4 GHz; measured (profiles/r3_valu_rate_probe.txt, "
                    "clock measured in the launch): an fp64 VALU wave-instruction occupies its SIMD for 2.40 / 2.23 / 2.15 ns at 2 / 3 / 4 "
                    "wavefronts per SIMD (5.7 / 5.0 / 4.7 cycles of a shader clock that itself sits at 2.2-2.3 GHz under fp64 FMA load), "
                    "a 32-bit one for 1.6 ns (profiles/r3_valu32_rate_probe.txt); the figures price every instruction as fp64, so values "
                    "above 1 mean part of the mix is cheaper; kinova_fused_kernel runs at 3 wavefronts per SIMD, the legged record "
                    "kernels at 2; v_mfma_f64_16x16x4_f64 sustains 26.9 ns per 2048 flops and SIMD = 78 TFLOP/s at an undisturbed 2.39 GHz"} if out else None


def _cpu_cores():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)  # the GPU box exports its CPU share here (16 for one GPU)
    return max(1, min(avail, cap) if cap > 0 else min(avail, 16))


def cpu_worker(workload, zfile, outfile):
    """One process of the all-cores CPU leg: the oracle over its own slice of z, timed."""
    import numpy as np

    from raptor_amd import _capi as A

    W = WORKLOADS[workload]()
    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    keep = []
    po = W.build(oracle, keep)
    Z = np.load(zfile)
    po.eval_host(Z[:1])
    t0 = time.perf_counter()
    po.eval_host(Z)
    json.dump({"nz": int(Z.shape[0]), "seconds": time.perf_counter() - t0}, open(outfile, "w"))


def cpu_baseline(W, Zs, budget_s, G, J):
    """The CPU oracle (restatement of the reference's serial per-sample loop) on the first few z of the same batch.
    Leg 1: one host core -- the reference's own execution model; also the last check of the GPU results on exactly those z.
    Leg 2: all host cores, one oracle process per core over disjoint z (the reference's objects are not thread-safe; a
    multi-start user would run processes) -- the fair "all of the CPU" figure."""
    import numpy as np

    from raptor_amd import _capi as A
    import helpers as H

    oracle = A.CApi(os.path.join(ROOT, "oracle", "liboracle.so"), "oracle_")
    keep = []
    po = W.build(oracle, keep)
    t0 = time.perf_counter()
    po.eval_host(Zs[:1])
    per_z = time.perf_counter() - t0
    nz = int(max(1, min(len(Zs), 0.6 * budget_s / max(per_z, 1e-6))))
    t0 = time.perf_counter()
    Go, Jo = po.eval_host(Zs[:nz])
    dt = time.perf_counter() - t0
    eg = H.rel_err(G[:nz].cpu().numpy(), Go)
    ej = H.rel_err(J[:nz].cpu().numpy(), Jo)
    if not (eg < 1e-9 and ej < 1e-9):
        raise SystemExit(f"GPU result differs from the oracle: g {eg:.2e} pg_pz {ej:.2e}")
    rec = {
        "value": nz * W.samples / dt,
        "unit": "sample-evals/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {nz} of the {len(Zs)} z of the same batch (x{W.samples} samples), eval_g+eval_jac_g, "
                  f"oracle/liboracle.so (-O3, serial like the reference's per-sample loop); GPU rows for those z checked "
                  f"against it: rel err g {eg:.1e}, pg_pz {ej:.1e}",
        "seconds": dt,
        "us_per_eval_jac_g": 1e6 * dt / nz,
    }
    # leg 2: one process per core, each its own problem object and slice
    cores = _cpu_cores()
    if cores > 1:
        per_core = int(max(1, min(len(Zs) // cores, 0.4 * budget_s / max(dt / nz, 1e-6))))
        with tempfile.TemporaryDirectory() as tmp:
            procs = []
            for c in range(cores):
                zf, of = os.path.join(tmp, f"z{c}.npy"), os.path.join(tmp, f"o{c}.json")
                np.save(zf, Zs[c * per_core:(c + 1) * per_core])
                procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", W.name, zf, of],
                                               env=dict(os.environ, OMP_NUM_THREADS="1")), of))
            t0 = time.perf_counter()
            ok = all(p.wait() == 0 for p, _ in procs)
            wall = time.perf_counter() - t0
            if ok:
                res = [json.load(open(of)) for _, of in procs]
                slow = max(r["seconds"] for r in res)
                rec["all_cores"] = {
                    "value": sum(r["nz"] for r in res) * W.samples / slow,
                    "unit": "sample-evals/s", "cores": cores, "host_cpus_visible": os.cpu_count(), "kind": "port",
                    "sample": f"{per_core} z per process x {cores} processes (disjoint slices of the same batch), slowest process {slow:.2f} s, "
                              f"wall clock including process start {wall:.2f} s",
                }
    return rec


if __name__ == "__main__":
    main()
