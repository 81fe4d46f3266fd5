"""Multi-start sharding: independent decision vectors split across the GPUs of one node.

The reference has no distributed code; its only batch dimension is the set of random
initial guesses it solves one after another (Examples/Digit/README.md:196-199).  Here the batch of decision
vectors is block-partitioned over ranks (one process per GPU, contiguous blocks of ceil(B/world) problems,
SURVEY.md 8e); evaluation needs no communication at all, and afterwards ONE collective gathers a fixed-size
record per problem -- feasibility, worst constraint violation and, on request, g itself.  Jacobians stay
resident on the GPU that produced them (a 3.5 GB ring all-gather would be per-link bound on xGMI).

The feasibility rule is the Optimizer's (Optimization/src/Optimizer.cpp:374-378):
    infeasible  <=>  min(g - g_lb) < -tol  or  min(g_ub - g) < -tol.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of problems owned by `rank` (blocks of ceil(total/world), last ones may be short/empty)."""
    if total < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    per = -(-total // world)
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def violation(G: torch.Tensor, g_lb: torch.Tensor, g_ub: torch.Tensor) -> torch.Tensor:
    """Worst bound violation per problem, max(0, g_lb - g, g - g_ub) over the rows of each problem."""
    if G.shape[0] == 0:
        return G.new_zeros((0,))
    v = torch.maximum(g_lb - G, G - g_ub).amax(dim=1)
    return torch.clamp_min(v, 0.0)


def evaluate_sharded(evaluate: Callable[[torch.Tensor], torch.Tensor], Z_all: torch.Tensor, g_lb: torch.Tensor,
                     g_ub: torch.Tensor, constr_viol_tol: float = 1e-4, gather_g: bool = False,
                     group=None, violation_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                     after_eval_event=None, mid_event=None, workspace: Optional[dict] = None,
                     violation_into: Optional[Callable[[torch.Tensor, torch.Tensor], None]] = None
                     ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """Evaluate this rank's block of `Z_all` with `evaluate` (Z_block -> G_block, on this rank's device) and gather
    per-problem records on every rank.

    `violation_fn` (G_block -> violation per problem) replaces the torch formula of `violation`, e.g. by the library's own
    kernel (`ProblemHandle.violation_device` = raptor_violation_batch).
    `violation_into` (G_block, out[n]) writes the record in place instead (no copy); `workspace`: a dict the caller keeps between
    sweeps, holding the record and gather buffers (a sweep then allocates nothing; the returned tensors are views of them and are
    overwritten by the next sweep).
    `after_eval_event` / `mid_event` (optional torch.cuda.Event): recorded on the current stream after the evaluation and after the
    record, i.e. before the collective.
    Returns (violation[B], feasible[B] (bool), G[B, m] or None), identical on all ranks.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = Z_all.shape[0]
    lo, hi = shard_range(B, world, rank)
    per = -(-B // world) if world > 0 else B
    G = evaluate(Z_all[lo:hi].contiguous())
    if after_eval_event is not None:  # (torch.cuda.Event: splits a sweep into evaluation | record | collective on the device's clock)
        after_eval_event.record()
    m = g_lb.numel()
    dev = G.device
    # fixed-size record per problem: [violation, g...] padded to the block size so one all_gather suffices
    width = 1 + (m if gather_g else 0)
    n = hi - lo
    key = (per, width, world, str(dev))
    if workspace is not None and workspace.get("key") == key:  # record and gather buffers of the previous sweep, reused
        rec, out_buf = workspace["rec"], workspace["out"]
        if n < per:
            rec[n:].zero_()
    else:
        rec = torch.zeros((per, width), dtype=torch.float64, device=dev)
        out_buf = torch.empty((world * per, width), dtype=torch.float64, device=dev) if dist.is_initialized() else None
        if workspace is not None:
            workspace.update(key=key, rec=rec, out=out_buf)
    if n > 0:
        if violation_into is not None and width == 1:
            violation_into(G, rec[:n, 0])  # the library's kernel writes the record in place (a contiguous run of n doubles)
        else:
            rec[:n, 0] = violation_fn(G) if violation_fn is not None else violation(G, g_lb.to(dev), g_ub.to(dev))
        if gather_g:
            rec[:n, 1:] = G
    if mid_event is not None:
        mid_event.record()
    if dist.is_initialized():  # also with ONE rank: the collective is the path's exchange step and runs whenever a group exists
        out = out_buf
        dist.all_gather_into_tensor(out, rec, group=group)
    else:
        out = rec
    out = out[:B]
    viol = out[:, 0]
    return viol, viol <= constr_viol_tol, (out[:, 1:] if gather_g else None)


def select_sharded(evaluate: Callable[[torch.Tensor], torch.Tensor], cost: Callable[[torch.Tensor], torch.Tensor],
                   Z_all: torch.Tensor, g_lb: torch.Tensor, g_ub: torch.Tensor, constr_viol_tol: float = 1e-4, group=None) -> dict:
    """The multi-start record of SURVEY.md 8e -- {cost f, worst violation, feasible} per problem -- and the winner: the feasible
    problem of least cost, which is what the reference keeps across its random restarts (the Optimizer's own rule for iterates,
    Optimization/src/Optimizer.cpp:187-193, applied across problems).  `cost`: Z_block -> f_block on this rank's device
    (raptor_kinova_eval_objective / raptor_eval_costs weighted by the caller).  One all-gather of two doubles per problem.

    Returns {"violation": [B], "feasible": [B] bool, "cost": [B], "best": index or -1}, identical on all ranks."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = Z_all.shape[0]
    lo, hi = shard_range(B, world, rank)
    per = -(-B // world) if world > 0 else B
    Zb = Z_all[lo:hi].contiguous()
    G = evaluate(Zb)
    rec = torch.zeros((per, 2), dtype=torch.float64, device=G.device)
    n = hi - lo
    if n > 0:
        rec[:n, 0] = violation(G, g_lb.to(G.device), g_ub.to(G.device))
        rec[:n, 1] = cost(Zb).to(G.device).reshape(-1)
    if dist.is_initialized():
        out = torch.empty((world * per, 2), dtype=torch.float64, device=G.device)
        dist.all_gather_into_tensor(out, rec, group=group)
    else:
        out = rec
    out = out[:B]
    feas = out[:, 0] <= constr_viol_tol
    best = -1
    if bool(feas.any()):
        masked = torch.where(feas, out[:, 1], torch.full_like(out[:, 1], float("inf")))
        best = int(torch.argmin(masked))  # first minimum: lowest problem index on a tie, on every rank alike
    return {"violation": out[:, 0], "feasible": feas, "cost": out[:, 1], "best": best}


def timed_sweep(evaluate: Callable[[torch.Tensor], torch.Tensor], Z_all: torch.Tensor, g_lb: torch.Tensor, g_ub: torch.Tensor,
                steps: int, warmup: int, sync: Optional[Callable[[], None]] = None, constr_viol_tol: float = 1e-4, group=None,
                violation_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                violation_into: Optional[Callable[[torch.Tensor, torch.Tensor], None]] = None) -> dict:
    """The multi-start sweep as a measured job: `steps` timed passes of `evaluate_sharded` (own block evaluated, per-problem
    record built, ONE all-gather) between barriers, after `warmup` untimed ones.  `sync` waits for this rank's device
    (torch.cuda.synchronize on a GPU; None on the CPU).  Returns, identically on every rank,
    {"elapsed_s": max over ranks, "per_rank_s": [...], "ranks_seen": all-reduced count of ranks, "violation", "feasible"}."""
    import time

    import torch.distributed as dist

    on = dist.is_initialized()

    def barrier():
        if on:
            dist.barrier(group)
        if sync is not None:
            sync()

    out = None
    ws: dict = {}
    for _ in range(max(warmup, 0)):
        out = evaluate_sharded(evaluate, Z_all, g_lb, g_ub, constr_viol_tol, False, group, violation_fn, None, None, ws, violation_into)
    # the split of a sweep on the device's own clock: events on the current stream before the evaluation, after it, after the
    # record and after the collective of every timed sweep (a GPU batch only; the CPU rehearsals under gloo have no events)
    cuda = bool(Z_all.is_cuda)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)] if cuda else []
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        if cuda:
            evs[k][0].record()
            out = evaluate_sharded(evaluate, Z_all, g_lb, g_ub, constr_viol_tol, False, group, violation_fn, evs[k][1], evs[k][2], ws, violation_into)
            evs[k][3].record()
        else:
            out = evaluate_sharded(evaluate, Z_all, g_lb, g_ub, constr_viol_tol, False, group, violation_fn, None, None, ws, violation_into)
    barrier()
    dt = time.perf_counter() - t0
    split = None
    if cuda and steps > 0:
        split = {"evaluation_ms": sum(e[0].elapsed_time(e[1]) for e in evs) / steps,
                 "record_ms": sum(e[1].elapsed_time(e[2]) for e in evs) / steps,
                 "gather_ms": sum(e[2].elapsed_time(e[3]) for e in evs) / steps}
    dev = out[0].device if out is not None else Z_all.device
    mine = torch.tensor([dt], dtype=torch.float64, device=dev)
    one = torch.ones(1, dtype=torch.float64, device=dev)
    if on:
        world = dist.get_world_size(group)
        allt = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, mine, group=group)
        dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
    else:
        allt = mine
    per_rank = [float(x) for x in allt.cpu()]
    return {"elapsed_s": max(per_rank), "per_rank_s": per_rank, "ranks_seen": int(round(float(one.item()))), "split_ms": split,
            "violation": None if out is None else out[0], "feasible": None if out is None else out[1]}
