"""Rank body used by tests/test_launcher.py: started by raptor_amd.launch.run_ranks (torch.distributed.run), gloo on the CPU."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raptor_amd.launch import dist_env  # noqa: E402

rank, local_rank, world = dist_env()
dist.init_process_group("gloo", rank=rank, world_size=world)
one = torch.ones(1, dtype=torch.float64)
dist.all_reduce(one)
json.dump({"rank": rank, "local_rank": local_rank, "world": world, "ranks_seen": int(one.item()),
           "master": os.environ.get("MASTER_ADDR")}, open(os.path.join(sys.argv[1], f"probe{rank}.json"), "w"))
dist.destroy_process_group()
