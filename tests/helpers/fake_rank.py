"""Stand-in for one rank of bench.py (tests/test_bench_launch.py): joins the gloo group the launcher described in the environment,
takes part in one all_reduce and, on rank 0, prints a bench-shaped JSON line.  FAKE_FAIL_RANK=R in the environment makes rank R exit non-zero at the end, FAKE_DIE_EARLY_RANK=R before the rendezvous."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--warmup", type=int, default=0)
a, _ = ap.parse_known_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if rank == int(os.environ.get("FAKE_DIE_EARLY_RANK", "-1")):      # dies before it ever joins the group: the others would wait in the rendezvous
    sys.exit(4)
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank and world == a.gpus
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.barrier()
dist.destroy_process_group()
if rank == int(os.environ.get("FAKE_FAIL_RANK", "-1")):
    sys.exit(3)
if rank == 0:
    print("some log line before the result")
    print(json.dumps({"metric": "fake", "value": float(t.item()), "n_gpus": world, "steps": a.steps, "warmup": a.warmup}))
