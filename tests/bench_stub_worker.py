"""Stands in for bench.py's per-rank part in the launcher's CPU test: joins the process group the launcher's
torch.distributed.run set up (gloo), and rank 0 prints one result line.  RT_STUB_FAIL_RANK makes that rank exit 3."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
print(f"rank {rank} says hello")  # not a result line: the launcher must pass it to stderr
if os.environ.get("RT_STUB_FAIL_RANK") == str(rank):
    sys.exit(3)
dist.barrier()
if rank == 0:
    print(json.dumps({"metric": "stub", "value": float(t.item()), "n_gpus": world, "argv": sys.argv[1:]}), flush=True)
dist.destroy_process_group()
