"""A stand-in rank for tests/test_bench_launcher.py: what bench.py's self-launcher must provide to a child -- RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_ADDR/PORT for a torch.distributed rendezvous -- checked by actually using it (gloo)."""
import json
import os
import sys

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    print("rank", rank, "fails before the rendezvous", file=sys.stderr)
    sys.exit(3)
import torch
import torch.distributed as dist

dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([rank + 1.0])
dist.all_reduce(t)
print(f"banner line of rank {rank} (not JSON)")
if rank == 0:
    print(json.dumps({"world": world, "sum": float(t.item()), "argv": sys.argv[1:]}), flush=True)
dist.destroy_process_group()
