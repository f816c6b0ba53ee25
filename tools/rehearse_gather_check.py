#!/usr/bin/env python3
"""bench.py's multi-GPU gather check run with ONE rank over RCCL (GPU box; no multi-GPU node was available): the same
function, in a thread of its own as bench.py runs it, over a one-rank nccl process group.  Prints its result."""
import json, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
box = {}
th = threading.Thread(target=lambda: box.update(bench.gather_check(0, 1, 0)), daemon=True)
th.start()
th.join(120.0)
print(json.dumps({"timed_out": th.is_alive(), "result": box}))
if th.is_alive():
    os._exit(3)
dist.destroy_process_group()
