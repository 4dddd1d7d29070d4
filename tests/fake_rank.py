"""Stand-in for bench.py's rank body in the launcher test (CPU, gloo): proves that `bench.py --gpus N` started N
ranks that can talk to each other; rank 0 prints the one JSON line the parent forwards."""
import json
import os
import sys

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
t = torch.ones(1)
dist.all_reduce(t)
if os.environ.get("FAKE_RANK_FAIL") == os.environ["RANK"]:
    sys.exit(7)
if dist.get_rank() == 0:
    print(json.dumps({"world_env": int(os.environ["WORLD_SIZE"]), "ranks_counted": int(t.item()), "argv": sys.argv[1:],
                      "master": os.environ.get("MASTER_ADDR")}), flush=True)
dist.destroy_process_group()
