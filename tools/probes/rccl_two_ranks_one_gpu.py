"""Probe (VERDICT r3 next#2c): can RCCL run N > 1 ranks on ONE MI355X?  Launch: python -m torch.distributed.run --nproc-per-node 2
--master-addr 127.0.0.1 --master-port 29655 tools/probes/rccl_two_ranks_one_gpu.py  (run under `timeout`).  Every rank uses device 0.
Prints one line per rank: OK with the all-reduce result, or the error RCCL raised (expected: "Duplicate GPU detected")."""
import os
import sys

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((1024,), float(rank + 1), device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print(f"rank {rank}: OK all_reduce -> {float(x[0])}", flush=True)
    dist.destroy_process_group()
except Exception as e:                                  # noqa: BLE001 -- the probe's whole point is to report the error text
    print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:600]}", flush=True)
    sys.exit(3)
