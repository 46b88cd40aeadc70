"""Data-parallel harness for the forward path: one process per GPU, images sharded across ranks, no activation or weight
traffic.  The only collective on the path is the scalar aggregation below (reference: `evaluation/tools.py:94-115`
`AverageMeter.all_reduce`, the single explicit `dist.all_reduce` in the reference).  Backend "nccl" is RCCL over xGMI on
MI355X; "gloo" is used by the CPU tests."""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl", device: torch.device = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's env; initialises the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world, local


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) slice of n_total independent units for this rank (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def global_rate(units_local: float, elapsed_local: float, device=None) -> Tuple[float, float, float]:
    """Whole-job throughput: (sum of units over ranks) / (max elapsed over ranks).  Returns (rate, total_units, max_elapsed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return units_local / elapsed_local, units_local, elapsed_local
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    u = torch.tensor([units_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item() / t.item()), float(u.item()), float(t.item())
