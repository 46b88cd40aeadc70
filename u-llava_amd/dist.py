"""Data-parallel harness for the forward path: one process per GPU, images sharded across ranks, no activation or weight
traffic.  The only collective on the path is the scalar aggregation below (reference: `evaluation/tools.py:94-115`
`AverageMeter.all_reduce`, the single explicit `dist.all_reduce` in the reference).  Backend "nccl" is RCCL over xGMI on
MI355X; "gloo" is used by the CPU tests.

Training adds the gradient synchronisation the reference delegates to DeepSpeed ZeRO-2 / DDP
(`configs/deepspeed/bf16_zero2.json:5-11`, `shells/finetune.sh:3`): `allreduce_gradients` below."""
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
    if not (dist.is_available() and dist.is_initialized()):
        return units_local / elapsed_local, units_local, elapsed_local
    # (a process group of ONE rank still goes through the collective: bench.py --init-pg runs RCCL on a single-GPU box this way)
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    u = torch.tensor([units_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item() / t.item()), float(u.item()), float(t.item())


def elapsed_spread(elapsed_local: float, device=None) -> Tuple[float, float]:
    """(min, max) over ranks of one rank-local elapsed time: the same scalar all-reduce as global_rate (MIN and MAX), so that a
    scaling record shows WHERE a slow whole-job step comes from (the job is as slow as its slowest rank)."""
    if not (dist.is_available() and dist.is_initialized()):
        return elapsed_local, elapsed_local
    lo = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo.item()), float(hi.item())


def _flat_buckets(grads, bucket_bytes: int):
    """Group gradient tensors (same dtype per bucket, declaration order) into buckets of at most `bucket_bytes`."""
    buckets, cur, size = [], [], 0
    for g in grads:
        n = g.numel() * g.element_size()
        if cur and (size + n > bucket_bytes or g.dtype != cur[0].dtype):
            buckets.append(cur)
            cur, size = [], 0
        cur.append(g)
        size += n
    if cur:
        buckets.append(cur)
    return buckets


_DIRECT_DTYPES = (torch.bfloat16, torch.float16)          # element types `ull_sum_slabs` has a build for
_KERNELS = None                                           # module that provides sum_slabs(); None = u-llava_amd.ops (the HIP kernels)


def _kernels():
    """The kernel module of the direct exchange.  `ops` (HIP only) unless a test has put a restatement in `_KERNELS` to drive the
    collective sequence itself over gloo on CPU ranks; the product never sets it."""
    if _KERNELS is not None:
        return _KERNELS
    from . import ops
    return ops


def allreduce_gradients(params, bucket_bytes: int = 512 << 20, group=None, force_direct: bool = False) -> int:
    """Average `.grad` of `params` over the data-parallel ranks; returns the number of buckets exchanged.

    MI355X design: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring is bound by one link.  Each bucket (default
    512 MiB, sized for 288 GB of HBM: ~27 buckets for LLaMA-7B) is reduced by DIRECT EXCHANGE: one all-to-all in which every rank
    sends shard j of its bucket to rank j over the link to j (all 7 links busy at once), a local fp32-accumulated sum of the N
    received shards (HIP kernel `ull_sum_slabs`, scaled by 1/N), then one all-gather of the reduced shards -- the reduce-scatter +
    all-gather split of ZeRO-2 (reference configs/deepspeed/bf16_zero2.json: stage 2, reduce_bucket_size 5e8) with 2 (N-1)/N x bucket
    bytes per GPU.  Process groups without all-to-all (gloo on CPU: the N > 1 unit tests) and gradient dtypes the kernel has no build
    for (fp32 master gradients) take one all_reduce per bucket.

    The bucket layout is a function of the PARAMETER LIST alone, never of which gradients happen to exist: a rank whose batch had no
    [SEG] / [LOC] row (or no image) leaves the seg / det heads, the mask decoder or the projector without a `.grad`; such a parameter
    enters the exchange as zeros and gets the averaged gradient written back, exactly as if it had produced a zero gradient (the
    reference keeps its collectives aligned the same way: the decoder always runs, text-only samples add a 0-weighted projector term,
    models/ullava_core.py:213-220).  Parameters with requires_grad == False are skipped on every rank alike.

    force_direct: take the direct-exchange branch whatever the backend and even at world size 1 (where it must be the identity up
    to the rounding of x * 1.0): lets a single-GPU box execute the RCCL calls and the kernel."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1 and not force_direct:
        return 0
    plist = [p for p in params if p.requires_grad]
    for p in plist:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in plist]
    direct_backend = force_direct or dist.get_backend(group) == "nccl"
    n_buckets = 0
    flat = recv = None
    for bucket in _flat_buckets(grads, bucket_bytes):
        numel = sum(g.numel() for g in bucket)
        shard = -(-numel // world)
        shard = -(-shard // 8) * 8                                  # 16-byte aligned shards
        dt, dev = bucket[0].dtype, bucket[0].device
        if flat is None or flat.numel() < shard * world or flat.dtype != dt or flat.device != dev:
            flat = torch.empty(shard * world, device=dev, dtype=dt)     # kept across buckets (the first one is the largest of its dtype)
            recv = None
        fb = flat[:shard * world]
        o = 0
        for g in bucket:
            fb[o:o + g.numel()].copy_(g.reshape(-1))
            o += g.numel()
        fb[o:].zero_()
        if direct_backend and dt in _DIRECT_DTYPES and (dev.type == "cuda" or force_direct):
            ops = _kernels()
            if recv is None:
                recv = torch.empty_like(flat)
            rb = recv[:shard * world]
            dist.all_to_all_single(rb, fb, group=group)             # shard j of every rank lands on rank j
            mine = ops.sum_slabs(rb.view(world, shard), 1.0 / world)
            dist.all_gather_into_tensor(fb, mine, group=group)
        else:
            dist.all_reduce(fb, op=dist.ReduceOp.SUM, group=group)
            fb /= world
        o = 0
        for g in bucket:
            g.copy_(fb[o:o + g.numel()].view_as(g))
            o += g.numel()
        n_buckets += 1
    return n_buckets
