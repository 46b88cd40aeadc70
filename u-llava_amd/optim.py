"""ZeRO-stage-2 AdamW for the training path.

The reference trains under `transformers.Trainer` + DeepSpeed ZeRO stage 2 in bf16 (`train_ullava.py:273-293`,
`configs/deepspeed/bf16_zero2.json:5-11`, `configs/train/ullava.yaml:139-156`: AdamW, lr 2e-5, weight decay 0): every rank holds the
16-bit parameters and its gradients, but the fp32 master weights and the two fp32 moments -- 12 of the 16 bytes per parameter -- exist only on
the rank that OWNS the shard; gradients are reduce-scattered to their owners, the owners update, the updated 16-bit parameters are
all-gathered.  `ShardedAdamW.step()` is that cycle on MI355X:

  * the parameter list is flattened into buckets (same rule as `dist.allreduce_gradients`: layout a function of the parameter list alone);
    a bucket is cut into `world` equal shards of a multiple of 8 elements;
  * gradients -> one flat 16-bit buffer; `all_to_all_single` sends shard j of every rank to rank j over the direct xGMI link to j (all
    seven links of a GPU busy at once -- a ring serialises on one), `ull_sum_slabs` sums the `world` received slabs in rank order in fp32;
  * `ull_adamw_step_f32` updates master / m / v of the shard (torch's AdamW arithmetic in fp32; the 1 / world average and the clipping
    coefficient ride in as one gradient scale) and writes the shard's new 16-bit parameters;
  * `all_gather_into_tensor` hands every rank the whole updated bucket, which is scattered back into the parameters.

World size 1 (or no process group) is the same code without the two collectives.  Gradient clipping (`max_grad_norm`, HF default 1.0):
the squared norm is reduced on the device (`ull_sumsq_f32`) and all-reduced as one scalar.  Memory per rank for LLaMA-7B at world 8:
13.5 GB parameters + 13.5 GB gradients + 10.1 GB of sharded fp32 state (80.9 GB at world 1) -- sized for 288 GB of HBM.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .dist import _flat_buckets


class ShardedAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 2e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_grad_norm: Optional[float] = 1.0, bucket_bytes: int = 512 << 20, group=None, force_collectives: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("ShardedAdamW: no trainable parameter")
        self._check_device()
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.max_grad_norm = max_grad_norm
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.force_collectives = bool(force_collectives) and self.distributed      # run the exchange even at world size 1 (GPU-box test of RCCL)
        self.step_count = 0
        self.buckets = []
        for plist in _flat_buckets(self.params, bucket_bytes):
            numel = sum(p.numel() for p in plist)
            shard = -(-numel // self.world)
            shard = -(-shard // 8) * 8
            dev, dt = plist[0].device, plist[0].dtype
            flat = torch.zeros(shard * self.world, device=dev, dtype=dt)            # gradients in, updated parameters out
            o = 0
            for p in plist:
                flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
                o += p.numel()
            mine = flat[self.rank * shard:(self.rank + 1) * shard]
            self.buckets.append(dict(params=plist, numel=numel, shard=shard, flat=flat, recv=None,
                                     master=mine.float().clone(), m=torch.zeros(shard, device=dev, dtype=torch.float32),
                                     v=torch.zeros(shard, device=dev, dtype=torch.float32)))

    def _check_device(self):
        if any(not p.is_cuda for p in self.params):
            raise RuntimeError("u-llava_amd: ShardedAdamW updates on the GPU (no CPU path exists)")

    # -- torch.optim.Optimizer's surface as the training scripts use it -------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def state_bytes(self) -> int:
        return sum(3 * b["shard"] * 4 for b in self.buckets)

    def _pack_grads(self, b) -> torch.Tensor:
        flat, o = b["flat"], 0
        for p in b["params"]:
            n = p.numel()
            if p.grad is None:
                flat[o:o + n].zero_()                       # a head this rank's batch never touched: a zero gradient (dist.allreduce_gradients)
            else:
                flat[o:o + n].copy_(p.grad.reshape(-1))
            o += n
        flat[o:].zero_()
        return flat

    @torch.no_grad()
    def step(self) -> Optional[float]:
        """One optimizer step; returns the global gradient norm (a python float) when clipping is on, else None."""
        self.step_count += 1
        direct = self.world > 1 or self.force_collectives
        shards = []
        for b in self.buckets:
            flat = self._pack_grads(b)
            if direct and flat.dtype in (torch.bfloat16, torch.float16):
                if b["recv"] is None:
                    b["recv"] = torch.empty_like(flat)
                dist.all_to_all_single(b["recv"], flat, group=self.group)          # shard j of every rank lands on rank j
                shards.append(ops.sum_slabs(b["recv"].view(self.world, b["shard"]), 1.0))   # the SUM over ranks (averaged below, in fp32)
            elif direct:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                shards.append(flat[self.rank * b["shard"]:(self.rank + 1) * b["shard"]].clone())
            else:
                shards.append(flat)
        scale = 1.0 / self.world
        norm = None
        if self.max_grad_norm is not None and self.max_grad_norm > 0:
            sq = torch.zeros(1, device=shards[0].device, dtype=torch.float32)
            for g in shards:
                ops.sumsq(g.contiguous(), sq)                                     # every rank sums ITS shards: the shards partition the gradient
            if direct:
                dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.group)
            norm = float(sq.sqrt()) * scale                                       # norm of the averaged gradient (host read: HF logs it too)
            scale *= min(1.0, self.max_grad_norm / (norm + 1e-6))                 # torch.nn.utils.clip_grad_norm_'s coefficient
        for b, g in zip(self.buckets, shards):
            mine = b["flat"][self.rank * b["shard"]:(self.rank + 1) * b["shard"]]
            ops.adamw_step(b["master"], b["m"], b["v"], g.contiguous(), mine, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                           self.step_count, scale)
            if direct:
                dist.all_gather_into_tensor(b["flat"], mine.clone(), group=self.group)
            o = 0
            for p in b["params"]:
                p.copy_(b["flat"][o:o + p.numel()].view_as(p))
                o += p.numel()
        return norm
