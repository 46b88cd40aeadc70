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

`param_groups` is torch.optim's surface (a list of dicts with "params", "lr", "betas", "eps", "weight_decay"): `step()` reads the
hyper-parameters from the group on every call, so `torch.optim.lr_scheduler.*` / HF `get_scheduler` (the reference: cosine with
warmup_ratio 0.03, configs/train/ullava.yaml:143-144) attach to it unchanged; `no_decay_groups()` builds HF Trainer's split (no weight
decay on biases and norm weights).  `state_dict()` / `load_state_dict()` hold THIS rank's shard of master / m / v plus the step count and
the hyper-parameters (DeepSpeed's per-rank `*_optim_states.pt`, reference save_steps 5000).

World size 1 (or no process group) is the same code without the two collectives.  Gradient clipping (`max_grad_norm`, HF default 1.0):
the squared norm is reduced on the device (`ull_sumsq_f32`) and all-reduced as one scalar.  Memory per rank for LLaMA-7B at world 8:
13.5 GB parameters + 13.5 GB gradients + 10.1 GB of sharded fp32 state (80.9 GB at world 1) -- sized for 288 GB of HBM.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .dist import _flat_buckets


class ShardedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 2e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_grad_norm: Optional[float] = 1.0, bucket_bytes: int = 512 << 20, group=None, force_collectives: bool = False):
        """params: an iterable of parameters, or torch.optim-style groups [{"params": [...], "lr": ..., "weight_decay": ...}, ...]."""
        params = list(params)
        if params and isinstance(params[0], dict):
            groups = [dict(g) for g in params]
        else:
            groups = [{"params": params}]
        defaults = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=float(weight_decay))
        kept = []
        for g in groups:
            g = dict(g)
            g["params"] = [p for p in g["params"] if p.requires_grad]
            if g["params"]:
                kept.append(g)
        if not kept:
            raise ValueError("ShardedAdamW: no trainable parameter")
        # torch.optim.Optimizer owns `param_groups` / `defaults` (duplicate checks, lr schedulers, HF Trainer's logging read them); the
        # per-parameter `state` of the base class stays empty -- the state here is per BUCKET SHARD, see state_dict()
        super().__init__(kept, defaults)
        self.params: List[torch.nn.Parameter] = [p for g in self.param_groups for p in g["params"]]
        self._check_device()
        self.max_grad_norm = max_grad_norm
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.force_collectives = bool(force_collectives) and self.distributed      # run the exchange even at world size 1 (GPU-box test of RCCL)
        self.step_count = 0
        self.buckets = []
        for gi, g in enumerate(self.param_groups):                                   # a bucket never spans two groups (one lr / decay per launch)
            for plist in _flat_buckets(g["params"], bucket_bytes):
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
                self.buckets.append(dict(params=plist, numel=numel, shard=shard, flat=flat, recv=None, group=gi,
                                         master=mine.float().clone(), m=torch.zeros(shard, device=dev, dtype=torch.float32),
                                         v=torch.zeros(shard, device=dev, dtype=torch.float32)))

    @staticmethod
    def no_decay_groups(named_parameters, weight_decay: float):
        """HF Trainer.create_optimizer's split (transformers trainer.py `get_decay_parameter_names`): weight decay on every trainable
        parameter except biases and normalisation weights (here: every 1-D parameter and every `*.bias`)."""
        decay, no_decay = [], []
        for n, p in named_parameters:
            if p.requires_grad:
                (no_decay if (p.dim() <= 1 or n.endswith(".bias") or "norm" in n.lower()) else decay).append(p)
        return [g for g in ({"params": decay, "weight_decay": float(weight_decay)}, {"params": no_decay, "weight_decay": 0.0}) if g["params"]]

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

    def state_dict(self) -> dict:
        """THIS rank's optimizer state (every rank saves its own, as DeepSpeed ZeRO does): step count, hyper-parameters per group, and per
        bucket the owned shard of the fp32 master weights and both moments (on the host)."""
        return {"step": self.step_count, "world": self.world, "rank": self.rank,
                "param_groups": [{**{k: v for k, v in g.items() if k != "params"}, "n_params": len(g["params"])} for g in self.param_groups],
                "buckets": [{"numel": b["numel"], "shard": b["shard"], "group": b["group"], "master": b["master"].cpu().clone(),
                             "m": b["m"].cpu().clone(), "v": b["v"].cpu().clone()} for b in self.buckets]}

    @torch.no_grad()
    def load_state_dict(self, sd: dict) -> None:
        """Resume from `state_dict()` of an optimizer built over the same parameter list at the same world size and rank.  The 16-bit
        parameters are re-derived from the restored master shard on the next step's all-gather; here the owned slice is rounded back into
        the flat buffer so that `step()` starts from the restored weights."""
        if sd["world"] != self.world or sd["rank"] != self.rank:
            raise ValueError(f"ShardedAdamW.load_state_dict: state of rank {sd['rank']}/{sd['world']} loaded on rank {self.rank}/{self.world}")
        if len(sd["buckets"]) != len(self.buckets) or len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("ShardedAdamW.load_state_dict: bucket / group layout differs (other parameter list or bucket_bytes)")
        for b, s in zip(self.buckets, sd["buckets"]):
            if (b["numel"], b["shard"], b["group"]) != (s["numel"], s["shard"], s["group"]):
                raise ValueError("ShardedAdamW.load_state_dict: bucket sizes differ")
        for g, s in zip(self.param_groups, sd["param_groups"]):
            if len(g["params"]) != s["n_params"]:
                raise ValueError("ShardedAdamW.load_state_dict: group sizes differ")
            g.update({k: (tuple(v) if k == "betas" else v) for k, v in s.items() if k != "n_params"})
        for b, s in zip(self.buckets, sd["buckets"]):
            for k in ("master", "m", "v"):
                b[k].copy_(s[k])
            b["flat"][self.rank * b["shard"]:(self.rank + 1) * b["shard"]].copy_(b["master"])
        self.step_count = int(sd["step"])

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
    def step(self, closure=None) -> Optional[float]:
        """One optimizer step; returns the global gradient norm (a python float) when clipping is on, else None."""
        if closure is not None:
            raise NotImplementedError("ShardedAdamW.step: closures are not used by the reference's training scripts")
        self.step_count += 1
        direct = self.world > 1 or self.force_collectives
        inv_world = 1.0 / self.world
        shards = []
        for b in self.buckets:
            flat = self._pack_grads(b)
            if direct and flat.dtype in (torch.bfloat16, torch.float16):
                if b["recv"] is None:
                    b["recv"] = torch.empty_like(flat)
                dist.all_to_all_single(b["recv"], flat, group=self.group)          # shard j of every rank lands on rank j
                # the MEAN over ranks: summed in fp32 in rank order, scaled by 1 / world in fp32, rounded to 16 bits once (an un-averaged
                # 16-bit sum over 8 ranks can overflow fp16 where the mean cannot; same form as dist.allreduce_gradients)
                shards.append(ops.sum_slabs(b["recv"].view(self.world, b["shard"]), inv_world))
            elif direct:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                shards.append(flat[self.rank * b["shard"]:(self.rank + 1) * b["shard"]] * inv_world)
            else:
                shards.append(flat)
        scale = 1.0
        norm = None
        if self.max_grad_norm is not None and self.max_grad_norm > 0:
            sq = torch.zeros(1, device=shards[0].device, dtype=torch.float32)
            for g in shards:
                ops.sumsq(g.contiguous(), sq)                                     # every rank sums ITS shards: the shards partition the gradient
            if direct:
                dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.group)
            norm = float(sq.sqrt())                                               # norm of the averaged gradient (host read: HF logs it too)
            scale = min(1.0, self.max_grad_norm / (norm + 1e-6))                  # torch.nn.utils.clip_grad_norm_'s coefficient
        for b, g in zip(self.buckets, shards):
            hp = self.param_groups[b["group"]]
            mine = b["flat"][self.rank * b["shard"]:(self.rank + 1) * b["shard"]]
            ops.adamw_step(b["master"], b["m"], b["v"], g.contiguous(), mine, hp["lr"], hp["betas"][0], hp["betas"][1], hp["eps"], hp["weight_decay"],
                           self.step_count, scale)
            if direct:
                dist.all_gather_into_tensor(b["flat"], mine.clone(), group=self.group)
            o = 0
            for p in b["params"]:
                p.copy_(b["flat"][o:o + p.numel()].view_as(p))
                o += p.numel()
        return norm
