#!/usr/bin/env python
"""The four GEMMs of a SAM ViT-H block with their real epilogues, at the RES workload's sizes (8 images: M = 32768 tokens in the global
blocks, 39200 in the 14 x 14-window blocks), beside the vendor BLAS on the bare GEMM (yardstick only).  TUNE=<bits> forces a kernel."""
import importlib
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
TUNE = int(os.environ.get("TUNE", 0))
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, it=40):
    for _ in range(40):          # the clock settles over the first ~10 ms of a new kernel mix: a short warm-up reads 3-10 % slow
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


tot = {}
for M in (32768, 39200):
    for name, N, K, kw in (("qkv", 3840, 1280, {}), ("proj", 1280, 1280, {"residual": True}), ("fc1", 5120, 1280, {"act": "gelu"}),
                           ("fc2", 1280, 5120, {"residual": True})):
        x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
        res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if kw.get("residual") else None
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.register_tiled(w)
        if name == "proj" and M == 39200:
            res = None                           # the windowed blocks add the residual in window_unpartition_add
        t = timeit(lambda: ops.linear(x, w, b, act=kw.get("act"), residual=res, out=out, tune=TUNE))
        tp = timeit(lambda: ops.linear(x, w, out=out, tune=TUNE))
        tv = timeit(lambda: F.linear(x, w))
        fl = 2.0 * M * N * K
        tot[M] = tot.get(M, 0) + t
        print(f"M={M} {name:5s} N={N:5d} K={K:5d}  ours {t:7.1f} us {fl / t / 1e6:7.1f} TF/s   bare {tp:7.1f} us {fl / tp / 1e6:7.1f}   vendor bare {tv:7.1f} us {fl / tv / 1e6:7.1f}")
print({k: round(v, 1) for k, v in tot.items()})
