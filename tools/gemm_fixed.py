#!/usr/bin/env python
"""Tile turn-around of the 256x256 kernels: exactly 4 rounds of tiles (M = 16384, N = 4096), K from 128 to 2048, so that
time / 4 = fixed + (K / 64) * step.  Variants: plain bf16 store, + bias, + bias + GELU(erf), + bias + residual.  TUNE env = tune bits."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev, BF = "cuda:0", torch.bfloat16
TUNE = int(os.environ.get("TUNE", 0))
M, N = 16384, 4096
g = torch.Generator(device=dev).manual_seed(0)
def t(fn, n=30):
    for _ in range(8): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
b = torch.randn(N, device=dev, generator=g).to(BF); r = torch.randn(M, N, device=dev, generator=g).to(BF)
out = torch.empty(M, N, device=dev, dtype=BF)
for K in (128, 256, 1024):
    x = torch.randn(M, K, device=dev, generator=g).to(BF)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(BF)
    ops.register_tiled(w)
    with ops.streamk_policy(None):
        a0 = t(lambda: ops.linear(x, w, out=out, tune=TUNE))
        a1 = t(lambda: ops.linear(x, w, b, out=out, tune=TUNE))
        a2 = t(lambda: ops.linear(x, w, b, act="gelu", out=out, tune=TUNE))
        a3 = t(lambda: ops.linear(x, w, b, residual=r, out=out, tune=TUNE))
        a4 = t(lambda: ops.linear(x, w, b, act="quick_gelu", out=out, tune=TUNE))
        a5 = t(lambda: ops.linear(x, w, swiglu=True, out=out[:, :N // 2], tune=TUNE))
    print(f"K={K:5d} ({K // 64:2d} steps)  per round: plain {a0 / 4:6.2f} us  bias {a1 / 4:6.2f}  bias+gelu {a2 / 4:6.2f}  bias+resid {a3 / 4:6.2f}"
          f"  bias+quick_gelu {a4 / 4:6.2f}  swiglu {a5 / 4:6.2f}")
