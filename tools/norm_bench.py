#!/usr/bin/env python
"""LayerNorm / RMSNorm kernel rate at the path's shapes (rows x D): SAM 32768 x 1280, CLIP 18464 x 1024, LLaMA 20576 x 4096."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev, BF = "cuda:0", torch.bfloat16
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, rows, D, ln in (("SAM LayerNorm", 32768, 1280, True), ("SAM windows LN", 39200, 1280, True), ("CLIP LayerNorm", 18464, 1024, True), ("LLaMA RMSNorm", 20576, 4096, False)):
    x = torch.randn(rows, D, device=dev).to(BF); w = torch.randn(D, device=dev).to(BF); b = torch.randn(D, device=dev).to(BF)
    y = torch.empty_like(x)
    us = t((lambda: ops.layernorm(x, w, b, 1e-6, out=y)) if ln else (lambda: ops.rmsnorm(x, w, 1e-6, out=y)))
    print(f"{name:16s} {rows} x {D}: {us:6.1f} us = {2 * rows * D * 2 / us / 1e6:.2f} TB/s")
