#!/usr/bin/env python
"""Small-M launches (the RES workload's LLaMA stream at batch 8: M = 3032 tokens; the training step: M = 2584): 4 waves vs 8 waves."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev, BF = "cuda:0", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
def t(fn, n=30):
    for _ in range(8): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (3032, 2584, 1024):
    for name, N, K, sw in (("qkv", 12288, 4096, False), ("o", 4096, 4096, False), ("gateup", 22016, 4096, True), ("down", 4096, 11008, False)):
        x = torch.randn(M, K, device=dev, generator=g).to(BF)
        w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(BF); ops.register_tiled(w)
        out = torch.empty(M, N // 2 if sw else N, device=dev, dtype=BF)
        r = {lab: t(lambda: ops.linear(x, w, swiglu=sw, out=out, tune=tn)) for lab, tn in (("auto", 0), ("w8", 1 << 21), ("w4", 1 << 22))}
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        print(f"M={M:5d} {name:7s} tiles={tiles:5d}  " + "  ".join(f"{k} {v:7.1f} us ({2.0 * M * N * K / v / 1e6:6.0f} TF/s)" for k, v in r.items()))
