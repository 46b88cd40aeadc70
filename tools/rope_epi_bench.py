#!/usr/bin/env python
"""Cost of the fused RoPE epilogue: the LLaMA q|k|v GEMM (M = 20576, N = 12288, K = 4096) with and without it, and o_proj / down with
and without their residual epilogue.  40 timed launches after 10 warm-ups each, alternating."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev, BF = "cuda:0", torch.bfloat16
T, D = 20576, 4096
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(T, D, device=dev, generator=g).to(BF)
w = (torch.randn(3 * D, D, device=dev, generator=g) * D ** -0.5).to(BF); ops.register_tiled(w)
wo = (torch.randn(D, D, device=dev, generator=g) * D ** -0.5).to(BF); ops.register_tiled(wo)
res = torch.randn(T, D, device=dev, generator=g).to(BF)
pos = (torch.arange(T, device=dev) % 643)
inv = (1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float, device=dev) / 128)))
cs, sn = ops.rope_table(pos, inv, BF)
out = torch.empty(T, 3 * D, device=dev, dtype=BF); out2 = torch.empty(T, D, device=dev, dtype=BF)
def t(fn, n=40):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(2):
    a = t(lambda: ops.linear(x, w, out=out)); b = t(lambda: ops.linear_qkv_rope(x, w, cs, sn, 2 * D, 128, out=out))
    c = t(lambda: ops.linear(x, wo, out=out2)); d = t(lambda: ops.linear(x, wo, residual=res, out=out2))
    print(f"qkv plain {a:7.1f} us   qkv + RoPE {b:7.1f} us (+{b - a:5.1f})     o_proj plain {c:6.1f} us   + residual {d:6.1f} us (+{d - c:4.1f})")
