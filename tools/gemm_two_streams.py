#!/usr/bin/env python
"""Experiment: one large Linear as TWO concurrent launches over disjoint column ranges on two streams, optionally with different
kernel forms (tile periods that differ drift out of phase, so one launch's round-end store burst falls into the other's K-loop)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
W4, W8 = 1 << 22, 1 << 21


def timeit(fn, it=40, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, M, N, K in (("qkv", 20576, 12288, 4096), ("down", 20576, 4096, 11008), ("sam qkv", 32768, 3840, 1280), ("sam fc2", 32768, 1280, 5120)):
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    ops.register_tiled(w)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_one = timeit(lambda: ops.linear(x, w, out=out))
    res = [f"{name}: one launch {t_one:7.1f} us"]
    for frac, t2 in ((0.5, W4), (0.5, W8), (0.625, W8)):
        n1 = int(N * frac) // 256 * 256
        wa, wb = w[:n1].contiguous(), w[n1:].contiguous()
        ops.register_tiled(wa); ops.register_tiled(wb)

        def two():
            main = torch.cuda.current_stream()
            s1.wait_stream(main); s2.wait_stream(main)
            with torch.cuda.stream(s1):
                ops.linear(x, wa, out=out[:, :n1], tune=W4)
            with torch.cuda.stream(s2):
                ops.linear(x, wb, out=out[:, n1:], tune=t2)
            main.wait_stream(s1); main.wait_stream(s2)
        res.append(f"split {frac:.3f} w4 | {'w4' if t2 == W4 else 'w8'}: {timeit(two):7.1f} us")
    print("   ".join(res), flush=True)
