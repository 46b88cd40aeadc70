#!/usr/bin/env python
"""Fixed (M, N), sweep K: time = fixed_per_launch + per_kstep * K/64.  Separates K-loop speed from epilogue/tail costs."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
TUNE = int(os.environ.get("TUNE", 0))
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, sw in (("qkv", 20576, 12288, False), ("gateup", 20576, 22016, True), ("o/down", 20576, 4096, False), ("full27", 256 * 27, 256 * 256, False)):
    res = []
    for K in (1024, 2048, 4096, 8192):
        x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
        out = torch.empty(M, N // 2 if sw else N, device=dev, dtype=torch.bfloat16)
        for _ in range(2):
            ops.linear(x, w, swiglu=sw, out=out, tune=TUNE)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            ops.linear(x, w, swiglu=sw, out=out, tune=TUNE)
        e1.record(); e1.synchronize()
        res.append((K, e0.elapsed_time(e1) / 8))
    (k1, t1), (k2, t2) = res[1], res[3]
    per = (t2 - t1) / ((k2 - k1) / 64)
    fixed = t1 - per * k1 / 64
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"{name:8s} tiles={tiles:5d} rounds={tiles / 256:6.2f}  " + "  ".join(f"K={k}: {t:.3f} ms" for k, t in res) +
          f"   per K-step/launch {per * 1e3:.2f} us (= {per * 1e3 / (tiles / 256):.3f} us per round)  fixed {fixed * 1e3:.1f} us (= {fixed * 1e3 / (tiles / 256):.1f} us per round)")
