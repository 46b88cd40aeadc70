#!/usr/bin/env python
"""Sustained (power-limited) GEMM rate: the four LLaMA-layer launches back to back for SECONDS, for the shipped dispatch, the 8-wave
form, and -- yardstick only, never used by the product -- the vendor BLAS behind torch.nn.functional.linear."""
import importlib, os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
T = 20576
SECONDS = float(os.environ.get("SECONDS_PER_LEG", 4))
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("qkv", 12288, 4096, False), ("o", 4096, 4096, False), ("gateup", 22016, 4096, True), ("down", 4096, 11008, False)]
ops_ = []
for name, N, K, sw in shapes:
    x = torch.randn(T, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    ops.register_tiled(w)
    out = torch.empty(T, N // 2 if sw else N, device=dev, dtype=torch.bfloat16)
    vout = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    ops_.append((name, x, w, out, vout, sw, 2.0 * T * N * K))
flops = sum(o[-1] for o in ops_)
def layer(which):
    for name, x, w, out, vout, sw, _ in ops_:
        if which == "vendor":
            torch.matmul(x, w.t(), out=vout)
        else:
            ops.linear(x, w, swiglu=sw, out=out, tune=which)
for label, which in (("shipped", 0), ("8 waves", ops.GEMM_TUNE_WAVES8), ("4 waves", ops.GEMM_TUNE_WAVES4), ("vendor (no SwiGLU, 2x output)", "vendor"), ("shipped", 0)):
    for _ in range(3):
        layer(which)
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    rates = []
    while time.time() - t0 < SECONDS:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            layer(which)
        e1.record(); e1.synchronize()
        rates.append(flops * 20 / e0.elapsed_time(e1) / 1e9)
    print(f"{label:32s} first 20 layers {rates[0]:7.1f} TF/s   last 20 layers {rates[-1]:7.1f} TF/s   mean {sum(rates) / len(rates):7.1f}  ({len(rates) * 20} layers)")
