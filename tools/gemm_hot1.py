#!/usr/bin/env python
"""One sustained leg (LLaMA-layer GEMMs back to back for SECONDS_PER_LEG) of whatever library ULL_LIB_PATH points at; prints one line.
argv[1] = label, argv[2] = tune bits (optional)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
T = 20576
SECONDS = float(os.environ.get("SECONDS_PER_LEG", 3))
tune = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator(device=dev).manual_seed(0)
L = []
for name, N, K, sw in [("qkv", 12288, 4096, False), ("o", 4096, 4096, False), ("gateup", 22016, 4096, True), ("down", 4096, 11008, False)]:
    x = torch.randn(T, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    ops.register_tiled(w)
    L.append((x, w, torch.empty(T, N // 2 if sw else N, device=dev, dtype=torch.bfloat16), sw, 2.0 * T * N * K))
flops = sum(o[-1] for o in L)
def layer():
    for x, w, out, sw, _ in L:
        ops.linear(x, w, swiglu=sw, out=out, tune=tune)
for _ in range(20):
    layer()
torch.cuda.synchronize()
t0 = time.time(); rates = []
while time.time() - t0 < SECONDS:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        layer()
    e1.record(); e1.synchronize()
    rates.append(flops * 20 / e0.elapsed_time(e1) / 1e9)
print(f"{sys.argv[1]:48s} sustained {sum(rates) / len(rates):7.1f} TF/s  (min {min(rates):7.1f}, max {max(rates):7.1f}, {len(rates) * 20} layers)")
