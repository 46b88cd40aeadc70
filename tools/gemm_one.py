#!/usr/bin/env python
"""Run one GEMM shape a few times (for rocprofv3 --pmc passes). usage: gemm_one.py M N K [swiglu]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
M, N, K = (int(v) for v in sys.argv[1:4])
sw = len(sys.argv) > 4
g = torch.Generator(device="cuda:0").manual_seed(0)
x = torch.randn(M, K, device="cuda:0", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda:0", generator=g) * K ** -0.5).to(torch.bfloat16)
ops.register_tiled(w)          # tile-major copy, as pack_weights() does for the model's Linears
out = torch.empty(M, N // 2 if sw else N, device="cuda:0", dtype=torch.bfloat16)
for _ in range(3):
    ops.linear(x, w, swiglu=sw, out=out, tune=int(os.environ.get('TUNE', 0)))
torch.cuda.synchronize()
