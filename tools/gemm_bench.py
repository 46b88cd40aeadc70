#!/usr/bin/env python
"""GEMM microbench on the LLaMA-7B / CLIP shapes of the C4 workload (M = 32 x 643 tokens). Random operands."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
T = int(os.environ.get("TOKENS", 20576))
shapes = [("qkv", T, 12288, 4096, False), ("o", T, 4096, 4096, False), ("gateup", T, 22016, 4096, True), ("down", T, 4096, 11008, False),
          ("lm_head", T, 32011, 4096, False), ("clip_qkv", 32 * 577, 3072, 1024, False), ("clip_fc1", 32 * 577, 4096, 1024, False),
          ("clip_fc2", 32 * 577, 1024, 4096, False), ("sq4096", 4096, 4096, 4096, False), ("sq8192", 8192, 8192, 8192, False),
          ("sam_qkv", 8 * 4096, 3840, 1280, False), ("sam_proj", 8 * 4096, 1280, 1280, False), ("sam_fc1", 8 * 4096, 5120, 1280, False),
          ("sam_fc2", 8 * 4096, 1280, 5120, False), ("K2048", T, 4096, 2048, False), ("K3072", T, 4096, 3072, False)]
TUNE = int(os.environ.get("TUNE", 0))           # ULL_GEMM_TUNE_* bits (e.g. 1 << 21: the 8-wave kernel)
TILED = int(os.environ.get("TILED", 0))         # register tile-major weight copies as pack_weights() does
g = torch.Generator(device=dev).manual_seed(0)
tot_f = tot_t = 0
for name, M, N, K, sw in shapes:
    PAD = int(os.environ.get("PADK", 0))          # row-stride experiment: operands become views of [rows, K + PAD] buffers
    x = torch.randn(M, K + PAD, device=dev, generator=g).to(torch.bfloat16)[:, :K]
    w = (torch.randn(N, K + PAD, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)[:, :K]
    out = torch.empty(M, N // 2 if sw else N, device=dev, dtype=torch.bfloat16)
    if TILED and not PAD:
        ops.register_tiled(w)
    for _ in range(2):
        ops.linear(x, w, swiglu=sw, out=out, tune=TUNE)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    it = 10
    for _ in range(it):
        ops.linear(x, w, swiglu=sw, out=out, tune=TUNE)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / it
    fl = 2.0 * M * N * K
    if name in ("qkv", "o", "gateup", "down"):
        tot_f += fl
        tot_t += ms
    print(f"{name:10s} M={M:6d} N={N:6d} K={K:6d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF/s")
print(f"llama layer GEMMs: {tot_t:.3f} ms  {tot_f / tot_t / 1e9:.1f} TF/s")
