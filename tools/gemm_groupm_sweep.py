#!/usr/bin/env python
"""Per-shape sweep of the 256x256 kernel's tile raster (ULL_GEMM_TUNE_GROUP_M) and of the stream-K tail, interleaved rounds in one
process (median of 5 rounds x 6 launches), tile-major weights like the model.  Random operands."""
import importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
T = 20576
shapes = [("qkv", T, 12288, 4096, False), ("o", T, 4096, 4096, False), ("gateup", T, 22016, 4096, True), ("down", T, 4096, 11008, False),
          ("lm_head", T, 32011, 4096, False), ("clip_qkv", 32 * 577, 3072, 1024, False), ("clip_fc1", 32 * 577, 4096, 1024, False),
          ("clip_fc2", 32 * 577, 1024, 4096, False), ("sam_qkv", 8 * 4900, 3840, 1280, False), ("sam_lin1", 8 * 4096, 5120, 1280, False),
          ("sam_lin2", 8 * 4096, 1280, 5120, False)]
g = torch.Generator(device=dev).manual_seed(0)
GMS = [0, 2, 3, 5, 6, 8, 12]
for name, M, N, K, sw in shapes:
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    ops.register_tiled(w)
    out = torch.empty(M, N // 2 if sw else N, device=dev, dtype=torch.bfloat16)
    variants = [(f"gm{g_}", dict(tune=g_ << 16), True) for g_ in GMS] + [("gm4-nosk", dict(tune=0), False)]
    res = {v[0]: [] for v in variants}
    for rnd in range(5):
        for vname, kw, sk in variants:
            pol = ops.streamk_policy(2048 if sk else None)
            with pol:
                ops.linear(x, w, swiglu=sw, out=out, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    ops.linear(x, w, swiglu=sw, out=out, **kw)
                e1.record(); e1.synchronize()
            res[vname].append(e0.elapsed_time(e1) / 6)
    fl = 2.0 * M * N * K
    line = "  ".join(f"{k}:{fl / statistics.median(v) / 1e9:7.1f}" for k, v in res.items())
    print(f"{name:9s} M={M} N={N} K={K}  TF/s  {line}", flush=True)
