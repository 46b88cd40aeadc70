#!/usr/bin/env python
"""Every large Linear of the C4 and RES workloads with its real epilogue, per kernel form (TUNE bits: 4 waves, 4 waves with the staged
epilogue, 8 waves, shipped policy), 20 warm-up + 40 timed launches per leg.  SHAPES=llama,clip,sam,res picks groups."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
W4, W8, STAGED = 1 << 22, 1 << 21, 512
FORMS = [("ship", 0), ("w4", W4), ("w4-staged", W4 | STAGED), ("w8", W8)]
if os.environ.get("FORMS"):
    FORMS = [f for f in FORMS if f[0] in os.environ["FORMS"].split(",")]
GROUPS = {
    "llama": [("qkv+rope", 20576, 12288, 4096, "rope"), ("o+res", 20576, 4096, 4096, "res"), ("gate_up+swiglu", 20576, 22016, 4096, "swiglu"),
              ("down+res", 20576, 4096, 11008, "res"), ("lm_head", 20576, 32011, 4096, "plain")],
    "clip": [("qkv", 18464, 3072, 1024, "bias"), ("out+res", 18464, 1024, 1024, "bias_res"), ("fc1+qgelu", 18464, 4096, 1024, "bias_qgelu"),
             ("fc2+res", 18464, 1024, 4096, "bias_res")],
    "sam": [("qkv", 32768, 3840, 1280, "bias"), ("proj+res", 32768, 1280, 1280, "bias_res"), ("fc1+gelu", 32768, 5120, 1280, "bias_gelu"),
            ("fc2+res", 32768, 1280, 5120, "bias_res")],
    "res": [("qkv+rope", 3032, 12288, 4096, "rope"), ("o+res", 3032, 4096, 4096, "res"), ("gate_up+swiglu", 3032, 22016, 4096, "swiglu"),
            ("down+res", 3032, 4096, 11008, "res")],
}


def timeit(fn, it=40, warm=20):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for grp in os.environ.get("SHAPES", "llama,clip,sam,res").split(","):
    tot = {f: 0.0 for f, _ in FORMS}
    for name, M, N, K, kind in GROUPS[grp]:
        x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
        ops.register_tiled(w)
        b = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
        n_out = N // 2 if kind == "swiglu" else N
        ldc = (n_out + 63) // 64 * 64 if name == "lm_head" and os.environ.get("PAD_LM") else n_out
        out = torch.empty(M, ldc, device=dev, dtype=torch.bfloat16)[:, :n_out]
        res = torch.randn(M, n_out, device=dev, generator=g).to(torch.bfloat16) if "res" in kind else None
        if kind == "rope":
            pos = (torch.arange(M, device=dev) % 643)
            inv = (1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))).to(dev)
            cs, sn = ops.rope_table(pos, inv, torch.bfloat16)
        line = f"{grp:5s} {name:15s} M={M:5d} N={N:5d} K={K:5d} "
        for fname, tune in FORMS:
            if kind == "rope":
                fn = lambda: ops.linear_qkv_rope(x, w, cs, sn, 8192, 128, out=out, tune=tune)
            elif kind == "swiglu":
                fn = lambda: ops.linear(x, w, swiglu=True, out=out, tune=tune)
            else:
                act = "quick_gelu" if "qgelu" in kind else ("gelu" if "gelu" in kind else None)
                fn = lambda: ops.linear(x, w, b if "bias" in kind else None, act=act, residual=res, out=out, tune=tune)
            t = timeit(fn)
            tot[fname] += t
            line += f" {fname} {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF/s |"
        print(line, flush=True)
    print(f"{grp:5s} total: " + "  ".join(f"{f} {v:8.1f} us" for f, v in tot.items()), flush=True)
