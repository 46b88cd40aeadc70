#!/usr/bin/env python
"""Tile-major operand layout experiment: W (and X) stored [rows/256][K/64][256][64] so that every 256x64 K-tile is 32 contiguous KiB."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
_lib = importlib.import_module("u-llava_amd._lib")
dev = "cuda:0"
T = 20576
W_TILED, X_TILED, SWIGLU = 64, 128, 16


def tile(a):
    R, K = a.shape
    Rp = (R + 255) // 256 * 256
    if Rp != R:
        a = torch.cat([a, torch.zeros(Rp - R, K, device=a.device, dtype=a.dtype)])
    return a.view(Rp // 256, 256, K // 64, 64).permute(0, 2, 1, 3).contiguous()


def run(x, w, out, flags, M, N, K):
    _lib.call("ull_gemm_bf16", x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), out.stride(0), None, None, 0, M, N, K, flags, None, 0,
              torch.cuda.current_stream().cuda_stream)


g = torch.Generator(device=dev).manual_seed(0)
tot = {}
for name, M, N, K, sw in [("qkv", T, 12288, 4096, False), ("o", T, 4096, 4096, False), ("gateup", T, 22016, 4096, True),
                          ("down", T, 4096, 11008, False), ("sq8192", 8192, 8192, 8192, False)]:
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    xt, wt = tile(x), tile(w)
    outs = []
    for label, xa, wa, fl in (("plain", x, w, 0), ("W tiled", x, wt, W_TILED), ("X tiled", xt, w, X_TILED), ("W+X tiled", xt, wt, W_TILED | X_TILED)):
        out = torch.empty(M, N // 2 if sw else N, device=dev, dtype=torch.bfloat16)
        flags = fl | (SWIGLU if sw else 0)
        for _ in range(2):
            run(xa, wa, out, flags, M, N, K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(xa, wa, out, flags, M, N, K)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        outs.append(out)
        if name != "sq8192":
            tot[label] = tot.get(label, 0) + ms
        print(f"{name:8s} {label:10s} {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF/s  equal_to_plain={bool(torch.equal(out, outs[0]))}")
fl = sum(2.0 * T * n * k for n, k in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)))
for k, v in tot.items():
    print(f"llama layer {k:10s}: {v:.3f} ms {fl / v / 1e9:.1f} TF/s")
