#!/usr/bin/env python
"""ViT patchify at batch 32 (27.6 MB @224, 60.6 MB @336 algorithmic bytes): im2col + MFMA GEMM."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
for size, alg_mb, gf in ((224, 27.6, 9.87), (336, 60.6, 22.2)):
    B, ps, D = 32, 14, 1024
    K = 3 * ps * ps
    Kp = ((K + 63) // 64) * 64
    img = torch.randn(B, 3, size, size, device=dev).to(torch.bfloat16)
    w = torch.zeros(D, Kp, device=dev, dtype=torch.bfloat16)
    w[:, :K] = torch.randn(D, K, device=dev).to(torch.bfloat16) * K ** -0.5
    wp = ops.pack_patch_weight(w[:, :K].reshape(D, 3, ps, ps).contiguous())
    fused = os.environ.get("UNFUSED") is None
    def run():
        return ops.patchify(img, wp, ps) if fused else ops.linear(ops.im2col(img, ps, Kp), w)
    ref = ops.linear(ops.im2col(img, ps, Kp), w).float()
    got = ops.patchify(img, wp, ps).float()
    print(f"   fused vs im2col+GEMM: max |d| = {float((got - ref).abs().max()):.4g} (max |ref| = {float(ref.abs().max()):.3g}), equal bits: {bool(torch.equal(got, ref))}")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"patchify {size}x{size} B=32: {us:7.1f} us  -> {alg_mb / us * 1e3 / 1e3:6.3f} TB/s algorithmic ({alg_mb / us * 1e3 / 8000 * 100:4.1f} % of 8 TB/s), {gf / us * 1e3:6.1f} TFLOP/s")
