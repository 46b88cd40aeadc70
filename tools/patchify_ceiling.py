#!/usr/bin/env python
"""where is the ceiling of the fused ViT patchify?  Times, at B = 32 and 224^2 / 336^2 (CLIP, ps 14, N 1024) and
B = 8 1024^2 (SAM, ps 16, N 1280):
  * the shipped fused kernel (pixels -> LDS by DMA, no im2col buffer);
  * a plain GEMM of the SAME M x N x K' on row-major operands that already sit in HBM (no gather at all), on each of the product's
    three GEMM kernels -- the rate this chip's MFMA path reaches at this K and tile count, i.e. the ceiling for any patchify that is
    a GEMM;
  * (yardstick, not used by the product) the vendor BLAS on the same plain GEMM;
  * a device copy of the op's algorithmic bytes (pixels + weight + output) -- the HBM roofline as this box delivers it."""
import importlib, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
BF = torch.bfloat16


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


for name, n, HW, ps, N in (("CLIP 224", 32, 224, 14, 1024), ("CLIP 336", 32, 336, 14, 1024), ("SAM 1024", 8, 1024, 16, 1280)):
    img = torch.randn(n, 3, HW, HW, device=dev).to(BF)
    K = 3 * ps * ps
    Kp = (3 * ps * 16 + 63) // 64 * 64
    wp = (torch.randn(N, Kp, device=dev) * 0.02).to(BF)
    M = n * (HW // ps) ** 2
    alg = img.numel() * 2 + N * K * 2 + M * N * 2
    t_fused = timeit(lambda: ops.patchify(img, wp, ps))
    x = torch.randn(M, Kp, device=dev).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    t_plain = {}
    for label, tune in (("shipped dispatch", 0), ("128x128", 1 << 20), ("256x256 / 8 waves", 1 << 21), ("256x256 / 4 waves", 1 << 22)):
        t_plain[label] = timeit(lambda: ops.linear(x, wp, out=out, tune=tune))
    t_vendor = timeit(lambda: F.linear(x, wp))
    src = torch.empty(alg // 2, device=dev, dtype=torch.uint8); dst = torch.empty_like(src)
    t_copy = timeit(lambda: dst.copy_(src))
    best = min(t_plain.values())
    print(f"{name}: M={M} N={N} K={K} (K'={Kp})  algorithmic {alg / 1e6:.1f} MB, {2.0 * M * N * K / 1e9:.1f} GF")
    print(f"   fused patchify            {t_fused:6.1f} us = {alg / t_fused / 1e6:.2f} TB/s ({alg / t_fused / 8e6 * 100:.1f} % of 8 TB/s), {2.0 * M * N * K / t_fused / 1e6:.0f} TF/s")
    for k, v in t_plain.items():
        print(f"   plain GEMM, {k:18s} {v:6.1f} us = {2.0 * M * N * Kp / v / 1e6:.0f} TF/s on K'")
    print(f"   plain GEMM, vendor BLAS    {t_vendor:6.1f} us")
    print(f"   copy of the same bytes     {t_copy:6.1f} us = {alg / t_copy / 1e6:.2f} TB/s (read + write counted once each)")
    print(f"   fused / best plain GEMM of the product = {t_fused / best:.2f}")
