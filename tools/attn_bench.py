#!/usr/bin/env python
"""Attention kernel timings at the shapes of the u-LLaVA path (LLaMA prefill, CLIP, SAM windowed / global, mask decoder)."""
import importlib, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev, BF = "cuda:0", torch.bfloat16


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def sam(side, NB, nH=16, hd=80, mode="fused"):
    S, C = side * side, nH * hd
    qkv = (torch.randn(NB * S, 3 * C, device=dev) * 0.5).to(BF)
    rph = (torch.randn(2 * side - 1, hd, device=dev) * 0.3).to(BF)
    rpw = (torch.randn(2 * side - 1, hd, device=dev) * 0.3).to(BF)
    strides = (S * 3 * C, hd, 3 * C)
    vt = ops.transpose_v(qkv[:, 2 * C:], S * 3 * C, 3 * C, NB, S, nH, hd)
    att = torch.empty(NB * S, C, device=dev, dtype=BF)
    if mode == "fused":
        f = lambda: ops.attention(qkv, qkv[:, C:], vt, att, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False,
                                  scale_mode=0, q_scale=hd ** -0.5, rel_h=rph, rel_w=rpw, rel_pos_hw=(side, side))
    elif mode == "tables":
        rel_h, rel_w = ops.sam_relpos(qkv, strides, rph, rpw, NB, nH, side, side, hd)
        f = lambda: ops.attention(qkv, qkv[:, C:], vt, att, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False,
                                  scale_mode=0, q_scale=hd ** -0.5, rel_h=rel_h, rel_w=rel_w)
    else:
        f = lambda: ops.attention(qkv, qkv[:, C:], vt, att, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False,
                                  scale_mode=0, q_scale=hd ** -0.5)
    us = timeit(f)
    gf = 4 * NB * nH * S * S * hd / 1e9
    print(f"sam side={side:3d} NB={NB:4d} {mode:7s}: {us:8.1f} us  {gf / us * 1e3:7.1f} TF/s")


def plain(name, B, H, S, hd, causal, scale_mode=1):
    D = H * hd
    qkv = (torch.randn(B * S, 3 * D, device=dev) * 0.5).to(BF)
    strides = (S * 3 * D, hd, 3 * D)
    vt = ops.transpose_v(qkv[:, 2 * D:], S * 3 * D, 3 * D, B, S, H, hd)
    att = torch.empty(B * S, D, device=dev, dtype=BF)
    mask = torch.ones(B, S, device=dev, dtype=torch.int32) if causal else None
    f = lambda: ops.attention(qkv, qkv[:, D:], vt, att, B, H, S, S, hd, strides, strides, (S * D, hd, D), mask, causal=causal,
                              scale_mode=scale_mode, scale=hd ** -0.5)
    us = timeit(f)
    gf = 4 * B * H * S * S * hd / 1e9 * (0.5 if causal else 1.0)
    print(f"{name:24s}: {us:8.1f} us  {gf / us * 1e3:7.1f} TF/s (useful)")


if __name__ == "__main__":
    for mode in ("fused", "tables", "nobias"):
        sam(14, 200, mode=mode)
    for mode in ("tables", "fused"):
        sam(64, 8, mode=mode)
    plain("llama S=643 B=32", 32, 32, 643, 128, True)
    plain("clip S=577 B=32", 32, 16, 577, 64, False)
    plain("clip S=257 B=32", 32, 16, 257, 64, False)


def decode(B, H, Sk, hd, masked):
    D = H * hd
    smax = ((Sk + 63) // 64) * 64 + 64
    q = (torch.randn(B, 3 * D, device=dev) * 0.5).to(BF)
    kc = (torch.randn(B, H, smax, hd, device=dev) * 0.5).to(BF)
    vtc = (torch.randn(B, H, hd, smax, device=dev) * 0.5).to(BF)
    att = torch.empty(B, D, device=dev, dtype=BF)
    km = torch.ones(B, Sk, device=dev, dtype=torch.int32) if masked else None
    f = lambda: ops.attention(q, kc, vtc, att, B, H, 1, Sk, hd, (3 * D, hd, 3 * D), (H * smax * hd, smax * hd, hd), (D, hd, D), km,
                              causal=True, scale_mode=1, scale=hd ** -0.5)
    us = timeit(f, 50)
    print(f"decode B={B} Sk={Sk} mask={masked}: {us:7.1f} us  (K+V {2 * B * H * Sk * hd * 2 / 1e6:.1f} MB -> {2 * B * H * Sk * hd * 2 / us / 1e6:.2f} TB/s)")


if __name__ == "__main__" and os.environ.get("DECODE"):
    for masked in (True, False):
        decode(1, 32, 676, 128, masked)
        decode(4, 32, 676, 128, masked)
