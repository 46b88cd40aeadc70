#!/usr/bin/env python
"""Does the token stride of the fused q|k|v buffer matter to the prefill attention?  3 x 4096 x 2 B = 24576 B puts the 64 key rows of a tile at
addresses that differ only above bit 13: if the L2 picks its channel from lower address bits, one tile is one channel.  Same data, same kernel,
row pitch 24576 B vs padded pitches."""
import importlib, os, sys, hashlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev, BF = "cuda:0", torch.bfloat16
B, H, S, hd = 32, 32, 643, 128
D = H * hd
base = (torch.randn(B * S, 3 * D, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 0.5).to(BF)
for pad in (0, 64, 128, 256, 512, 1024, 2048):
    ld = 3 * D + pad
    buf = torch.zeros(B * S, ld, device=dev, dtype=BF)
    buf[:, :3 * D] = base
    qkv = buf[:, :3 * D]
    st = (S * ld, hd, ld)
    att = torch.empty(B * S, D, device=dev, dtype=BF)
    f = lambda: ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], att, B, H, S, S, hd, st, st, (S * D, hd, D), None, causal=True, scale_mode=1,
                              scale=hd ** -0.5, v_strides=st)
    for _ in range(20):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    print(f"row pitch {ld * 2:6d} B (pad {pad * 2:5d} B): {us:7.1f} us  sha {hashlib.sha256(att.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]}")
