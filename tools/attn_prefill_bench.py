#!/usr/bin/env python
"""The prefill attention of LLaMA (causal, hd 128) and CLIP (hd 64) exactly as the models call it: q | k | v consumed in place from the fused
projection buffer, V as rows (no V^T pass).  20 warm-up + 40 timed launches."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev, BF = "cuda:0", torch.bfloat16


def run(name, B, H, S, hd, causal):
    D = H * hd
    qkv = (torch.randn(B * S, 3 * D, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 0.5).to(BF)
    st = (S * 3 * D, hd, 3 * D)
    att = torch.empty(B * S, D, device=dev, dtype=BF)
    f = lambda: ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], att, B, H, S, S, hd, st, st, (S * D, hd, D), None, causal=causal, scale_mode=1,
                              scale=hd ** -0.5, v_strides=st)
    gf = 4 * B * H * S * S * hd / 1e9 * (0.5 if causal else 1.0)
    for _ in range(20):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        f()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    if os.environ.get("DUMP"):
        torch.save(att.cpu(), os.environ["DUMP"] + "_" + name.split()[0] + f"_S{S}.pt")
    if os.environ.get("CMP"):
        ref = torch.load(os.environ["CMP"] + "_" + name.split()[0] + f"_S{S}.pt")
        d = (att.cpu().float() - ref.float()).abs()
        print(f"    vs {os.environ['CMP']}: {float((att.cpu() != ref).float().mean()) * 100:.5f} % of elements differ, max |d| {float(d.max()):.3g} "
              f"(max |ref| {float(ref.float().abs().max()):.3g}), nan {int(torch.isnan(att.float()).sum())}")
    print(f"{name:28s}: {us:8.1f} us  {gf / us * 1e3:7.1f} TF/s (useful)  checksum {float(att.float().abs().mean()):.6f}  "
          f"sha {__import__('hashlib').sha256(att.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]}")


run("llama B=32 S=643 (C4)", 32, 32, 643, 128, True)
if os.environ.get("ONLY_C4"):
    sys.exit(0)
run("llama B=8 S=379 (RES)", 8, 32, 379, 128, True)
run("llama B=16 S=323 (train)", 16, 32, 323, 128, True)
run("clip B=32 S=577 (C4)", 32, 16, 577, 64, False)
run("clip B=32 S=257", 32, 16, 257, 64, False)
