#!/usr/bin/env python
"""SAM prompt-encoder + two-way mask decoder alone: fused kernels (csrc/sam_decoder.hip) vs the op-by-op chain, n = 3 and n = 24 prompts
over 8 images (random weights, bf16).  usage: decoder_bench.py [fused|unfused|both] [iters]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C, S = importlib.import_module("u-llava_amd.configuration"), importlib.import_module("u-llava_amd.sam")
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
cfg = C.SamConfig(depth=0)
holder = S.build_sam_holder(cfg, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
for n_, p in holder.named_parameters():
    p.data.normal_(0.0, 0.05, generator=g)
    if "norm" in n_ and n_.endswith("weight"):
        p.data.fill_(1.0)
holder.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix.normal_(0.0, 1.0, generator=g)
eng = S.SamEngine(holder, cfg)
emb = torch.randn(8, 4096, 256, device=dev, generator=g).to(torch.bfloat16)
for n in (3, 24):
    text = torch.randn(n, 256, device=dev, generator=g).to(torch.bfloat16)
    idx = (torch.arange(n, device=dev) * 8 // n).to(torch.int64)
    for fused in ([True, False] if mode == "both" else [mode == "fused"]):
        eng.fused_decoder = fused
        for _ in range(3):
            eng.decode(emb, text, idx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            masks, iou = eng.decode(emb, text, idx)
        e1.record(); e1.synchronize()
        wall = (time.perf_counter() - t0) / iters * 1e3
        print(f"n={n:2d} {'fused' if fused else 'op-by-op':8s}: {e0.elapsed_time(e1) / iters:.3f} ms GPU per decode ({wall:.3f} ms wall)", flush=True)
