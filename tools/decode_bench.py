#!/usr/bin/env python
"""KV-cached greedy decoding: prefill a 336x336 image + 64-token prompt (S = 643), then time the
token-by-token steps.  At batch <= 4 a step is one pass over the 13.5 GB of LLaMA-7B weights through the GEMV kernels, so the
roofline is HBM: 13.48 GB / 8 TB/s = 1.69 ms per step."""
import argparse, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--new", type=int, default=33)
a = ap.parse_args()
dev = "cuda:0"
model, cfg = bench.build_model(336, dev)
images, ids, mask = bench.make_inputs(cfg, a.batch, 64, dev, 0)
with torch.no_grad():
    for n in (2, a.new):            # warm-up (allocator, packed weights), then the timed run
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.generate(input_ids=ids, images=images, max_new_tokens=1, do_sample=False, use_cache=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out = model.generate(input_ids=ids, images=images, max_new_tokens=n, do_sample=False, use_cache=True)
        torch.cuda.synchronize(); t2 = time.perf_counter()
prefill = t1 - t0
per_tok = ((t2 - t1) - prefill) / (a.new - 1)
wbytes = sum(p.numel() * p.element_size() for n_, p in model.named_parameters() if "vision" not in n_)
print(f"batch {a.batch}: prefill+1 token {prefill * 1e3:.1f} ms; decode {per_tok * 1e3:.3f} ms/step = {a.batch / per_tok:.1f} tokens/s; "
      f"weights {wbytes / 1e9:.2f} GB -> {wbytes / per_tok / 1e12:.2f} TB/s ({wbytes / per_tok / 8e12 * 100:.1f} % of 8 TB/s)")
