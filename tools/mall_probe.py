#!/usr/bin/env python
"""Infinity Cache (MALL) probe: read bandwidth of repeated passes over a buffer as a function of its size.  rocprofv3 on gfx950 / ROCm 7.2
exposes no counter behind the Infinity Cache (no DF / UMC block in `rocprofv3 -L`: profiles/r03_counter_list_memory.txt), so HBM-side
bytes cannot be read directly; this shows how much of a re-read working set the 256 MiB cache actually serves.  Tool only (uses torch's
reduction as the streaming reader)."""
import torch

dev = "cuda:0"
for mb in (32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    x = torch.ones(n, device=dev, dtype=torch.float32)
    for _ in range(3):
        x.sum()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = max(5, 4096 // mb)
    e0.record()
    for _ in range(it):
        x.sum()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / it * 1e-3
    print(f"re-read of {mb:5d} MiB: {mb * (1 << 20) / t / 1e12:6.2f} TB/s")
    del x
