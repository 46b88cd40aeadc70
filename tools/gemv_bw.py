#!/usr/bin/env python
"""Weight-stream bandwidth of the decode GEMV against plain streaming reads of the same bytes (torch reductions as the yardstick)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for N, K in ((4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008), (32011, 4096), (131072, 4096)):
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    x = torch.randn(1, K, device=dev).to(torch.bfloat16)
    nb = w.numel() * 2
    # rotate over several copies so that the 256 MB Infinity Cache does not serve the stream
    copies = [w] + [w.clone() for _ in range(max(1, min(8, int(2e9 // nb))))]
    i = [0]
    def gv():
        i[0] = (i[0] + 1) % len(copies)
        ops.linear(x, copies[i[0]])
    def rd():
        i[0] = (i[0] + 1) % len(copies)
        copies[i[0]].view(torch.int32).sum()
    t = timeit(gv); tr = timeit(rd)
    print(f"N={N:6d} K={K:5d} {nb / 1e6:7.1f} MB  gemv {t:7.1f} us {nb / t / 1e6:5.2f} TB/s   torch int32 sum {tr:7.1f} us {nb / tr / 1e6:5.2f} TB/s")
