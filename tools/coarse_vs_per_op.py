#!/usr/bin/env python
"""GPU time of one step through the coarse layer-stack entries vs through one ctypes call per launch (same kernels, same box, interleaved):
usage: python tools/coarse_vs_per_op.py [c4|c2|c5|res]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ops = importlib.import_module("u-llava_amd.ops")
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda:0")
with torch.no_grad():
    step, batch, S, cfg, desc, fl, model = bench.workload_step(name, dev, 0)
    for _ in range(3):
        step()
    for rep in range(3):
        for label, ctx in (("coarse", ops.per_op_layers(False)), ("per_op", ops.per_op_layers(True))):
            with ctx:
                step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    step()
                e1.record()
                torch.cuda.synchronize()
            print(f"{name} {label}: {e0.elapsed_time(e1) / 8:.2f} ms per step ({batch * 8 / e0.elapsed_time(e1) * 1e3:.1f} images/s)", flush=True)
