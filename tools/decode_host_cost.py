#!/usr/bin/env python
"""Host time to ENQUEUE one KV-cached decode step on an IDLE GPU (synchronize before every step, so that no queue back-pressure is counted):
coarse layer-stack entry vs one ctypes call per launch.  Batch 1, prefix S = 643.  usage: python tools/decode_host_cost.py"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ops = importlib.import_module("u-llava_amd.ops")
dev = torch.device("cuda:0")
with torch.no_grad():
    model, cfg = bench.build_model(336, dev)
    images, ids, mask = bench.make_inputs(cfg, 1, 64, dev, 0)
    out = model.forward(input_ids=ids, images=images, use_cache=True)
    cache = out.past_key_values
    tok = out.logits[:, -1].argmax(-1, keepdim=True)

    def step(t):
        o = model.forward(input_ids=t, past_key_values=cache, use_cache=True)
        return o.logits[:, -1].argmax(-1, keepdim=True)
    for _ in range(4):
        tok = step(tok)
    for label, ctx in (("coarse", ops.per_op_layers(False)), ("per_op", ops.per_op_layers(True)), ("coarse", ops.per_op_layers(False))):
        with ctx:
            hs, ws = [], []
            for _ in range(24):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                tok = step(tok)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                hs.append(t1 - t0)
                ws.append(t2 - t0)
            hs.sort(); ws.sort()
            print(f"{label}: host enqueue median {hs[12] * 1e3:.3f} ms (min {hs[0] * 1e3:.3f}), enqueue + drain median {ws[12] * 1e3:.3f} ms per step", flush=True)
