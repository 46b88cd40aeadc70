#!/usr/bin/env python
"""Is the KV-cached decode step limited by the host (161 launches through ctypes per token) or by the GPU?  Times at batch 1, S = 643 prefix:
  (a) the eager step as generate() runs it: host enqueue time per step (no sync) and wall time per step (sync every 16 steps);
  (b) the same step captured ONCE in a HIP graph (fixed position: the launch arguments of a step are host scalars) and replayed: GPU-only time.
usage: python tools/decode_graph_probe.py"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = "cuda:0"
model, cfg = bench.build_model(336, dev)
images, ids, mask = bench.make_inputs(cfg, 1, 64, dev, 0)
M = importlib.import_module("u-llava_amd.modeling_core")
with torch.no_grad():
    out = model.forward(input_ids=ids, images=images, use_cache=True)
    cache = out.past_key_values
    tok = out.logits[:, -1].argmax(-1, keepdim=True)
    L0 = cache.length

    def step(t):
        o = model.forward(input_ids=t, past_key_values=cache, use_cache=True)
        return o.logits[:, -1].argmax(-1, keepdim=True)
    for _ in range(4):
        tok = step(tok)
    torch.cuda.synchronize()
    n = 32
    t0 = time.perf_counter()
    for _ in range(n):
        tok = step(tok)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"eager: host enqueue {t_enq / n * 1e3:.3f} ms per step, wall {t_all / n * 1e3:.3f} ms per step (cache length {L0} -> {cache.length})")
    # (b) one step in a graph, at a fixed cache position
    del cache.last_hidden[:]
    pos0 = cache.length
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(3):
            cache.length = pos0
            y = step(tok)
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cache.length = pos0
    with torch.cuda.graph(g):
        y = step(tok)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 50
    wbytes = sum(p.numel() * p.element_size() for n_, p in model.named_parameters() if "vision" not in n_)
    print(f"graph replay of one step at position {pos0}: {ms:.3f} ms per step = {wbytes / ms / 1e9:.2f} TB/s of weights")
