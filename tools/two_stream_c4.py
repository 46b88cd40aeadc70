#!/usr/bin/env python
"""Experiment: the C4 batch (32 images) as ONE forward vs two half-batches on two HIP streams (the blocks of one half fill the
epilogue / hand-over gaps of the other, as the SAM encoder does for the LLaMA stream in the RES forward)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
NS = int(os.environ.get("NSTREAMS", 2))
with torch.no_grad():
    step, batch, S, cfg, desc, flops, model = bench.workload_step("c4", dev, 0)
    vis, ids, mask = bench.make_inputs(cfg, batch, 64, dev, 0)
    streams = [torch.cuda.Stream() for _ in range(NS)]
    parts = [(vis[i::NS].contiguous(), ids[i::NS].contiguous(), mask[i::NS].contiguous()) for i in range(NS)]

    def split_step():
        main = torch.cuda.current_stream()
        outs = []
        for st, (v, i, m) in zip(streams, parts):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(model.forward(input_ids=i, attention_mask=m, images=v).logits)
        for st in streams:
            main.wait_stream(st)
        return outs

    def t(fn, n=10, w=3):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    a = t(step)
    b = t(split_step)
    a2 = t(step)
    print(f"one forward of 32: {a:.1f} ms ({32 / a * 1e3:.1f} images/s); {NS} streams x {32 // NS}: {b:.1f} ms ({32 / b * 1e3:.1f} images/s); one forward again: {a2:.1f} ms")
    full = step().logits
    outs = split_step()
    ok = all(torch.equal(full[i::NS], o) for i, o in enumerate(outs))
    print("split results bit-identical to the single forward:", ok)
