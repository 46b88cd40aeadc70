#!/usr/bin/env python
"""Where does the HOST time of one enqueue go?  cProfile of N un-synchronised C4 steps (and of KV-cached decode steps) on an idle GPU.
usage: python tools/host_profile.py [c4|decode]"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
what = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda:0")
with torch.no_grad():
    if what == "c4":
        step, batch, S, cfg, desc, fl, model = bench.workload_step("c4", dev, 0)
        n = 4
    else:
        model, cfg = bench.build_model(336, dev)
        images, ids, mask = bench.make_inputs(cfg, 1, 64, dev, 0)
        out = model.forward(input_ids=ids, images=images, use_cache=True)
        cache = out.past_key_values
        tok = [out.logits[:, -1].argmax(-1, keepdim=True)]

        def step():
            o = model.forward(input_ids=tok[0], past_key_values=cache, use_cache=True)
            tok[0] = o.logits[:, -1].argmax(-1, keepdim=True)
        n = 32
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{what}: host enqueue {(t1 - t0) / n * 1e3:.3f} ms per step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(25)
