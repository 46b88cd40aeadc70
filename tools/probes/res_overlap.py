import importlib, sys, os, torch
sys.path.insert(0, "/root/repo")
bench = importlib.import_module("bench")
dev = torch.device("cuda:0")
step, batch, S, cfg, desc, fl, model = bench.workload_step("res", dev, 0)
def timeit(n=6, w=2):
    with torch.no_grad():
        for _ in range(w): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): step()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(3):
    for ov in (True, False):
        model.overlap_sam_encoder = ov
        model._side = None
        print(f"overlap_sam_encoder={ov}: {timeit():.2f} ms per step", flush=True)
