"""RES step (batch 8): does a stream PRIORITY change the two-stream overlap?  SAM encoder stream high / normal, CLIP + LLaMA stream high / normal.
usage: python tools/probes/res_stream_priority.py"""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
bench = importlib.import_module("bench")
dev = torch.device("cuda:0")
step, batch, S, cfg, desc, fl, model = bench.workload_step("res", dev, 0)
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a", flush=True)


def timeit(n=8, w=2):
    with torch.no_grad():
        for _ in range(w):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for rep in range(2):
    for side_p, main_p in ((0, None), (-1, None), (0, -1), (-1, -1)):
        model._side = torch.cuda.Stream(priority=side_p)
        if main_p is None:
            t = timeit()
        else:
            ms = torch.cuda.Stream(priority=main_p)
            ms.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ms):
                t = timeit()
            torch.cuda.current_stream().wait_stream(ms)
        print(f"SAM stream priority {side_p}, LLM stream {'default' if main_p is None else main_p}: {t:.2f} ms per step", flush=True)
