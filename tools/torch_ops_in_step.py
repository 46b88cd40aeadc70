"""Which torch-native (non-library) device kernels run inside one forward step, and from which line of the package: one profiled step of
a bench workload, aten ops with device time grouped by the innermost u-llava_amd frame.
usage: python tools/torch_ops_in_step.py [c4|res|c1]"""
import importlib, os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda:0")
step = bench.workload_step(name, dev, 0)[0]
with torch.no_grad():
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if not t or not ev.name.startswith("aten::"):
        continue
    site = "?"
    for fr in ev.stack or []:
        if "u-llava_amd" in fr or "bench.py" in fr:
            site = fr.split("u-llava_amd/")[-1] if "u-llava_amd" in fr else fr
            break
    k = (ev.name, site[:110])
    by[k][0] += 1
    by[k][1] += t
tot = sum(v[1] for v in by.values())
print(f"{name}: torch-native device time in one step: {tot / 1e3:.2f} ms")
for (n, site), (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / 1e3:8.3f} ms  x{c:<4d} {n:28s} {site}")
