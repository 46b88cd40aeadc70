"""Decode-shape Linear at M = 1..16 on the LLaMA-7B layer shapes: GEMV (M <= 4) / tiled GEMM (M > 4) against the skinny MFMA kernel.
usage: python tools/skinny_bench.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"


def timeit(fn, it=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


shapes = (("qkv", 12288, 4096, False), ("o", 4096, 4096, False), ("gate/up", 22016, 4096, True), ("down", 4096, 11008, False))
ws = {}
for name, N, K, sw in shapes:
    ws[name] = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(6)]     # rotate: defeat the Infinity Cache
for M in (1, 2, 3, 4, 8, 16):
    row = f"M={M:2d} "
    tot_a = tot_b = 0.0
    for name, N, K, sw in shapes:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        i = [0]
        def run(tune):
            i[0] = (i[0] + 1) % 6
            ops.linear(x, ws[name][i[0]], swiglu=sw, tune=tune)
        ta = timeit(lambda: run(1 << 16))          # any tune bit: the previous dispatch (GEMV for M <= 4, tiled GEMM above)
        tb = timeit(lambda: run(0))
        tot_a += ta; tot_b += tb
        row += f" {name}: {ta:6.1f} -> {tb:6.1f} us"
    print(row + f"   layer {tot_a:6.1f} -> {tot_b:6.1f} us")
