"""The LLaMA-layer GEMM launches in bf16 and fp16 (same shapes, tile-major weights, real epilogues), 20 warm-up + 40 timed launches each.
usage: python tools/gemm_dtype_ab.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
M = 20576
for dt in (torch.bfloat16, torch.float16, torch.bfloat16, torch.float16):
    g = torch.Generator(device=dev).manual_seed(0)
    line = []
    for name, N, K, sw, res in (("qkv", 12288, 4096, False, False), ("o+res", 4096, 4096, False, True), ("gate_up+swiglu", 22016, 4096, True, False),
                                ("down+res", 4096, 11008, False, True)):
        x = torch.randn(M, K, device=dev, generator=g).to(dt)
        w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(dt)
        if sw:
            w = ops.pack_swiglu(w[: N // 2], w[N // 2:]) if hasattr(ops, "pack_swiglu") else w
        ops.register_tiled(w)
        out = torch.empty(M, N // 2 if sw else N, device=dev, dtype=dt)
        r = torch.randn(M, N, device=dev, generator=g).to(dt) if res else None
        fn = lambda: ops.linear(x, w, swiglu=sw, residual=r, out=out)
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        line.append(f"{name} {us:7.1f} us {2 * M * N * K / us / 1e6:6.0f} TF/s")
        del x, w, out, r
    print(f"{str(dt)[6:]:9s}: " + " | ".join(line), flush=True)
