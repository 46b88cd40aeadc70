"""LLaMA / CLIP prefill attention: V^T pass + attention on the V^T image against attention on V rows (transposing LDS reads).
usage: python tools/vrows_bench.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev, BF = "cuda:0", torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, B, H, S, hd, causal in (("llama S=643", 32, 32, 643, 128, True), ("clip S=577", 32, 16, 577, 64, False), ("clip S=257", 32, 16, 257, 64, False),
                                  ("llama S=379 B=8", 8, 32, 379, 128, True)):
    D = H * hd
    qkv = (torch.randn(B * S, 3 * D, device=dev) * 0.5).to(BF)
    st = (S * 3 * D, hd, 3 * D)
    out = torch.empty(B * S, D, device=dev, dtype=BF)
    vt = ops.transpose_v(qkv[:, 2 * D:], S * 3 * D, 3 * D, B, S, H, hd)
    kw = dict(causal=causal, scale_mode=1, scale=hd ** -0.5)
    for rep in range(2):
        t_tr = timeit(lambda: ops.transpose_v(qkv[:, 2 * D:], S * 3 * D, 3 * D, B, S, H, hd, out=vt))
        t_a = timeit(lambda: ops.attention(qkv, qkv[:, D:], vt, out, B, H, S, S, hd, st, st, (S * D, hd, D), None, **kw))
        t_r = timeit(lambda: ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], out, B, H, S, S, hd, st, st, (S * D, hd, D), None, v_strides=st, **kw))
        print(f"{name:18s} V^T pass {t_tr:6.1f} us + attention {t_a:6.1f} us = {t_tr + t_a:6.1f}    on V rows {t_r:6.1f} us")
