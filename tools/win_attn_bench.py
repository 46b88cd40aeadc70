"""SAM window attention at the RES shape (8 images of 64 x 64 tokens, 16 heads, hd 80): the generic chain (window_partition -> V^T pass ->
register attention with fused rel-pos -> window_unpartition) against ull_sam_window_attention on image-order tokens.
usage: python tools/win_attn_bench.py"""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda"
B, g, nH, side, hd = 8, 64, 16, 14, 80
S, C = side * side, nH * hd
gen = torch.Generator().manual_seed(0)
qkv = torch.randn(B * g * g, 3 * C, generator=gen).to(torch.bfloat16).to(dev)
bias = torch.randn(3 * C, generator=gen).to(torch.bfloat16).to(dev)
rph = (0.3 * torch.randn(2 * side - 1, hd, generator=gen)).to(torch.bfloat16).to(dev)
rpw = (0.3 * torch.randn(2 * side - 1, hd, generator=gen)).to(torch.bfloat16).to(dev)


def timeit(fn, it=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


# generic chain on window-major rows (the padded positions' q|k|v = the bias, as Linear gives them)
zero = torch.zeros(B * g * g, 3 * C, device=dev, dtype=torch.bfloat16)
qkv_w = ops.window_partition(qkv - bias, B, g, g, side) + bias     # stand-in for the window-major qkv (same values on real tokens)
NB = qkv_w.shape[0] // S
strides = (S * 3 * C, hd, 3 * C)
att = torch.empty(NB * S, C, device=dev, dtype=torch.bfloat16)


def generic():
    vt = ops.transpose_v(qkv_w[:, 2 * C:], S * 3 * C, 3 * C, NB, S, nH, hd)
    ops.attention(qkv_w, qkv_w[:, C:], vt, att, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False, scale_mode=0,
                  q_scale=hd ** -0.5, rel_h=rph, rel_w=rpw, rel_pos_hw=(side, side))


for rep in range(2):
    t0 = timeit(generic)
    t1 = timeit(lambda: ops.sam_window_attention(qkv, bias, rph, rpw, B, g, g, nH, hd, side))
    print(f"V^T pass + generic window attention on 4900 positions {t0:.1f} us   image-order kernel on 4096 tokens {t1:.1f} us")
