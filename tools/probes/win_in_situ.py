"""Why is the SAM window attention ~1.5x slower inside a RES step than in a loop of its own?  Times the kernel (HIP events around it)
  a) back to back on the same q|k|v buffer (the standalone figure),
  b) right after the qkv GEMM that produces its input (M 32768, N 3840, K 1280: what precedes it in a SAM block),
  c) after a 1 GiB device copy (cold L2 / Infinity Cache, no matrix-core power draw),
  d) after the GEMM AND a 300 us idle gap on the host side.
usage: python tools/probes/win_in_situ.py"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda"
B, g, nH, side, hd = 8, 64, 16, 14, 80
C = nH * hd
gen = torch.Generator().manual_seed(0)
x = (0.5 * torch.randn(B * g * g, C, generator=gen)).to(torch.bfloat16).to(dev)
w = (0.03 * torch.randn(3 * C, C, generator=gen)).to(torch.bfloat16).to(dev)
bias = torch.randn(3 * C, generator=gen).to(torch.bfloat16).to(dev)
rph = (0.3 * torch.randn(2 * side - 1, hd, generator=gen)).to(torch.bfloat16).to(dev)
rpw = (0.3 * torch.randn(2 * side - 1, hd, generator=gen)).to(torch.bfloat16).to(dev)
big_a = torch.empty(1 << 29, dtype=torch.bfloat16, device=dev)
big_b = torch.empty(1 << 29, dtype=torch.bfloat16, device=dev)
qkv = ops.linear(x, w, bias)


def attn():
    return ops.sam_window_attention(qkv, bias, rph, rpw, B, g, g, nH, hd, side)


def timed(pre, n=30):
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for i in range(n):
        pre()
        e0[i].record(); attn(); e1[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in zip(e0, e1))
    return ts[len(ts) // 2], ts[0], ts[-1]


def gemm():
    global qkv
    qkv = ops.linear(x, w, bias)


def gemm_gap():
    gemm(); torch.cuda.synchronize(); time.sleep(300e-6)


for _ in range(5):
    attn()
for name, pre in (("a) back to back", lambda: None), ("b) after the qkv GEMM", gemm), ("c) after a 1 GiB copy", lambda: big_b.copy_(big_a)),
                  ("d) after GEMM + idle gap", gemm_gap), ("a) back to back", lambda: None)):
    med, lo, hi = timed(pre)
    print(f"{name:28s}: median {med:7.1f} us   min {lo:7.1f}   max {hi:7.1f}", flush=True)
