#!/usr/bin/env python
"""Single-image prefill shapes (M = 291 .. 643 tokens): what the LLaMA-layer Linears cost below the 256x256 kernels' range (M < 1024:
gemm128_kernel) against the weight stream (bytes / 5 TB/s) and the MFMA work (1.5 PF/s)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev, BF = "cuda:0", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def t(fn, n=30):
    for _ in range(8):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, split in ((291, False), (291, True), (323, False), (323, True), (643, False), (643, True), (1000, True)):
  with ops.small_m_split_k(split):
      print(f"--- M = {M}, split-K latency mode {'on' if split else 'off'}")
      tot = 0.0
      for name, N, K, sw in (("qkv", 12288, 4096, False), ("o", 4096, 4096, False), ("gateup", 22016, 4096, True), ("down", 4096, 11008, False)):
          x = torch.randn(M, K, device=dev, generator=g).to(BF)
          # eight weight copies cycled so that the launch streams its weights from HBM like a real layer does (one copy would sit in the MALL)
          ws = [(torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(BF) for _ in range(8)]
          out = torch.empty(M, N // 2 if sw else N, device=dev, dtype=BF)
          i = [0]

          def f():
              i[0] = (i[0] + 1) % 8
              ops.linear(x, ws[i[0]], swiglu=sw, out=out)
          us = t(f)
          tot += us
          tiles = ((M + 127) // 128) * ((N + 127) // 128)
          print(f"M={M:5d} {name:7s} 128-tiles={tiles:5d}  {us:7.1f} us  ({2.0 * M * N * K / us / 1e6:6.0f} TF/s; weights at 5 TB/s {N * K * 2 / 5e6:6.1f} us, "
                f"flops at 1.5 PF/s {2.0 * M * N * K / 1.5e9:6.1f} us)")
      print(f"M={M:5d} layer total {tot:7.1f} us")
