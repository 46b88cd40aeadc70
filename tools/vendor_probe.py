"""Yardstick only (never used by the product): which kernels does the vendor BLAS of this image pick for the LLaMA-layer shapes,
and with what launch geometry?  Run under `rocprofv3 --kernel-trace --output-format csv`; tools/vendor_probe.sh prints the table."""
import torch, torch.nn.functional as F
M = 20576
for n, k in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (8192, 8192)):
    x = torch.randn(M if n != 8192 else 8192, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.02
    for _ in range(3):
        y = F.linear(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = F.linear(x, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"vendor N={n} K={k}: {ms:.3f} ms  {2 * x.shape[0] * n * k / ms / 1e9:.0f} TF/s")
