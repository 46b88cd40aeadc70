#!/usr/bin/env python
"""GPU-box tool: measures the two ceilings the roofline fractions are read against on THIS chip and writes them as JSON
("re-measure on the box and store them in roofline.json"; copied to <repo>/roofline.json):

  * sustained bf16 MFMA rate under the socket power cap: tools/probes/mfma_power.hip (register-resident random operands, ~4 s per shape);
  * HBM streaming bandwidth: a 2-GiB device copy (read + write bytes) and a 2-GiB read-only reduction, torch kernels as the yardstick
    (measurement of the memory system, not product code).

    python tools/roofline_probe.py > gpurun_out/roofline.json
"""
import json
import os
import re
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timeit(fn, it=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


def main():
    out = {"device": torch.cuda.get_device_name(0), "datasheet": {"bf16_mfma_dense_tflops": 2500.0, "hbm_GBps": 8000.0}}
    exe = "/tmp/mfma_power"
    src = os.path.join(ROOT, "tools", "probes", "mfma_power.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, src], capture_output=True, text=True, timeout=600)
    if r.returncode == 0:
        p = subprocess.run([exe, "4"], capture_output=True, text=True, timeout=300)
        rates = {}
        for m in re.finditer(r"mfma (\d+)x\d+: last-10-launch rate (\d+) TF/s, mean (\d+) TF/s", p.stdout):
            rates.setdefault(m.group(1), []).append(float(m.group(3)))
        out["mfma_sustained_tflops"] = {("16x16x32" if k == "16" else "32x32x16"): max(v) for k, v in rates.items()}
        out["mfma_probe_stdout"] = p.stdout.strip().splitlines()
    else:
        out["mfma_probe_error"] = r.stderr[-500:]
    n = 1 << 30                                             # 2 GiB of bf16: far beyond the 256-MiB Infinity Cache
    a = torch.empty(n, device="cuda", dtype=torch.bfloat16).normal_()
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    out["hbm_copy_GBps"] = round(2 * a.numel() * 2 / t / 1e9, 1)
    t = timeit(lambda: a.view(torch.int32).sum())
    out["hbm_read_GBps"] = round(a.numel() * 2 / t / 1e9, 1)
    t = timeit(lambda: b.zero_())
    out["hbm_write_GBps"] = round(a.numel() * 2 / t / 1e9, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
