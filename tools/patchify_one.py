#!/usr/bin/env python
"""One shape of the fused ViT patchify, a few dozen launches: the profiling target of `rocprofv3 --pmc` passes (tools/prof_r04.sh)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 336
g = torch.Generator(device="cuda").manual_seed(9)
img = torch.randn(32, 3, size, size, device="cuda", generator=g).to(torch.bfloat16)
wp = ops.pack_patch_weight((torch.randn(1024, 3, 14, 14, device="cuda", generator=g) * 0.02).to(torch.bfloat16))
for _ in range(30):
    out = ops.patchify(img, wp, 14)
torch.cuda.synchronize()
print(float(out.float().abs().mean()))
