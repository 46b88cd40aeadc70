"""Yardstick only: the vendor BLAS on one GEMM shape (for rocprofv3 --pmc passes).  usage: vendor_one.py M N K"""
import sys, torch, torch.nn.functional as F
M, N, K = (int(v) for v in sys.argv[1:4])
g = torch.Generator(device="cuda:0").manual_seed(0)
x = torch.randn(M, K, device="cuda:0", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda:0", generator=g) * K ** -0.5).to(torch.bfloat16)
for _ in range(3):
    y = F.linear(x, w)
torch.cuda.synchronize()
