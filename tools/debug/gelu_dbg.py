import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
g = torch.Generator().manual_seed(3)
x = (torch.randn(160, 1024, generator=g) * 2.5).to(torch.bfloat16).to(dev)
eye = torch.eye(1024, dtype=torch.bfloat16, device=dev)
a = ops.linear(x, eye, act="gelu")
b = ops.gelu_fwd(x)
c = torch.nn.functional.gelu(x.cpu()).to(dev)
idt = ops.linear(x, eye)
print("linear(x, I) == x:", bool(torch.equal(idt, x)))
print("gemm-epilogue vs gelu_fwd differ:", float((a != b).float().mean()), " gelu_fwd vs torch:", float((b != c).float().mean()), " gemm vs torch:", float((a != c).float().mean()))
bad = (a != b).nonzero()[:8]
for i, j in bad.tolist():
    print(float(x[i, j]), float(a[i, j]), float(b[i, j]), float(c[i, j]))
