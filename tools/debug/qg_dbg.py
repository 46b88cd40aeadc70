import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import per_op_inputs, load_fixture
ops = importlib.import_module("u-llava_amd.ops")
dt = torch.float16
fx = load_fixture("g6_per_op_fp16.pt")
x = per_op_inputs(fx["seed"], dt)
xa = x["act_x"].to("cuda:0")
eye = torch.eye(1024, dtype=dt, device="cuda:0")
got = ops.linear(xa, eye, act="quick_gelu").cpu()
ref = (x["act_x"] * torch.sigmoid(1.702 * x["act_x"]))
d = (got.float() - ref.float()).abs()
rel = d / ref.float().abs().clamp_min(1e-3 * float(ref.float().abs().max()))
idx = rel.flatten().topk(6).indices
for i in idx.tolist():
    t = x["act_x"].flatten()[i]
    u = (1.702 * t)
    s = torch.sigmoid(u)
    print(f"t={float(t):.6f} u={float(u):.6f} s={float(s):.8f} ref={float(ref.flatten()[i]):.8f} got={float(got.flatten()[i]):.8f} rel/2^-10={float(rel.flatten()[i]) * 1024:.2f}")
