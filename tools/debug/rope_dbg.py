import importlib, sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ops = importlib.import_module("u-llava_amd.ops")
from oracle import ullava_oracle as O
H = torch.float16
g = torch.Generator().manual_seed(12)
Hn, hd, T = 4, 128, 50
qk = torch.randn(T, 2 * Hn * hd, generator=g).to(H)
pos = torch.arange(T).unsqueeze(0)
cos, sin = O.rope_tables(pos, hd, 10000.0, H)
q = qk[:, :Hn * hd].view(1, T, Hn, hd).transpose(1, 2)
k = qk[:, Hn * hd:].view(1, T, Hn, hd).transpose(1, 2)
rq, rk = O.apply_rope(q, k, cos, sin)
buf = qk.clone().cuda()
inv = (1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))).cuda()
ops.rope_inplace(buf, 2 * Hn * hd, pos[0].cuda(), inv, T, 2 * Hn, hd)
got = buf.cpu()[:, :Hn * hd].view(T, Hn, hd)
ref = rq[0].transpose(0, 1)
d = (got.float() - ref.float()).abs()
print("n diff", int((d > 0).sum()), "of", d.numel())
idx = (d > 0).nonzero()[:6]
c32 = cos.float(); s32 = sin.float()
for t, h, j in idx.tolist():
    x1 = float(q[0, h, t, j]); jj = (j + 64) % 128
    x2 = float(q[0, h, t, jj])
    c, s = float(cos[0, t, j]), float(sin[0, t, j])
    rot = -x2 if j < 64 else x2
    p1 = torch.tensor(x1 * c).to(H); p2 = torch.tensor(rot * s).to(H)
    manual = (p1.float() + p2.float()).to(H)
    print(t, h, j, "got", float(got[t, h, j]), "ref", float(ref[t, h, j]), "manual", float(manual), "| x1", x1, "x2", x2, "c", c, "s", s, "p1", float(p1), "p2", float(p2),
          "torch q*cos", float((q[0, h, t, j] * cos[0, t, j])), "rot*sin", float((torch.tensor(rot).to(H) * sin[0, t, j])))
