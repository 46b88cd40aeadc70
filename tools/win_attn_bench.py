"""SAM window attention at the RES shape (8 images -> 200 windows x 16 heads x 196 keys, hd 80): generic fused rel-pos kernel against
the row-padded form.  usage: python tools/win_attn_bench.py"""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda"
NB, nH, side, hd = 200, 16, 14, 80
S, C = side * side, nH * hd
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(NB * S, 3 * C, generator=g)).to(torch.bfloat16).to(dev)
rph = (0.3 * torch.randn(2 * side - 1, hd, generator=g)).to(torch.bfloat16).to(dev)
rpw = (0.3 * torch.randn(2 * side - 1, hd, generator=g)).to(torch.bfloat16).to(dev)
strides = (S * 3 * C, hd, 3 * C)
att = torch.empty(NB * S, C, device=dev, dtype=torch.bfloat16)


def run(win):
    vt = ops.transpose_v(qkv[:, 2 * C:], S * 3 * C, 3 * C, NB, S, nH, hd, win_kw=side if win else 0)
    def one():
        ops.attention(qkv, qkv[:, C:], vt, att, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False, scale_mode=0,
                      q_scale=hd ** -0.5, rel_h=rph, rel_w=rpw, rel_pos_hw=(side, side), win_padded=win)
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        one()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3, att.clone()


for rep in range(2):
    t0, a0 = run(False)
    t1, a1 = run(True)
    print(f"generic {t0:.1f} us   row-padded {t1:.1f} us   outputs differing {float((a0 != a1).float().mean()):.4%}")
