"""A/B of two builds of the SAM global attention (64 x 64 grid, 4096 keys, decomposed rel-pos, V as rows: attn_stream_kernel) at the RES
shape (8 images, 16 heads, hd 80): the first run dumps its output, later runs compare with that dump.
usage: ULL_LIB_PATH=<other libullava_hip.so> python tools/global_attn_ab.py <dump.pt>;  python tools/global_attn_ab.py <dump.pt>"""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda"
nH, side, hd = 16, 64, 80
S, C = side * side, nH * hd
outs = {}
CASES = ((8, torch.bfloat16), (1, torch.bfloat16), (8, torch.float16))
if os.environ.get('ONLY_FIRST'):
    CASES = CASES[:1]
for (NB, dt) in CASES:
    gen = torch.Generator().manual_seed(NB)
    qkv = (0.5 * torch.randn(NB * S, 3 * C, generator=gen)).to(dt).to(dev)
    rph = (0.3 * torch.randn(2 * side - 1, hd, generator=gen)).to(dt).to(dev)
    rpw = (0.3 * torch.randn(2 * side - 1, hd, generator=gen)).to(dt).to(dev)
    strides = (S * 3 * C, hd, 3 * C)
    att = torch.empty(NB * S, C, device=dev, dtype=dt)
    fn = lambda: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], att, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False,
                               scale_mode=0, q_scale=hd ** -0.5, rel_h=rph, rel_w=rpw, rel_pos_hw=(side, side), v_strides=strides)
    fn(); torch.cuda.synchronize()
    outs[(NB, str(dt))] = att.cpu().clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{os.path.basename(os.environ.get('ULL_LIB_PATH', 'libullava_hip.so'))}  {str(dt)[6:]} NB={NB}: " + " ".join(f"{t:.1f}" for t in ts) + " us", flush=True)
path = sys.argv[1]
if os.path.exists(path):
    ref = torch.load(path)
    for k, o in outs.items():
        same = torch.equal(o.view(torch.int16), ref[k].view(torch.int16))
        print(f"   {k}: bit-identical to the first run: {same}  (max |diff| {(o.float() - ref[k].float()).abs().max().item():.3g}, finite {bool(torch.isfinite(o.float()).all())})")
else:
    torch.save(outs, path)
