"""A/B of two builds of the SAM window-attention kernel (ull_sam_window_attention) at the RES shape and at shapes with partial windows: the
first run dumps its outputs, every later run compares bit for bit with that dump.  Both dtypes.
usage: ULL_LIB_PATH=<other libullava_hip.so> python tools/win_attn_ab.py <dump.pt>;  python tools/win_attn_ab.py <dump.pt>"""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda"
nH, side, hd = 16, 14, 80
C = nH * hd
outs = {}
for (B, g, dt) in ((8, 64, torch.bfloat16), (1, 64, torch.bfloat16), (3, 20, torch.bfloat16), (2, 37, torch.bfloat16), (8, 64, torch.float16), (2, 37, torch.float16)):
    gen = torch.Generator().manual_seed(B * 100 + g)
    qkv = torch.randn(B * g * g, 3 * C, generator=gen).to(dt).to(dev)
    bias = torch.randn(3 * C, generator=gen).to(dt).to(dev)
    rph = (0.3 * torch.randn(2 * side - 1, hd, generator=gen)).to(dt).to(dev)
    rpw = (0.3 * torch.randn(2 * side - 1, hd, generator=gen)).to(dt).to(dev)
    fn = lambda: ops.sam_window_attention(qkv, bias, rph, rpw, B, g, g, nH, hd, side)
    o = fn()
    torch.cuda.synchronize()
    outs[(B, g, str(dt))] = o.cpu()
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"{os.path.basename(os.environ.get('ULL_LIB_PATH', 'libullava_hip.so'))}  {str(dt)[6:]} B={B} grid={g}x{g}: " + " ".join(f"{t:.1f}" for t in ts) + " us", flush=True)
path = sys.argv[1]
if os.path.exists(path):
    ref = torch.load(path)
    for k, o in outs.items():
        same = torch.equal(o.view(torch.int16), ref[k].view(torch.int16))
        d = (o.float() - ref[k].float()).abs().max().item()
        print(f"   {k}: bit-identical to the first run: {same}  (max |diff| {d:.3g}, finite {bool(torch.isfinite(o.float()).all())})")
else:
    torch.save(outs, path)
