#!/usr/bin/env python
"""Where does a wave of the prefill attention (attn_reg_kernel, LLaMA C4 shape) spend its cycles?  Needs the debug build with per-wave cycle
totals per phase: tools/build_attn_stamps.sh -> tools/debug/libullava_attn_stamps.so, loaded through ULL_LIB_PATH.
Every wave of one launch records (shader clock, s_memtime): prologue, phase-1 wait (vmcnt + barrier + next DMA issue) and compute (QK^T MFMA +
score epilogue), softmax, phase-3 wait and compute (P V), epilogue; plus the tiles it computed on and the CU it ran on.
usage: ULL_LIB_PATH=tools/debug/libullava_attn_stamps.so python tools/attn_phase_times.py [B H S hd causal]"""
import ctypes, importlib, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
L = importlib.import_module("u-llava_amd._lib")
lib = ctypes.CDLL(L.LIB_PATH)
B, H, S, hd, causal = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (32, 32, 643, 128, 1)
NWV = int(os.environ.get("NWV", "4"))
dev, BF = "cuda:0", torch.bfloat16
D = H * hd
qkv = (torch.randn(B * S, 3 * D, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 0.5).to(BF)
st = (S * 3 * D, hd, 3 * D)
att = torch.empty(B * S, D, device=dev, dtype=BF)
f = lambda: ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], att, B, H, S, S, hd, st, st, (S * D, hd, D), None, causal=bool(causal), scale_mode=1,
                          scale=hd ** -0.5, v_strides=st)
nq = (S + 16 * NWV - 1) // (16 * NWV)
nblk = ((B * H + 7) // 8) * 8 * nq
stamps = torch.zeros(nblk * NWV, 12, dtype=torch.int64, device=dev)
for _ in range(5):
    f()
torch.cuda.synchronize()
fn = lib.ull_debug_attn_stamps_bf16
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(stamps.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record()
torch.cuda.synchronize()
assert fn(ctypes.c_void_p(0)) == 0
t = stamps.cpu()
t = t[t[:, 7] != 0]
us_launch = e0.elapsed_time(e1) * 1e3
names = ["prologue (Q, mask, first DMA)", "phase-1 wait (vmcnt+barrier+DMA issue)", "phase-1 compute (QK^T + scores)", "softmax (2 passes)",
         "phase-3 wait", "phase-3 compute (P V)", "epilogue (O stores)"]
tot = t[:, 7].double()
nkt_w = (t[:, 8] & 0xffff).double()
nkt = ((t[:, 8] >> 16) & 0xffff).double()
print(f"B {B} H {H} S {S} hd {hd} causal {causal}: {t.shape[0]} waves, launch {us_launch:.1f} us (with stamps)")
print(f"  wave lifetime: mean {tot.mean():.0f} cycles, p10 {tot.quantile(0.1):.0f}, p90 {tot.quantile(0.9):.0f}; tiles computed per wave: mean {nkt_w.mean():.2f}, "
      f"tile steps per block: mean {nkt.mean():.2f}")
allc = tot.sum()
for i, n in enumerate(names):
    v = t[:, i].double()
    per = v / (nkt if i in (1, 4) else nkt_w if i in (2, 3, 5) else torch.ones_like(v)).clamp_min(1)
    print(f"  {n:42s}: {100 * v.sum() / allc:5.1f} % of wave cycles   mean {v.mean():8.0f} cycles/wave   per tile(step): mean {per.mean():7.0f}  p10 {per.quantile(0.1):7.0f}  p90 {per.quantile(0.9):7.0f}")
# clock rate: cycles of the whole launch per CU vs wall
by_cu = collections.defaultdict(list)
for r in t.tolist():
    by_cu[((r[10] >> 32) & 0xf, (r[10] >> 8) & 0xff)].append(r)
occ = []
for cu, rs in by_cu.items():
    lo = min(r[9] for r in rs); hi = max(r[9] + r[7] for r in rs)
    occ.append(sum(r[7] for r in rs) / max(1, hi - lo))
occ = torch.tensor(occ)
span = max(r[9] + r[7] for r in t.tolist()) - min(r[9] for r in t.tolist())
print(f"  CUs seen {len(by_cu)}; waves resident per CU (sum of lifetimes / span): mean {occ.mean():.2f}  min {occ.min():.2f}  max {occ.max():.2f}; "
      f"launch span {span} cycles = {span / us_launch / 1e3:.2f} GHz x {us_launch:.1f} us")
