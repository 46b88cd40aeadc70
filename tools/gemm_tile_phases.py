"""Where does a tile of the 4-wave GEMM spend its time?  Needs the debug build with per-block timestamps
(hipcc -DULL_GEMM_STAMPS -c gemm.hip, linked as tools/debug/libullava_stamps.so; ULL_LIB_PATH points at it).
Every block of one launch records (100-MHz wall clock): start, index arithmetic done, first K-tile in LDS, K-loop done, stores issued,
stores retired, and the CU it ran on; blocks are then chained per CU to get the hand-over gap between a block's last stamp and the next
block's first one.
usage: ULL_LIB_PATH=tools/debug/libullava_stamps.so python tools/gemm_tile_phases.py M N K [sw]"""
import ctypes, importlib, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("u-llava_amd.ops")
L = importlib.import_module("u-llava_amd._lib")
lib = ctypes.CDLL(L.LIB_PATH)
M, N, K = (int(v) for v in sys.argv[1:4])
sw = len(sys.argv) > 4
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
ops.register_tiled(w)
out = torch.empty(M, N // 2 if sw else N, device=dev, dtype=torch.bfloat16)
nblk = ((M + 255) // 256) * ((N + 255) // 256) + 4096
stamps = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.linear(x, w, swiglu=sw, out=out)
torch.cuda.synchronize()
lib.ull_debug_gemm_stamps_bf16.argtypes = [ctypes.c_void_p]
assert lib.ull_debug_gemm_stamps_bf16(ctypes.c_void_p(stamps.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.linear(x, w, swiglu=sw, out=out); e1.record()
torch.cuda.synchronize()
assert lib.ull_debug_gemm_stamps_bf16(ctypes.c_void_p(0)) == 0
st = stamps.cpu()
used = (st[:, :6] != 0).all(dim=1)                   # a block that never reached a stamp (stream-K helper blocks, blocks of another launch's
dropped = int((st[:, 0] != 0).sum()) - int(used.sum())  # shape) leaves zeros behind: its differences are wrapped garbage, not durations
st = st[used]
mono = (st[:, 1:6] >= st[:, 0:5]).all(dim=1)          # ... and so is a row whose stamps are not monotone (two launches writing the same slot)
dropped += int((~mono).sum())
st = st[mono]
if dropped:
    print(f"  ({dropped} block rows with missing / non-monotone stamps dropped)")
print(f"M {M} N {N} K {K}{' swiglu' if sw else ''}: {int(used.sum())} blocks, launch {e0.elapsed_time(e1) * 1e3:.1f} us")
t0 = st[:, 0].min()
us = lambda v: float(v) / 100.0                     # 10-ns ticks -> us
ph = {"index arithmetic": st[:, 1] - st[:, 0], "first K-tile (DMA latency)": st[:, 2] - st[:, 1], "K-loop": st[:, 3] - st[:, 2],
      "epilogue to last store issued": st[:, 4] - st[:, 3], "stores retire": st[:, 5] - st[:, 4], "whole block": st[:, 5] - st[:, 0]}
for k, v in ph.items():
    v = v.float() / 100.0
    print(f"  {k:32s}: mean {v.mean():7.2f} us   p10 {v.quantile(0.1):7.2f}   p50 {v.quantile(0.5):7.2f}   p90 {v.quantile(0.9):7.2f}   max {v.max():7.2f}")
# chain blocks per CU
by_cu = collections.defaultdict(list)
for r in st.tolist():
    by_cu[((r[7] >> 32) & 0xf, (r[7] >> 8) & 0xff)].append(r)      # (XCC, SE | SH | CU bits of HW_ID)
gaps, per_cu = [], []
for cu, rs in by_cu.items():
    rs.sort(key=lambda r: r[0])
    per_cu.append(len(rs))
    for a, b in zip(rs, rs[1:]):
        gaps.append((b[0] - a[5]) / 100.0)
gaps = torch.tensor(gaps)
print(f"  CUs seen {len(by_cu)}, blocks per CU {min(per_cu)}..{max(per_cu)}")
print(f"  hand-over gap (end of a block -> start of the next on the same CU): mean {gaps.mean():.2f} us  p10 {gaps.quantile(0.1):.2f}  p50 {gaps.quantile(0.5):.2f}  p90 {gaps.quantile(0.9):.2f}  max {gaps.max():.2f}")
first = (st[:, 0] - t0).float() / 100.0
last = (st[:, 5] - t0).float() / 100.0
print(f"  first block starts: spread {first.sort().values[:256].max():.2f} us over the first 256; last block ends at {last.max():.1f} us; "
      f"blocks ending in the last 10 %: {(last > 0.9 * last.max()).sum().item()}")
kl = (st[:, 3] - st[:, 2]).float() / 100.0
print(f"  K-loop per K-step (64): {kl.mean() / (K // 64) * 1e3:.1f} ns")
