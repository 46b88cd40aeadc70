"""RES step with the SAM stream and the CLIP + LLaMA stream on DISJOINT compute units (hipExtStreamCreateWithCUMask) instead of time-slicing
the whole chip between two ordinary streams (round-5 record: in-step SAM window attention +45 %, global +17 % vs stand-alone).
Sweeps the share of CUs given to the SAM stream; prints ms per RES step (batch 8) for: one stream, two ordinary streams (shipped), and each split.
usage: python tools/probes/res_cu_partition.py [--interleave]"""
import ctypes, importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
bench = importlib.import_module("bench")
ops = importlib.import_module("u-llava_amd.ops")
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count
WORDS = (NCU + 31) // 32


def masked_stream(bits):
    """bits: iterable of CU indices that are ENABLED."""
    arr = (ctypes.c_uint32 * WORDS)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(WORDS), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


step, batch, S, cfg, desc, fl, model = bench.workload_step("res", dev, 0)


def timeit(n=6, w=2):
    with torch.no_grad():
        for _ in range(w):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"{NCU} CUs, mask words {WORDS}", flush=True)
model.overlap_sam_encoder = False
model._side = None
print(f"one stream: {timeit():.2f} ms per step", flush=True)
model.overlap_sam_encoder = True
model._side = None
print(f"two ordinary streams (shipped): {timeit():.2f} ms per step", flush=True)
interleave = "--interleave" in sys.argv
main_default = torch.cuda.current_stream()
for sam_cus in (96, 112, 128, 144, 160):
    if interleave:          # SAM gets every CU whose index mod 8 falls in a set (a slice of every XCD if bits enumerate XCD-major ... or whole XCDs otherwise)
        k = sam_cus * 8 // NCU
        sam_bits = [i for i in range(NCU) if i % 8 < k]
    else:
        sam_bits = list(range(sam_cus))
    rest = [i for i in range(NCU) if i not in set(sam_bits)]
    side = masked_stream(sam_bits)
    mainm = masked_stream(rest)
    model._side = side
    for policy in (8192, 2048):
        with torch.cuda.stream(mainm):
            # the model reads ops.streamk_policy inside forward (8192 when two streams run); here the step itself runs on a masked main stream
            t = timeit()
        print(f"SAM on {len(sam_bits)} CUs / LLM on {len(rest)} CUs ({'interleaved' if interleave else 'contiguous'} bits): {t:.2f} ms per step", flush=True)
        break
    # SAM stream masked, main stream unmasked (may use every CU, the SAM kernels only theirs)
    model._side = side
    t = timeit()
    print(f"SAM on {len(sam_bits)} CUs / LLM unmasked: {t:.2f} ms per step", flush=True)
