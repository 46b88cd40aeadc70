#!/usr/bin/env python
"""Build the GEMM traffic record (profiles/rNN_gemm_traffic.json) from the rocprofv3 --pmc csv directories of tools/prof_r02.sh.
FETCH_SIZE is doubled (gfx950 counts the 128-B requests of wide coalesced reads as 64 B: MI355X_MICROARCH.md, HBM section)."""
import collections, csv, glob, hashlib, json, os, sys

root = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def means(sub):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm256" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


h = hashlib.sha256()
for n in ("gemm.hip", "ull_common.h"):
    h.update(open(os.path.join(ROOT, "u-llava_amd", "csrc", n), "rb").read())
f, w, l2, b, m = means("pmc_fetch"), means("pmc_write"), means("pmc_l2"), means("pmc_busy"), means("pmc_mfma")
ea, eaw = means("pmc_ea"), means("pmc_eaw")
M, N, K = 20576, 22016, 4096
alg = M * K * 2 + N * K * 2 + M * (N // 2) * 2
fetch_kb, write_kb = f.get("FETCH_SIZE", 0.0), w.get("WRITE_SIZE", 0.0)
rec = {"kernel": "big::gemm256w4_kernel<true> (gate/up + SwiGLU, M=20576 N=22016 K=4096, tile-major W)", "kernel_source_sha": h.hexdigest()[:16],
       "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
       "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, MI355X_MICROARCH.md HBM section)",
       "traffic_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024, "algorithmic_bytes_per_launch": alg,
       "traffic_over_algorithmic": (fetch_kb * 1024 * 2 + write_kb * 1024) / alg,
       "note": "counted at the L2->fabric boundary; Infinity Cache hits are included (not HBM-only)",
       "l2": {"TCC_HIT": l2.get("TCC_HIT_sum"), "TCC_MISS": l2.get("TCC_MISS_sum"), "TCC_REQ": l2.get("TCC_REQ_sum"),
              "hit_rate": (l2["TCC_HIT_sum"] / (l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"])) if l2.get("TCC_HIT_sum") else None,
              "TCC_BUSY_over_TCC_CYCLE": (b["TCC_BUSY_sum"] / b["TCC_CYCLE_sum"]) if b.get("TCC_CYCLE_sum") else None},
       "mfma": {"SQ_VALU_MFMA_BUSY_CYCLES": m.get("SQ_VALU_MFMA_BUSY_CYCLES"), "SQ_BUSY_CYCLES": m.get("SQ_BUSY_CYCLES"),
                "GRBM_GUI_ACTIVE_all_xcds": m.get("GRBM_GUI_ACTIVE"),
                "mfma_pipe_busy_fraction_of_simd_cycles": (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 256 * 4))
                if m.get("GRBM_GUI_ACTIVE") and m.get("SQ_VALU_MFMA_BUSY_CYCLES") else None},
       # memory-side (EA) request counters: every read leaves the L2 as a 128-byte request to the DRAM address space; none of them can tell an
       # Infinity-Cache hit from an HBM access (rocprofv3 -L on gfx950 / ROCm 7.2 lists no DF / UMC block: profiles/r03_counter_list_memory.txt)
       "ea": {"TCC_EA0_RDREQ": ea.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_DRAM": ea.get("TCC_EA0_RDREQ_DRAM_sum"), "TCC_EA0_RDREQ_128B": ea.get("TCC_EA0_RDREQ_128B_sum"),
              "read_bytes_128B_requests": (ea.get("TCC_EA0_RDREQ_128B_sum") or 0) * 128, "TCC_EA0_WRREQ": eaw.get("TCC_EA0_WRREQ_sum"),
              "TCC_EA0_WRREQ_64B": eaw.get("TCC_EA0_WRREQ_64B_sum"), "write_bytes_64B_requests": (eaw.get("TCC_EA0_WRREQ_64B_sum") or 0) * 64},
       # what the tile geometry alone predicts for the L2 -> fabric reads: an XCD's 32 CUs hold a 4 x 8 patch of 256x256 tiles that walk K
       # together, so a K-step needs 12 distinct 32-KiB panels for 64 panel loads: miss fraction 12 / 64 of T tiles x 2 x 256 x K x 2 B
       "fabric_read_floor_model_bytes": 12.0 / 64.0 * ((M + 255) // 256) * ((N + 255) // 256) * 2 * 256 * K * 2,
       "hbm_bytes_bounds": {"lower_compulsory": alg, "upper_all_fabric_traffic": fetch_kb * 1024 * 2 + write_kb * 1024,
                            "note": "no post-Infinity-Cache counter exists here; tools/mall_probe.py (profiles/r03_mall_probe.txt) shows re-read working sets "
                                    "up to 256 MiB served above HBM rate, and the GEMM re-reads each 2-MiB operand panel from all 8 XCDs within one round of tiles"},
       "source": "tools/prof_r03.sh / prof_r06.sh (separate rocprofv3 --pmc passes on tools/gemm_one.py 20576 22016 4096 sw)"}
print(json.dumps(rec, indent=1))
