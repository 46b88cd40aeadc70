#!/usr/bin/env python
"""Per-basic-block instruction census of one kernel in a hipcc -save-temps .s file.  usage: asm_blocks.py file.s kernel-substring"""
import re, sys
L = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2]
start = next(i for i, l in enumerate(L) if re.match(r"^_Z\S*:", l) and sub in l)
end = next(i for i in range(start + 1, len(L)) if "codeLenInByte" in L[i])
print(L[start].split(":")[0], [l.strip() for l in L[end:end + 12] if any(k in l for k in ("codeLen", "NumVgprs", "NumAgprs", "ScratchSize"))])
cur = ["entry", start, 0, 0, 0, 0, 0]; blocks = [cur]
for i in range(start, end):
    l = L[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), i - start, 0, 0, 0, 0, 0]; blocks.append(cur)
    for k, pat in ((2, "v_mfma"), (3, "v_accvgpr"), (4, "scratch_"), (5, "ds_read"), (6, "global_load_lds")):
        if pat in l: cur[k] += 1
print("block, line, mfma, accvgpr, scratch, ds_read, lds_dma")
for b in blocks:
    if b[2] or b[3] > 4 or b[4]: print(b)
