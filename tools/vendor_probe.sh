cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/vendor; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python $GRAFT_REPO_ROOT/tools/vendor_probe.py > $O/run.log 2>&1
cat $O/run.log | grep vendor
python - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/vendor/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
seen = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    d = seen.setdefault(k, dict(n=0, t=0, r=r))
    d["n"] += 1; d["t"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, d in seen.items():
    r = d["r"]
    if d["t"] / d["n"] < 2e5: continue
    print(f"{d['t']/d['n']/1e3:9.1f} us x{d['n']:3d} wg={r.get('Workgroup_Size_X')} grid={r.get('Grid_Size_X')}x{r.get('Grid_Size_Y')}x{r.get('Grid_Size_Z')} lds={r.get('LDS_Block_Size')} vgpr={r.get('VGPR_Count')} agpr={r.get('Accum_VGPR_Count')} sgpr={r.get('SGPR_Count')} scratch={r.get('Scratch_Size')}\n    {k}")
PY
