# GROUP_M sweep of the gate/up launch: time (plain run) and L2 -> fabric read bytes (FETCH_SIZE x 2) per raster
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/raster; rm -rf $O; mkdir -p $O
for gm in 1 2 4 8 16; do
  export TUNE=$((gm << 16))
  # (FETCH_SIZE takes 3 of the 4 TCC slots: it goes alone -- with two more TCC counters in the same pass rocprofv3 aborts and hangs)
  timeout 150 rocprofv3 --pmc FETCH_SIZE -d $O/f_$gm --output-format csv -- python $R/tools/gemm_one.py 20576 22016 4096 sw > $O/f_$gm.log 2>&1
  timeout 150 rocprofv3 --kernel-trace --stats -d $O/t_$gm --output-format csv -- python $R/tools/gemm_one.py 20576 22016 4096 sw > $O/t_$gm.log 2>&1
done
unset TUNE
cd $R
python - <<'PY'
import csv, glob, collections
for gm in (1, 2, 4, 8, 16):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/raster/f_{gm}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm256" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = []
    for f in glob.glob(f"gpurun_out/raster/t_{gm}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm256" in r.get("Kernel_Name", ""):
                t.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    print(f"GROUP_M {gm:2d}: fabric reads {m.get('FETCH_SIZE', 0) * 1024 * 2 / 1e9:6.2f} GB  kernel {min(t) if t else 0:8.1f} us (min of {len(t)})")
PY
