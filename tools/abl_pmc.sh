# usage: abl_pmc.sh VARIANT...   (SHIP = the shipped library; others = tools/probes/lib_<VARIANT>.so; W8 = shipped library, 8-wave kernel)
# gate/up GEMM at K = 4096 and K = 8192 under `rocprofv3 --pmc`: cycles per K-step of the loop alone = the difference / (rounds * 64).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl_pmc; rm -rf $O; mkdir -p $O
for who in "$@"; do
  unset ULL_LIB_PATH; export TUNE=0
  if [ $who = W8 ]; then export TUNE=2097152; elif [ $who != SHIP ]; then export ULL_LIB_PATH=$R/tools/probes/lib_$who.so; fi
  for K in 4096 8192; do
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/${who}_$K --output-format csv -- python $R/tools/gemm_one.py 20576 22016 $K sw > $O/l.log 2>&1
  done
done
python - "$@" <<'PY'
import csv, glob, collections, sys, os
def load(who, K):
    f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + f"/gpurun_out/abl_pmc/{who}_{K}/*/*counter_collection.csv")[0]
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "gemm256" in r["Kernel_Name"]:
            d[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            d[r["Dispatch_Id"]]["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    v = list(d.values())[1:]          # drop the cold first launch
    n = len(v)
    return sum(x["GRBM_GUI_ACTIVE"] / 8 for x in v) / n, sum(x["dur"] for x in v) / n, sum(x["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 for x in v) / n
rounds = 6966 / 256
for who in sys.argv[1:]:
    c1, t1, m1 = load(who, 4096); c2, t2, m2 = load(who, 8192)
    per = (c2 - c1) / (rounds * 64)
    print(f"{who:22s} K=4096: {c1:.3e} cycles {t1/1e3:5.0f} us ({c1/t1:.2f} GHz, mfma busy {m1/c1:.3f})   loop: {per:6.0f} cycles / K-step (MFMA floor 2048)   "
          f"fixed: {(c1 - per * rounds * 64) / rounds:6.0f} cycles / round")
PY
