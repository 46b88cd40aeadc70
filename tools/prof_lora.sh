cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lora_prof; rm -rf $O; mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats -d $O/p -- python $R/bench.py --workload train --train-config lora --steps 2 --warmup 1 > $O/run.log 2>&1
cd $R; python tools/rocpd_stats.py $(ls $O/p/*/*.db | head -1) $O/stats.md > /dev/null
head -22 $O/stats.md | cut -c1-150
