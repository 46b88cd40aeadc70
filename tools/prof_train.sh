cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/train_prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/p -- python $R/tools/train_bench.py > $O/run.log 2>&1
cd $R; python tools/rocpd_stats.py $(ls $O/p/*/*.db | head -1) $O/train_kernel_stats.md > /dev/null
head -30 $O/train_kernel_stats.md | cut -c1-150; grep "training step" $O/run.log
