cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02dec; rm -rf $O; mkdir -p $O
python $R/tools/decoder_bench.py both 20 > $O/decoder_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/p_fused -- python $R/tools/decoder_bench.py fused 10 > $O/p_fused.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/p_unfused -- python $R/tools/decoder_bench.py unfused 10 > $O/p_unfused.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/p_fused/*/*.db | head -1) $O/decoder_fused_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_unfused/*/*.db | head -1) $O/decoder_unfused_kernel_stats.md > /dev/null
cat $O/decoder_bench.txt; head -24 $O/decoder_fused_kernel_stats.md; echo; head -12 $O/decoder_unfused_kernel_stats.md
