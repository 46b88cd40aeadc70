# Round-5 profiling pass (run on the GPU box through gpurun): rocprofv3 --kernel-trace --stats of the C4 / RES benches and the decode loop.
# (gemm.hip is unchanged since round 4: the gate/up PMC traffic record profiles/r04_gemm_traffic.json still carries the kernel source's sha.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05prof
rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_c4 -- python $R/bench.py --steps 3 --warmup 1 --no-res --no-cpu-baseline > $O/p_c4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_res -- python $R/bench.py --workload res --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/p_res.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_dec -- python $R/tools/decode_bench.py --new 65 > $O/p_dec.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/p_c4/*/*.db | head -1) $O/c4_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_res/*/*.db | head -1) $O/res_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_dec/*/*.db | head -1) $O/decode_kernel_stats.md > /dev/null
tail -1 $O/p_c4.log | cut -c1-400
head -14 $O/c4_kernel_stats.md | cut -c1-130; head -16 $O/res_kernel_stats.md | cut -c1-130; head -8 $O/decode_kernel_stats.md | cut -c1-130
