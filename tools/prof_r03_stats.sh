# Round-3 kernel-trace refresh after the attention / decode changes (the PMC passes of prof_r03.sh are on the GEMM, whose source did
# not change): C4, RES and decode-loop kernel statistics, then the default bench line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03stats
rm -rf $O; mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats -d $O/p_c4 -- python $R/bench.py --steps 3 --warmup 1 --no-res --no-cpu-baseline > $O/p_c4.log 2>&1
timeout 500 rocprofv3 --kernel-trace --stats -d $O/p_res -- python $R/bench.py --workload res --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/p_res.log 2>&1
timeout 500 rocprofv3 --kernel-trace --stats -d $O/p_dec -- python $R/tools/decode_bench.py --new 65 > $O/p_dec.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/p_c4/*/*.db | head -1) $O/c4_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_res/*/*.db | head -1) $O/res_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_dec/*/*.db | head -1) $O/decode_kernel_stats.md > /dev/null
python tools/attn_prefill_bench.py > $O/attn_prefill.txt 2>&1
for b in 1 2 4 8; do python tools/decode_bench.py --batch $b --new 33 2>&1 | tail -1; done > $O/decode_bench.txt
python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
head -12 $O/c4_kernel_stats.md | cut -c1-130; head -12 $O/res_kernel_stats.md | cut -c1-130; cat $O/decode_bench.txt; tail -c 1800 $O/bench_default.json
