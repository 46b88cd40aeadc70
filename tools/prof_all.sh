cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/p_c4 -- python $R/bench.py --steps 3 --warmup 1 > $O/p_c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/p_res -- python $R/bench.py --workload res --steps 3 --warmup 1 > $O/p_res.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python $R/tools/gemm_one.py 20576 22016 4096 sw > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python $R/tools/gemm_one.py 20576 22016 4096 sw > $O/pmc_w.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- python $R/tools/gemm_one.py 20576 22016 4096 sw > $O/pmc_m.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/p_c4/*/*.db | head -1) $O/c4_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_res/*/*.db | head -1) $O/res_stats.md > /dev/null
python tools/pmc_csv.py $O/pmc_fetch gemm256 > $O/pmc_summary.txt
python tools/pmc_csv.py $O/pmc_write gemm256 >> $O/pmc_summary.txt
python tools/pmc_csv.py $O/pmc_mfma gemm256 >> $O/pmc_summary.txt
python bench.py > $O/bench_c4.json 2>/dev/null
python bench.py --workload res > $O/bench_res.json 2>/dev/null
ls $O/p_c4/*/ | head; cat $O/pmc_summary.txt; tail -c 600 $O/bench_c4.json
