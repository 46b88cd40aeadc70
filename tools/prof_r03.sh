# Round-3 profiling pass (run on the GPU box through gpurun): kernel-trace stats of the C4 and RES benches, PMC passes on the gate/up
# GEMM (separate rocprofv3 --pmc runs, never combined with a trace domain), traffic record with the kernel-source sha, the list of
# memory-side counters this rocprofv3 offers (there is none behind the Infinity Cache) and the Infinity-Cache probe.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/p_c4 -- python $R/bench.py --steps 3 --warmup 1 --no-res --no-cpu-baseline > $O/p_c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/p_res -- python $R/bench.py --workload res --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/p_res.log 2>&1
G="python $R/tools/gemm_one.py 20576 22016 4096 sw"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $G > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $G > $O/pmc_w.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/pmc_l2 --output-format csv -- $G > $O/pmc_l2.log 2>&1
rocprofv3 --pmc TCC_BUSY_sum TCC_CYCLE_sum -d $O/pmc_busy --output-format csv -- $G > $O/pmc_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- $G > $O/pmc_m.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmc_ea --output-format csv -- $G > $O/pmc_ea.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc_eaw --output-format csv -- $G > $O/pmc_eaw.log 2>&1
rocprofv3 -L 2>/dev/null | grep -i "Counter_Name" | grep -i -E "mall|dram|hbm|EA0|EA_|umc|df_|fabric" | sort -u > $O/counter_list_memory.txt
cd $R
python tools/rocpd_stats.py $(ls $O/p_c4/*/*.db | head -1) $O/c4_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_res/*/*.db | head -1) $O/res_kernel_stats.md > /dev/null
for d in pmc_fetch pmc_write pmc_l2 pmc_busy pmc_mfma pmc_ea pmc_eaw; do python tools/pmc_csv.py $O/$d gemm256; done > $O/gemm_pmc_summary.txt
python tools/make_traffic_json.py $O > $O/gemm_traffic.json
python tools/mall_probe.py > $O/mall_probe.txt 2>&1
python tools/gemm_shapes.py > $O/gemm_shapes.txt 2>&1
python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
head -14 $O/c4_kernel_stats.md; cat $O/gemm_traffic.json; cat $O/mall_probe.txt; tail -c 2500 $O/bench_default.json
