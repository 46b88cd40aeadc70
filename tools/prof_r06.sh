# Round-6 profiling pass (run on the GPU box through gpurun): rocprofv3 --kernel-trace --stats of the C4 / RES benches and the decode loop, the
# PMC passes on the gate/up GEMM (separate rocprofv3 --pmc runs, never combined with a trace domain) and the traffic record with the kernel
# source's sha (gemm.hip's TEXT changed in round 6 -- the ablation switches moved to tools/probes/lab -- its device code did not).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06prof
rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_c4 -- python $R/bench.py --steps 3 --warmup 1 --no-res --no-cpu-baseline > $O/p_c4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_res -- python $R/bench.py --workload res --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/p_res.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_dec -- python $R/tools/decode_bench.py --new 65 > $O/p_dec.log 2>&1
G="python $R/tools/gemm_one.py 20576 22016 4096 sw"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $G > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $G > $O/pmc_w.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/pmc_l2 --output-format csv -- $G > $O/pmc_l2.log 2>&1
timeout 300 rocprofv3 --pmc TCC_BUSY_sum TCC_CYCLE_sum -d $O/pmc_busy --output-format csv -- $G > $O/pmc_b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- $G > $O/pmc_m.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmc_ea --output-format csv -- $G > $O/pmc_ea.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc_eaw --output-format csv -- $G > $O/pmc_eaw.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/p_c4/*/*.db | head -1) $O/c4_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_res/*/*.db | head -1) $O/res_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_dec/*/*.db | head -1) $O/decode_kernel_stats.md > /dev/null
for d in pmc_fetch pmc_write pmc_l2 pmc_busy pmc_mfma pmc_ea pmc_eaw; do python tools/pmc_csv.py $O/$d gemm256; done > $O/gemm_pmc_summary.txt
python tools/make_traffic_json.py $O > $O/gemm_traffic.json
rm -rf $O/p_c4 $O/p_res $O/p_dec $O/pmc_*/          # (the raw databases are hundreds of MB: only the summaries travel back)
tail -1 $O/p_c4.log | cut -c1-400
head -14 $O/c4_kernel_stats.md | cut -c1-130; head -16 $O/res_kernel_stats.md | cut -c1-130; head -8 $O/decode_kernel_stats.md | cut -c1-130; cat $O/gemm_traffic.json | head -20
