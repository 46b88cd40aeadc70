# Round-4 profiling pass (run on the GPU box through gpurun): kernel-trace stats of the C4 / RES benches and the decode loop, the PMC
# passes on the gate/up GEMM that the bench line's `roofline.traffic` quotes (separate rocprofv3 --pmc runs, never combined with a trace
# domain), and FETCH / WRITE passes on the fused patchify (the north_star HBM line).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04prof
rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_c4 -- python $R/bench.py --steps 3 --warmup 1 --no-res --no-cpu-baseline > $O/p_c4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_res -- python $R/bench.py --workload res --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/p_res.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_dec -- python $R/tools/decode_bench.py --new 65 > $O/p_dec.log 2>&1
G="python $R/tools/gemm_one.py 20576 22016 4096 sw"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $G > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $G > $O/pmc_w.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/pmc_l2 --output-format csv -- $G > $O/pmc_l2.log 2>&1
rocprofv3 --pmc TCC_BUSY_sum TCC_CYCLE_sum -d $O/pmc_busy --output-format csv -- $G > $O/pmc_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- $G > $O/pmc_m.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum -d $O/pmc_ea --output-format csv -- $G > $O/pmc_ea.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc_eaw --output-format csv -- $G > $O/pmc_eaw.log 2>&1
P="python $R/tools/patchify_one.py 336"
rocprofv3 --pmc FETCH_SIZE -d $O/pp_fetch --output-format csv -- $P > $O/pp_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pp_write --output-format csv -- $P > $O/pp_w.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/pp_l2 --output-format csv -- $P > $O/pp_l2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/pp_mfma --output-format csv -- $P > $O/pp_m.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/pp_sq --output-format csv -- $P > $O/pp_sq.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/p_c4/*/*.db | head -1) $O/c4_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_res/*/*.db | head -1) $O/res_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $(ls $O/p_dec/*/*.db | head -1) $O/decode_kernel_stats.md > /dev/null
for d in pmc_fetch pmc_write pmc_l2 pmc_busy pmc_mfma pmc_ea pmc_eaw; do python tools/pmc_csv.py $O/$d gemm256; done > $O/gemm_pmc_summary.txt
python tools/make_traffic_json.py $O > $O/gemm_traffic.json
(echo "rocprofv3 --pmc passes on big::patchify_strip_kernel<9> (tools/patchify_one.py 336: B = 32, 336x336, 30 launches; counter means per launch)";
 echo "algorithmic bytes per launch: 60.87 MB (pixels 21.68 + patches 37.75 + packed weights 1.44); FETCH_SIZE / WRITE_SIZE are KiB;";
 echo "FETCH_SIZE x 2 for wide coalesced reads per MI355X_MICROARCH.md (the 16-byte pixel gathers of the A tiles are NOT that pattern: uncalibrated)";
 for d in pp_fetch pp_write pp_l2 pp_mfma pp_sq; do python tools/pmc_csv.py $O/$d patchify; done) > $O/patchify_pmc.txt
python tools/gemm_shapes.py > $O/gemm_shapes.txt 2>&1
for b in 1 2 4 8; do python tools/decode_bench.py --batch $b --new 33 2>&1 | tail -1; done > $O/decode_bench.txt
head -12 $O/c4_kernel_stats.md | cut -c1-130; head -12 $O/res_kernel_stats.md | cut -c1-130; cat $O/patchify_pmc.txt; cat $O/gemm_traffic.json | head -30
