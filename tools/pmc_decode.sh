# PMC passes on the decode GEMV stream (tools/decode_bench.py, batch 1): memory-side request counters per gemv launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_decode; rm -rf $O; mkdir -p $O
G="python $R/tools/decode_bench.py --new 9"
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum GRBM_GUI_ACTIVE -d $O/a --output-format csv -- $G > $O/a.log 2>&1
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum -d $O/b --output-format csv -- $G > $O/b.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum SQ_WAVES SQ_BUSY_CYCLES -d $O/c --output-format csv -- $G > $O/c.log 2>&1
cd $R
for d in a b c; do python tools/pmc_csv.py $O/$d "gemv_kernel"; done
tail -2 $O/a.log
