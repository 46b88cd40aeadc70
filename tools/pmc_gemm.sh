cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc2; mkdir -p $O
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCC_BUSY_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  for v in default XONLY; do
    if [ $v = default ]; then unset ULL_LIB_PATH; else export ULL_LIB_PATH=$R/build/abl/lib_$v.so; fi
    rocprofv3 --pmc $set -d $O/${v}_$i --output-format csv -- python $R/tools/gemm_one.py 20576 22016 4096 sw > $O/${v}_$i.log 2>&1
  done
done
unset ULL_LIB_PATH
cd $R
for v in default XONLY; do echo "== $v"; for i in 1 2 3 4 5; do python tools/pmc_csv.py $O/${v}_$i gemm256 | grep -v "^void"; done; done
