cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vendor_pmc; rm -rf $O; mkdir -p $O
for who in vendor ours; do
  if [ $who = vendor ]; then G="python $R/tools/vendor_one.py 20576 22016 4096"; else G="python $R/tools/gemm_one.py 20576 22016 4096 sw"; fi
  rocprofv3 --pmc FETCH_SIZE -d $O/${who}_fetch --output-format csv -- $G > $O/l.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/${who}_l2 --output-format csv -- $G > $O/l.log 2>&1
  rocprofv3 --pmc TCC_BUSY_sum TCC_CYCLE_sum -d $O/${who}_busy --output-format csv -- $G > $O/l.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU -d $O/${who}_inst --output-format csv -- $G > $O/l.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/${who}_mfma --output-format csv -- $G > $O/l.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/${who}_lds --output-format csv -- $G > $O/l.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum -d $O/${who}_tcp --output-format csv -- $G > $O/l.log 2>&1
done
cd $R
for who in vendor ours; do for d in fetch l2 busy inst mfma lds tcp; do python tools/pmc_csv.py $O/${who}_$d $([ $who = vendor ] && echo Cijk || echo gemm256); done; done
