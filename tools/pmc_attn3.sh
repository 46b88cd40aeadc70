# PMC passes on the LLaMA prefill attention kernel (C4 shape): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_attn3; rm -rf $O; mkdir -p $O
G="python $R/tools/attn_prefill_bench.py"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $O/a --output-format csv -- $G > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/b --output-format csv -- $G > $O/b.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH -d $O/c --output-format csv -- $G > $O/c.log 2>&1
cd $R
for d in a b c; do python tools/pmc_csv.py $O/$d "attn_reg_kernel<128"; done
