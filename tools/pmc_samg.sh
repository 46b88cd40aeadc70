# PMC passes on the SAM global attention (RES shape, tools/global_attn_ab.py): $1 = kernel-name substring, ULL_LIB_PATH picks the build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${2:-new}; O=$R/gpurun_out/pmc_samg_$TAG; rm -rf $O; mkdir -p $O
G="python $R/tools/global_attn_ab.py /tmp/ga_pmc_$TAG.pt"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $O/a --output-format csv -- $G > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/b --output-format csv -- $G > $O/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC -d $O/c --output-format csv -- $G > $O/c.log 2>&1
cd $R
for d in a b c; do python tools/pmc_csv.py $O/$d "$1"; done
