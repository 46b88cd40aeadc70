# usage: pmc_attn.sh "<counters>" -- one rocprofv3 --pmc pass over tools/attn_bench.py, LLaMA prefill attention kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmca; rm -rf $O; mkdir -p $O
rocprofv3 --pmc $1 -d $O/p --output-format csv -- python $R/tools/attn_bench.py > $O/l.log 2>&1
python $R/tools/pmc_csv.py $O/p "${2:-attn_reg_kernel<128, 11, 0}"
