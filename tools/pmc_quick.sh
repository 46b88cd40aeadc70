# usage: pmc_quick.sh "<counters>" -- one rocprofv3 --pmc pass over the shipped gate/up GEMM launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O
rocprofv3 --pmc $1 -d $O/p --output-format csv -- python $R/tools/gemm_one.py 20576 22016 4096 sw > $O/l.log 2>&1
python $R/tools/pmc_csv.py $O/p gemm256
