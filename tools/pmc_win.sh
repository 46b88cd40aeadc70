# usage: pmc_win.sh "<counters>" -- one rocprofv3 --pmc pass over tools/win_attn_bench.py, the SAM window attention kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcw; rm -rf $O; mkdir -p $O
rocprofv3 --pmc $1 -d $O/p --output-format csv -- python $R/tools/win_attn_bench.py > $O/l.log 2>&1
python $R/tools/pmc_csv.py $O/p "sam_window_kernel"
python $R/tools/pmc_csv.py $O/p "attn_reg_kernel<128, 4, 2, 13, true,"
