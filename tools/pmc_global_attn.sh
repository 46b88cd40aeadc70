# usage: pmc_global_attn.sh "<counters>" -- one rocprofv3 --pmc pass over tools/global_attn_ab.py (SAM global attention, attn_stream_kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcg; rm -rf $O; mkdir -p $O
rocprofv3 --pmc $1 -d $O/p --output-format csv -- python $R/tools/global_attn_ab.py /tmp/pmcg_dump.pt > $O/l.log 2>&1
python $R/tools/pmc_csv.py $O/p "attn_stream_kernel"
