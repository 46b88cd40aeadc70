# usage: hot_tunes.sh TUNEBITS...  -- sustained LLaMA-layer GEMM rate of the shipped library under ULL_GEMM_TUNE_* bits
cd $GRAFT_REPO_ROOT
for t in "$@"; do python tools/gemm_hot1.py "tune=$t" $t; done
