# usage: hot_variants.sh VARIANT...  -- sustained LLaMA-layer GEMM rate of tools/probes/lib_<VARIANT>.so, the shipped library before and after
cd $GRAFT_REPO_ROOT
python tools/gemm_hot1.py shipped
for v in "$@"; do ULL_LIB_PATH=tools/probes/lib_$v.so python tools/gemm_hot1.py "$v"; done
python tools/gemm_hot1.py shipped
