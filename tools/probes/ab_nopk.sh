mkdir -p gpurun_out/r06; O=gpurun_out/r06/attn_nopk_ab.txt; : > $O
P=$PWD/tools/probes
for rep in 1 2; do
for v in base nopk; do
  echo "== $v (rep $rep)" >> $O
  ULL_LIB_PATH=$P/libatt_$v.so python tools/attn_prefill_bench.py 2>&1 | grep -v amdgpu >> $O
  ULL_LIB_PATH=$P/libatt_$v.so python tools/win_attn_ab.py /tmp/win_dump.pt 2>&1 | grep -v amdgpu >> $O
  ULL_LIB_PATH=$P/libatt_$v.so python tools/global_attn_ab.py /tmp/glob_dump.pt 2>&1 | grep -v amdgpu >> $O
done; done
cat $O
