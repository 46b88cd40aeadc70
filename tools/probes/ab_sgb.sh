mkdir -p gpurun_out/r06; O=gpurun_out/r06/samg_sgb_ab.txt; : > $O
P=$PWD/tools/probes
for rep in 1 2; do
for v in base sgb22 sgb23 sgb11 sgb58; do
  ULL_LIB_PATH=$P/libatt_$v.so python tools/global_attn_ab.py /tmp/glob_dump.pt 2>&1 | grep -v amdgpu >> $O
done; done
cat $O
