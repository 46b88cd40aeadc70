# sustained GEMM legs with the SMI's clock / power sampled beside them
cd $GRAFT_REPO_ROOT
SECONDS_PER_LEG=6 python tools/gemm_hot.py > gpurun_out/hot.log 2>&1 &
PID=$!
sleep 14     # import + setup
for i in $(seq 1 60); do
  kill -0 $PID 2>/dev/null || break
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|Socket\|junction\|Average Graphics" | tr '\n' ' ' | sed 's/  */ /g'; echo
  sleep 0.7
done
wait $PID
cat gpurun_out/hot.log
