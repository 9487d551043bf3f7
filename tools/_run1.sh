export DPVO_UPDATE_AUTOTUNE=0
for lib in "" dw2c4 dw2c6; do
  for rep in 1 2; do
  if [ -z "$lib" ]; then unset DPVO_HIP_LIB; else export DPVO_HIP_LIB=$PWD/dpvo_amd/libdpvo_hip_$lib.so; fi
  echo "== lib=$lib rep=$rep"; WHICH=fused REPS=30 python tools/update_bench.py 2>&1 | grep fused
  done
done
unset DPVO_HIP_LIB
echo "== cfg0 (96x1)"; DPVO_FU_CFG=0 WHICH=fused REPS=30 python tools/update_bench.py 2>&1 | grep fused
echo "== trace default tiling"; MODE=seven python tools/fu_trace.py 2>&1 | tail -60
