#!/bin/bash
# Is this box one of the "slow" ones (one-workgroup-per-CU kernels ~2x slower)?  If so, run the experiments that need one:
# soft start 0 / 8 / 20 us x tilings, and the per-workgroup timeline.  Otherwise exit at once (costs ~15 s of GPU time).
mkdir -p gpurun_out/slowbox
t=$(WHICH=fused REPS=5 python tools/update_bench.py 2>/dev/null | grep "E=47712 fused" | awk '{print int($3)}')
echo "fused update: $t us"
if [ -z "$t" ] || [ "$t" -lt 720 ]; then echo "normal box"; exit 0; fi
echo "SLOW BOX" | tee gpurun_out/slowbox/found.txt
{
for cfg in 3 0; do for sk in 0 8 20 40; do
  echo -n "tiling $cfg soft start $sk us: "; DPVO_FU_CFG=$cfg DPVO_FU_SKEW=$sk DPVO_UPDATE_AUTOTUNE=0 WHICH=fused REPS=10 python tools/update_bench.py 2>/dev/null | grep "E=47712 fused" | cut -c1-40
done; done
WHICH=unfused REPS=10 python tools/update_bench.py 2>/dev/null | grep "E=47712"
python tools/bench_summary.py gpurun_out/slowbox 1
if [ -f dpvo_amd/libdpvo_hip_trace.so ]; then for sk in 0 20; do echo "== trace, soft start $sk"; DPVO_FU_SKEW=$sk DPVO_HIP_LIB=$PWD/dpvo_amd/libdpvo_hip_trace.so MODE=seven python tools/fu_trace.py 2>&1 | grep -v amdgpu | grep "==\|GEMM h\|GEMM 896\|LN\|gate\*res"; done; fi
} 2>&1 | tee gpurun_out/slowbox/results.txt
