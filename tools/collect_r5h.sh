#!/bin/bash
# tools/collect_r5h.sh <tag>: chol.hip's panel kernel (LDS round trip off the dependent chain, packed-f32 trailing updates) against the
# round-4 kernel (dpvo_amd/libdpvo_hip_oldchol.so) on one box: tests, solve times, per-kernel table
tag=${1:-r5h}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 600 python -m pytest tests/test_gpu_chol.py tests/test_gpu_ba.py -m gpu -q 2>&1 | grep -v "$F" | tail -3 | tee $out/pytest_chol.txt
for v in "" oldchol ""; do
  lib=$root/dpvo_amd/libdpvo_hip${v:+_$v}.so; echo "== ${v:-product}" | tee -a $out/chol_bench.txt
  DPVO_HIP_LIB=$lib timeout 200 python tools/chol_bench.py 2>&1 | grep -v "$F" | tee -a $out/chol_bench.txt
done
for v in "" oldchol; do
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cb$v && DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip${v:+_$v}.so NS=630,4794 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb$v -- python $root/tools/chol_bench.py > /dev/null 2>&1; f=$(find /tmp/cb$v -name "*kernel_stats.csv" | head -1); echo "== ${v:-product}" >> $out/chol_kernels.txt; python $root/tools/kstats.py $f 6 >> $out/chol_kernels.txt )
done; cat $out/chol_kernels.txt
