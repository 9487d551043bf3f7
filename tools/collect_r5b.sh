#!/bin/bash
# tools/collect_r5b.sh <tag>: round 5, second GPU call -- K7 with the state parked in AGPRs against the image round trips (A/B on one box),
# 96-row chain tiles at two workgroups per CU, the refined tracker-level checker, and the search for the free-running scenario.
tag=${1:-r5b}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 600 python -m pytest tests/test_gpu_update.py tests/test_golden.py -m gpu -q 2>&1 | grep -v "$F" | tail -5 > $out/pytest_update.txt; cat $out/pytest_update.txt
for rep in 1 2; do
for v in "" nopark c96d2 c96d3 k7d5 k7d4; do
  lib=$root/dpvo_amd/libdpvo_hip${v:+_$v}.so
  echo "== ${v:-product} (rep $rep)" >> $out/update_ab.txt
  DPVO_HIP_LIB=$lib WHICH=fused REPS=30 timeout 120 python tools/update_bench.py 2>&1 | grep -v "$F" >> $out/update_ab.txt
done; done
cat $out/update_ab.txt
( cd /tmp && export TMPDIR=/tmp && for v in "" nopark c96d2; do lib=$root/dpvo_amd/libdpvo_hip${v:+_$v}.so; rm -rf /tmp/ks_$v; DPVO_HIP_LIB=$lib WHICH=fused REPS=20 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -- python $root/tools/update_bench.py > /dev/null 2>&1; f=$(find /tmp/ks_$v -name "*kernel_stats.csv" | head -1); echo "== ${v:-product}" >> $out/update_kernels.txt; python $root/tools/kstats.py $f 12 >> $out/update_kernels.txt; done )
cat $out/update_kernels.txt
timeout 900 python -m pytest tests/test_zz_ref_pipeline.py -m gpu -q -s 2>&1 | grep -v "$F" > $out/pytest_zz.txt; tail -5 $out/pytest_zz.txt
for s in 0.003 0.001; do
  timeout 300 python tools/ref_parity.py --frames 80 --scenarios A,T --attribute --delta-scale $s 2>&1 | grep -v "$F" > $out/well_${s}_none.txt
done
ls -la $out
