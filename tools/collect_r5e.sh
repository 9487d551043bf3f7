#!/bin/bash
# tools/collect_r5e.sh <tag>: round 5 (second session), first GPU call -- the full GPU suite at HEAD (no -x: every failure is seen),
# K7 parked-in-AGPRs against the image round trips (A/B on one box + per-kernel table), and a bench line.  Output: gpurun_out/<tag>/.
tag=${1:-r5e}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 1200 python -m pytest tests -m gpu -q -s --durations=8 2>&1 | grep -v "$F" > $out/pytest_gpu.txt; tail -30 $out/pytest_gpu.txt
for rep in 1 2; do
for v in "" nopark; do
  lib=$root/dpvo_amd/libdpvo_hip${v:+_$v}.so
  echo "== ${v:-product} (rep $rep)" >> $out/update_ab.txt
  DPVO_HIP_LIB=$lib WHICH=fused REPS=30 timeout 120 python tools/update_bench.py 2>&1 | grep -v "$F" >> $out/update_ab.txt
done; done
cat $out/update_ab.txt
( cd /tmp && export TMPDIR=/tmp && for v in "" nopark; do lib=$root/dpvo_amd/libdpvo_hip${v:+_$v}.so; rm -rf /tmp/ks_$v; DPVO_HIP_LIB=$lib WHICH=fused REPS=20 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -- python $root/tools/update_bench.py > /dev/null 2>&1; f=$(find /tmp/ks_$v -name "*kernel_stats.csv" | head -1); echo "== ${v:-product}" >> $out/update_kernels.txt; python $root/tools/kstats.py $f 12 >> $out/update_kernels.txt; done )
cat $out/update_kernels.txt
timeout 600 python bench.py --steps 60 --warmup 20 > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json; echo
ls -la $out
