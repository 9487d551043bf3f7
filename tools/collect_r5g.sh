#!/bin/bash
# tools/collect_r5g.sh <tag>: is the update operator slower under rocprofv3 in bench.py (K1 180 / K7 320 us in r5f against 115 / 153 in
# update_bench under the same profiler), or was that the box?  One box: bench unprofiled, profiled, unprofiled again, and with the nopark
# K7; then the tracker-level checker three times (repeatability on this box).
tag=${1:-r5g}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
B="--steps 40 --warmup 20 --no-cpu-baseline --no-ref-baseline"
pick() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2], d['value'], 'update', d['roofline_update']['avg_ms'], 'corr', d['roofline']['avg_launch_ms'], 'lc', (d.get('with_loop_closure') or {}).get('frames_per_sec'), d['box']['mfma_clock_ghz_256_cus'])" $1 $2; }
timeout 300 python bench.py $B > $out/bench_1.json 2> /dev/null; pick $out/bench_1.json unprofiled_1 | tee -a $out/summary.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ft && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ft -- python $root/bench.py $B > $out/bench_prof.json 2> /dev/null
  f=$(find /tmp/ft -name "*kernel_stats.csv" | head -1); python $root/tools/kstats.py $f 14 > $out/bench_prof_kernels.txt )
pick $out/bench_prof.json profiled | tee -a $out/summary.txt; head -8 $out/bench_prof_kernels.txt
timeout 300 python bench.py $B > $out/bench_2.json 2> /dev/null; pick $out/bench_2.json unprofiled_2 | tee -a $out/summary.txt
DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_nopark.so timeout 300 python bench.py $B > $out/bench_nopark.json 2> /dev/null; pick $out/bench_nopark.json nopark | tee -a $out/summary.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ft2 && DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_nopark.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ft2 -- python $root/bench.py $B > $out/bench_nopark_prof.json 2> /dev/null
  f=$(find /tmp/ft2 -name "*kernel_stats.csv" | head -1); python $root/tools/kstats.py $f 8 > $out/bench_nopark_prof_kernels.txt )
pick $out/bench_nopark_prof.json nopark_profiled | tee -a $out/summary.txt; head -8 $out/bench_nopark_prof_kernels.txt
for rep in 1 2 3; do
  timeout 600 python -m pytest tests/test_zz_ref_pipeline.py -m gpu -q 2>&1 | grep -v "$F" | tail -2 > $out/pytest_zz_$rep.txt; tail -1 $out/pytest_zz_$rep.txt
done
