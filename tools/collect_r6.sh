#!/bin/bash
# tools/collect_r6.sh <tag>: the round-6 profile set in one GPU call (~8 min).  Output: gpurun_out/<tag>/ (copy to profiles/).
tag=${1:-r6}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 1200 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -v "$F" > $out/pytest_gpu.txt; tail -3 $out/pytest_gpu.txt
python bench.py --steps 60 --warmup 45 > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ref-baseline > $out/bench_driver_flags.json 2>> $out/bench.err
python bench.py --steps 60 --warmup 45 --config fast --no-cpu-baseline --no-ref-baseline > $out/bench_fast.json 2>> $out/bench.err
for t in 1 13 29 1 13; do python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-ref-baseline --update-tiling $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('update tiling $t: frames/sec', d['value'], 'period', d['frame_period_ms']['median'], 'update ms', d['roofline_update']['avg_ms'], 'corr ms', d['roofline']['avg_launch_ms'], 'config 5 leg', d['with_loop_closure']['frames_per_sec'], 'drop leg', d['with_keyframe_drops']['frames_per_sec'])"; done > $out/bench_tilings.txt 2>&1; cat $out/bench_tilings.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && DPVO_BENCH_NO_BOX=1 DPVO_BENCH_NO_DROP_LEG=1 DPVO_BENCH_NO_LC_LEG=1 DPVO_BENCH_NO_PROBE_LEG=1 DPVO_BENCH_NO_HOST_LEG=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $root/bench.py --steps 60 --warmup 45 --no-cpu-baseline --no-ref-baseline > $out/bench_under_rocprof.json 2> /tmp/ks.err )
f=$(find /tmp/ks -name "*kernel_stats.csv" | xargs ls -S | head -1); cp $f $out/kernel_stats.csv; python tools/kstats.py $f 45 > $out/kernel_stats_short.txt
t=$(find /tmp/ks -name "*kernel_trace.csv" | xargs ls -S | head -1); python tools/frame_timeline.py $t 3 > $out/frame_timeline.txt
python tools/kernel_tail_avg.py $t corr_pyramid 20 > $out/corr_steady_state.txt
python tools/stream_stamps.py 2>&1 | grep -v "$F" > $out/stream_stamps.txt
bash tools/pmc_update.sh > $out/update_pmc_sq.txt 2>&1
bash tools/pmc_update_mem.sh > $out/update_pmc_mem.txt 2>&1
rm -rf $root/gpurun_out/pmc_update $root/gpurun_out/pmc_update_mem
WHICH=fused python tools/update_bench.py 2>&1 | grep -v "$F" > $out/update_bench.txt
python tools/update_tilings.py 1,13,29 3 0 2>&1 | grep -v "$F" > $out/update_tilings.txt
python tools/corr_bench.py 2>&1 | grep -v "$F" > $out/corr_bench.txt
python tools/ba_bench.py 2>&1 | grep -v "$F" > $out/ba_bench.txt
python tools/gba_bench.py 2>&1 | grep -v "$F" > $out/gba_bench.txt
python tools/chol_bench.py 2>&1 | grep -v "$F" > $out/chol_bench.txt
python tools/host_time.py tottime 2>&1 | grep -v "$F" | head -30 > $out/host_profile.txt
[ -x tools/probes/clock_probe.bin ] && tools/probes/clock_probe.bin > $out/clock_probe.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
ls -la $out; du -sh $root/gpurun_out
