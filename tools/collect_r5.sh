#!/bin/bash
# tools/collect_r5.sh <tag>: the round-5 profile set in one GPU call (~8 min).  Output: gpurun_out/<tag>/ (copy to profiles/).
tag=${1:-r5}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 1200 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -v "$F" > $out/pytest_gpu.txt; tail -3 $out/pytest_gpu.txt
python bench.py --steps 60 --warmup 45 > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json; echo
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && DPVO_BENCH_NO_BOX=1 DPVO_BENCH_NO_DROP_LEG=1 DPVO_BENCH_NO_LC_LEG=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $root/bench.py --steps 60 --warmup 45 --no-cpu-baseline --no-ref-baseline > $out/bench_under_rocprof.json 2> /tmp/ks.err )
f=$(find /tmp/ks -name "*kernel_stats.csv" | xargs ls -S | head -1); cp $f $out/kernel_stats.csv; python tools/kstats.py $f 45 > $out/kernel_stats_short.txt
t=$(find /tmp/ks -name "*kernel_trace.csv" | xargs ls -S | head -1); python tools/frame_timeline.py $t 3 > $out/frame_timeline.txt
python tools/kernel_tail_avg.py $t corr_pyramid 20 > $out/corr_steady_state.txt
python tools/stream_stamps.py 2>&1 | grep -v "$F" > $out/stream_stamps.txt
bash tools/pmc_corr.sh > $out/pmc_corr.log 2>&1; cp gpurun_out/corr_pmc.json $out/corr_pmc.json
bash tools/pmc_update.sh > $out/update_pmc_sq.txt 2>&1
bash tools/pmc_update_mem.sh > $out/update_pmc_mem.txt 2>&1
WHICH=fused python tools/update_bench.py 2>&1 | grep -v "$F" > $out/update_bench.txt
python tools/corr_bench.py 2>&1 | grep -v "$F" > $out/corr_bench.txt
python tools/ba_bench.py 2>&1 | grep -v "$F" > $out/ba_bench.txt
python tools/gba_bench.py 2>&1 | grep -v "$F" > $out/gba_bench.txt
python tools/chol_bench.py 2>&1 | grep -v "$F" > $out/chol_bench.txt
python tools/host_time.py tottime 2>&1 | grep -v "$F" | head -30 > $out/host_profile.txt
LC_SYNC=1 timeout 300 python tools/lc_profile.py 2>&1 | grep -v "$F" | head -52 > $out/lc_profile_sync.txt; grep "global-BA frames" $out/lc_profile_sync.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lc && LC_SYNC=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lc -- python $root/tools/lc_profile.py > $out/lc_profile.txt 2>&1
  t=$(find /tmp/lc -name "*kernel_trace.csv" | xargs ls -S | head -1); python $root/tools/lc_timeline.py $t 2 > $out/lc_timeline.txt 2>&1 )
tail -1 $out/lc_timeline.txt
[ -x tools/probes/clock_probe.bin ] && tools/probes/clock_probe.bin > $out/clock_probe.txt 2>&1
ls -la $out
rm -rf $root/gpurun_out/pmc_corr $root/gpurun_out/pmc_update $root/gpurun_out/pmc_update_mem
du -sh $root/gpurun_out
