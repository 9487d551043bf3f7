#!/bin/bash
# tools/collect_r5z.sh <tag>: the end-of-round set of round 5's last session in one GPU call (~7 min): the full GPU suite, the bench line
# with all its legs and baselines, the same run under rocprofv3 (kernel stats, one frame's timeline), the config-5 leg's A/B and the
# global-BA frame's timeline, global BA / Cholesky sweeps.  Output: gpurun_out/<tag>/ (copy to profiles/).
tag=${1:-r5z}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 900 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -v "$F" > $out/pytest_gpu.txt; tail -3 $out/pytest_gpu.txt
python bench.py --steps 60 --warmup 45 > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.json; echo
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && DPVO_BENCH_NO_BOX=1 DPVO_BENCH_NO_DROP_LEG=1 DPVO_BENCH_NO_LC_LEG=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $root/bench.py --steps 60 --warmup 45 --no-cpu-baseline --no-ref-baseline > $out/bench_under_rocprof.json 2> /tmp/ks.err )
f=$(find /tmp/ks -name "*kernel_stats.csv" | xargs ls -S | head -1); cp $f $out/kernel_stats.csv; python tools/kstats.py $f 45 > $out/kernel_stats_short.txt
t=$(find /tmp/ks -name "*kernel_trace.csv" | xargs ls -S | head -1); python tools/frame_timeline.py $t 3 > $out/frame_timeline.txt
python tools/kernel_tail_avg.py $t corr_pyramid 20 > $out/corr_steady_state.txt
timeout 420 python tools/lc_ab.py 1 > $out/lc_ab.txt 2>&1; tail -9 $out/lc_ab.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lc && LC_SYNC=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lc -- python $root/tools/lc_profile.py > $out/lc_profile.txt 2>&1
  t=$(find /tmp/lc -name "*kernel_trace.csv" | xargs ls -S | head -1); python $root/tools/lc_timeline.py $t 2 > $out/lc_timeline.txt 2>&1 )
tail -1 $out/lc_timeline.txt
timeout 200 python tools/gba_bench.py 2>&1 | grep -v "$F" > $out/gba_bench.txt; tail -6 $out/gba_bench.txt
timeout 200 python tools/chol_bench.py 2>&1 | grep -v "$F" > $out/chol_bench.txt
timeout 100 python tools/gba_bits.py 2>&1 | grep -v "$F" > $out/gba_bits.txt
[ -x tools/probes/clock_probe.bin ] && tools/probes/clock_probe.bin > $out/clock_probe.txt 2>&1
ls -la $out
