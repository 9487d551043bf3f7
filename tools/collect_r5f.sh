#!/bin/bash
# tools/collect_r5f.sh <tag>: config-5 global-BA frame (host-synchronous per-frame times, kernel timeline under rocprofv3), gba_bench, and
# the headline frame's kernel timeline
tag=${1:-r5f}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
LC_SYNC=1 timeout 300 python tools/lc_profile.py 2>&1 | grep -v "$F" | head -100 > $out/lc_profile_sync.txt; grep "global-BA frames" $out/lc_profile_sync.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lc && LC_SYNC=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lc -- python $root/tools/lc_profile.py > $out/lc_profile.txt 2>&1
  t=$(find /tmp/lc -name "*kernel_trace.csv" | head -1); python $root/tools/lc_timeline.py $t 2 > $out/lc_timeline.txt 2>&1 )
tail -2 $out/lc_timeline.txt
python tools/gba_bench.py 2>&1 | grep -v "$F" > $out/gba_bench.txt; cat $out/gba_bench.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ft && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -- python $root/bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-ref-baseline > $out/bench_prof.json 2> /dev/null
  t=$(find /tmp/ft -name "*kernel_trace.csv" | head -1); python $root/tools/frame_timeline.py $t > $out/frame_timeline.txt 2>&1 )
head -60 $out/frame_timeline.txt
ls -la $out
