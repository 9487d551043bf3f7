#!/bin/bash
# tools/collect_r5d.sh <tag>: round 5, fourth GPU call -- full suite after the plan's group kernel / global-BA index changes, config-5 leg
tag=${1:-r5d}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -15 > $out/pytest_gpu.txt; tail -5 $out/pytest_gpu.txt
LC_SYNC=1 timeout 300 python tools/lc_profile.py 2>&1 | grep -v "$F" | head -52 > $out/lc_profile_sync.txt; grep "global-BA frames" $out/lc_profile_sync.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lc && LC_SYNC=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lc -- python $root/tools/lc_profile.py > $out/lc_profile.txt 2>&1
  t=$(find /tmp/lc -name "*kernel_trace.csv" | head -1); python $root/tools/lc_timeline.py $t 2 > $out/lc_timeline.txt 2>&1 )
tail -2 $out/lc_timeline.txt
python tools/gba_bench.py 2>&1 | grep -v "$F" > $out/gba_bench.txt; cat $out/gba_bench.txt
ls -la $out
