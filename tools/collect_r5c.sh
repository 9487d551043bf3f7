#!/bin/bash
# tools/collect_r5c.sh <tag>: round 5, third GPU call -- the tracker-level checker twice (repeatability), a bench line with K7 parked,
# and the kernel timeline of a global-BA frame of BASELINE config 5.
tag=${1:-r5c}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
for rep in 1 2; do
  timeout 900 python -m pytest tests/test_zz_ref_pipeline.py -m gpu -q -s 2>&1 | grep -v "$F" > $out/pytest_zz_$rep.txt; tail -4 $out/pytest_zz_$rep.txt
done
timeout 600 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-ref-baseline > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json; echo
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lc && LC_SYNC=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lc -- python $root/tools/lc_profile.py > $out/lc_profile.txt 2>&1
  t=$(find /tmp/lc -name "*kernel_trace.csv" | head -1); python $root/tools/lc_timeline.py $t 2 > $out/lc_timeline.txt 2>&1 )
LC_SYNC=1 timeout 300 python tools/lc_profile.py 2>&1 | grep -v "$F" | head -60 > $out/lc_profile_sync.txt
tail -3 $out/lc_timeline.txt
ls -la $out
