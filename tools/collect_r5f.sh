#!/bin/bash
# tools/collect_r5f.sh <tag>: the config-5 (LOOP_CLOSURE) work of round 5's last session in one GPU call (~7 min): full GPU suite, the
# A/B of the leg against its measurement switches, a bench line, the global-BA frame's kernel timeline, global BA / Cholesky sweeps.
tag=${1:-r5f}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" > $out/pytest_gpu.txt; tail -4 $out/pytest_gpu.txt
timeout 420 python tools/lc_ab.py 2 > $out/lc_ab.txt 2>&1; tail -9 $out/lc_ab.txt
timeout 300 python bench.py --steps 60 --warmup 45 --no-cpu-baseline --no-ref-baseline > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json; echo
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lc && LC_SYNC=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lc -- python $root/tools/lc_profile.py > $out/lc_profile.txt 2>&1
  t=$(find /tmp/lc -name "*kernel_trace.csv" | xargs ls -S | head -1); python $root/tools/lc_timeline.py $t 2 > $out/lc_timeline.txt 2>&1 )
tail -1 $out/lc_timeline.txt
timeout 200 python tools/gba_bench.py 2>&1 | grep -v "$F" > $out/gba_bench.txt; tail -8 $out/gba_bench.txt
timeout 200 python tools/chol_bench.py 2>&1 | grep -v "$F" > $out/chol_bench.txt
LC_SYNC=1 timeout 300 python tools/lc_profile.py 2>&1 | grep -v "$F" | head -90 > $out/lc_profile_sync.txt
ls -la $out
