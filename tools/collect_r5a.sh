#!/bin/bash
# tools/collect_r5a.sh <tag>: round 5, first GPU call -- the full GPU suite (no -x: every failure is seen), the search for the
# well-conditioned tracker-level scenario (tests/test_zz_ref_pipeline.py: WELL) and a bench line.  Output: gpurun_out/<tag>/.
tag=${1:-r5a}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "$F" > $out/pytest_gpu.txt; tail -5 $out/pytest_gpu.txt
for v in "0.1 0.3,-0.2" "0.0 0.3,-0.2" "0.03 0.1,-0.07" "0.1 1.0,-0.7" "0.3 1.0,-0.7"; do
  set -- $v
  timeout 300 python tools/ref_parity.py --frames 80 --scenarios A,T --attribute --delta-scale $1 --delta-bias=$2 2>&1 | grep -v "$F" > $out/well_$1_$2.txt
done
timeout 300 python tools/ref_parity.py --frames 80 --scenarios A,T --attribute --delta-scale 0.01 2>&1 | grep -v "$F" > $out/well_0.01_none.txt
timeout 600 python bench.py --steps 60 --warmup 20 > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json; echo
ls -la $out
