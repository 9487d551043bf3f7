#!/bin/bash
# tools/collect_final.sh <tag>: the full GPU suite at HEAD, smoke(), and a default bench line, as the driver runs them at the end of a round
tag=${1:-final}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
git -C $root rev-parse HEAD > $out/head.txt 2>/dev/null
timeout 900 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -v "$F" > $out/pytest_gpu.txt; tail -3 $out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "$F" | tail -3 > $out/smoke.txt; cat $out/smoke.txt
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json; echo
