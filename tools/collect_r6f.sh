#!/bin/bash
# tools/collect_r6f.sh <tag>: full GPU suite after the environment-switch cleanup, the PMC passes of the new correlation kernel (default + fast), bench
tag=${1:-r6f}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -60 > $out/pytest_gpu.txt; tail -5 $out/pytest_gpu.txt
timeout 600 bash tools/pmc_corr.sh > $out/pmc_corr.log 2>&1; tail -2 $out/pmc_corr.log | cut -c1-400
CONFIG=fast timeout 600 bash tools/pmc_corr.sh > $out/pmc_corr_fast.log 2>&1; tail -2 $out/pmc_corr_fast.log | cut -c1-400
cp gpurun_out/corr_pmc.json gpurun_out/corr_pmc_fast.json $out/ 2>/dev/null
timeout 300 python bench.py --steps 40 --warmup 10 > $out/bench.json 2> $out/bench.err; python - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("frames/sec", d["value"], "period", d["frame_period_ms"]["median"], "update ms", d["roofline_update"]["avg_ms"], "corr", d["roofline"]["avg_launch_ms"], "lc", d["with_loop_closure"]["frames_per_sec"], "drops", d["with_keyframe_drops"]["frames_per_sec"], "host", d["per_rank"][0]["host_cpu_us_per_frame"])
PY
