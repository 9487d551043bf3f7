#!/bin/bash
# tools/collect_r6a.sh <tag>: round 6, first GPU call -- the flow-head scale sweep VERDICT r5 asked for (which delta_scale keeps the REFERENCE
# regular over 80 frames with a trajectory extent worth the name), step sizes of every captured BA (tests/ref_harness.py: step_ref), a baseline
# bench line of this box
tag=${1:-r6a}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
for s in 0.003 0.01 0.03 0.1 0.3; do
  timeout 240 python tools/ref_parity.py --frames 80 --scenarios T,A --attribute --delta-scale $s 2>&1 | grep -v "$F" > $out/sweep_$s.txt
  python - $out/sweep_$s.txt <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try: d = json.loads(line)
    except Exception: continue
    print(sys.argv[1].split('/')[-1], d.get("scenario"), {k: d.get(k) for k in ("int_equal_frames", "pose_max", "extent_last", "yard_max", "ba_dist_max", "step_ref_min_med_max", "ate_raw_after_terminate", "seconds")}, "above_1e3:", len(d.get("above_1e3") or []))
PY
done
timeout 300 python bench.py --steps 40 --warmup 10 > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
