#!/bin/bash
# tools/collect_r6b.sh <tag>: round 6, second GPU call -- the 12-wave geometry of the update kernels (three waves per SIMD): tests, the
# operator alone per tiling, per-kernel times under rocprofv3, the frame (bench.py) per tiling; and the reference against ITSELF free
# running at the swept flow-head scales (the yard-stick for profiles/r06_a_delta_scale_sweep.txt)
tag=${1:-r6b}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 300 python -m pytest tests/test_gpu_update.py tests/test_golden.py -q -x 2>&1 | grep -v "$F" | tail -15 > $out/pytest_update.txt; tail -3 $out/pytest_update.txt
timeout 300 python tools/update_tilings.py 1,5,9,13,29,28,17 3 2>&1 | grep -v "$F" > $out/update_tilings.txt; cat $out/update_tilings.txt
for t in 1 29; do
  TILING=$t WHICH=fused REPS=10 timeout 200 bash tools/kstat_cmd.sh python $root/tools/update_bench.py > $out/update_kernels_tiling_$t.txt 2>&1; cat $out/update_kernels_tiling_$t.txt | head -9
done
cd $root
for t in 1 13 29 1 13; do
  timeout 300 python bench.py --steps 40 --warmup 10 --update-tiling $t > $out/bench_$t.json 2> $out/bench_$t.err
  python - $out/bench_$t.json $t <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tiling", sys.argv[2], "frames/sec", d["value"], "period", d["frame_period_ms"]["median"], "update ms", d["roofline_update"]["avg_ms"], "corr", d["roofline"]["avg_launch_ms"], "lc", d["with_loop_closure"]["frames_per_sec"], "drops", d["with_keyframe_drops"]["frames_per_sec"])
PY
done 2>&1 | tee $out/bench_tilings.txt
for s in 0.003 0.01 0.03; do
  timeout 200 python tools/ref_parity.py --frames 80 --scenarios R,A --delta-scale $s 2>&1 | grep -v "$F" > $out/refref_$s.txt
  python - $out/refref_$s.txt <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try: d = json.loads(line)
    except Exception: continue
    print(sys.argv[1].split('/')[-1], d.get("scenario"), {k: d.get(k) for k in ("int_equal_frames", "pose_max", "extent_last", "ate_raw_after_terminate", "pose_series")})
PY
done 2>&1 | tee $out/refref.txt
