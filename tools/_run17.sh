mkdir -p gpurun_out/r3
( for i in $(seq 1 12); do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -2 | head -1; sleep 0.1; done > gpurun_out/r3/smi_during_probe.txt ) &
timeout 60 tools/probes/clock_probe.bin | tee gpurun_out/r3/clock_probe.txt
wait
timeout 60 tools/probes/clock_probe.bin --json
python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r3/bench_quick.json; python -c "
import json; d=json.load(open('gpurun_out/r3/bench_quick.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['traffic_source'], d['roofline_update']['avg_ms'], d['with_keyframe_drops'], d['per_rank'], d['box'])"
