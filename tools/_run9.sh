pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_update']['avg_ms'], d['with_keyframe_drops']['frames_per_sec'], d['per_rank'][0]['host_cpu_us_per_frame'])"; }
echo base; python bench.py --no-cpu-baseline 2>&1 | tail -1 | pr
for k in 0 3 5 6 7; do echo "HOLD_AT=$k"; DPVO_ENC_AFTER_UPDATE=1 DPVO_ENC_HOLD_AT=$k python bench.py --no-cpu-baseline 2>&1 | tail -1 | pr; done
echo base; python bench.py --no-cpu-baseline 2>&1 | tail -1 | pr
