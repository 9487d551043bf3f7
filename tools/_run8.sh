python -m pytest tests/test_gpu_encoders.py tests/test_gpu_update.py -x -q -m gpu 2>&1 | tail -3
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_update']['avg_ms'], d['with_keyframe_drops']['frames_per_sec'], d['per_rank'][0]['host_cpu_us_per_frame'])"; }
for rep in 1 2 3; do python bench.py --no-cpu-baseline 2>&1 | tail -1 | pr; done
bash tools/_run4.sh 2>&1 | grep "q 2\|frame_state\|k7_gru\|copyBuffer" | head -24
