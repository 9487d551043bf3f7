python -m pytest tests/test_gpu_dpvo.py tests/test_gpu_trajectory.py tests/test_gpu_frontend.py -x -q -m gpu 2>&1 | tail -3
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_update']['avg_ms'], d['with_keyframe_drops']['frames_per_sec'], d['per_rank'][0]['host_cpu_us_per_frame'])"; }
for rep in 1 2 3; do python bench.py --no-cpu-baseline 2>&1 | tail -1 | pr; done
python tools/stream_stamps.py 2>&1 | grep -v amdgpu | head -4
