for st in 0 1; do for rep in 1 2; do DPVO_CORR_STAGED=$st python tools/corr_bench.py 2>&1 | grep -v amdgpu; done; done
DPVO_CORR_OCC=2 python tools/corr_bench.py 2>&1 | grep -v amdgpu
DPVO_CORR_STAGED=1 python -m pytest tests/test_gpu_corr.py tests/test_gpu_ref.py -x -q -m gpu 2>&1 | tail -3
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_update']['avg_ms'], d['with_keyframe_drops']['frames_per_sec'])"; }
for st in 0 1 0 1; do echo "STAGED=$st"; DPVO_CORR_STAGED=$st python bench.py --no-cpu-baseline 2>&1 | tail -1 | pr; done
