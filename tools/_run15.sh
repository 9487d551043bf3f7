python -m pytest tests/test_gpu_chol.py tests/test_gpu_ba.py tests/test_gpu_ref.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_dpvo.py -x -q -m gpu -k "loop_closure or global" 2>&1 | tail -3
python tools/gba_bench.py 2>&1 | grep -v amdgpu
