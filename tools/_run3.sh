python -m pytest tests/test_gpu_dpvo.py tests/test_gpu_trajectory.py tests/test_gpu_evaluate.py tests/test_dropin.py tests/test_multiseq.py -x -q -m gpu 2>&1 | tail -5
for bs in 1 0; do
  echo "== DPVO_BLOCKING_SYNC=$bs"; DPVO_BLOCKING_SYNC=$bs python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['per_rank'], d['roofline']['avg_launch_ms'], d['roofline_update']['avg_ms'], d['with_keyframe_drops'])"
done
echo "== old path"; DPVO_FRAME_CALL=0 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['per_rank'], d['roofline']['avg_launch_ms'], d['roofline_update']['avg_ms'], d['with_keyframe_drops'])"
python tools/host_time.py tottime 2>&1 | grep -v amdgpu | head -12
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && DPVO_BENCH_NO_DROP_LEG=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/ks.out 2> /tmp/ks.err
cd $GRAFT_REPO_ROOT
t=$(find /tmp/ks -name "*kernel_trace.csv" | head -1); python tools/frame_timeline.py $t 3 | grep -v "q 2" 
