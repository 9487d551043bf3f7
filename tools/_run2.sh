export DPVO_BENCH_NO_DROP_LEG=1
for bs in 1 0; do
  echo "== DPVO_BLOCKING_SYNC=$bs"; DPVO_BLOCKING_SYNC=$bs python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['per_rank'], d['roofline']['avg_launch_ms'], d['roofline_update']['avg_ms'])"
done
echo "== old path"; DPVO_FRAME_CALL=0 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['per_rank'], d['roofline']['avg_launch_ms'], d['roofline_update']['avg_ms'])"
python tools/host_time.py tottime 2>&1 | grep -v amdgpu | head -40
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/ks.out 2> /tmp/ks.err
cd $GRAFT_REPO_ROOT
t=$(find /tmp/ks -name "*kernel_trace.csv" | head -1); python tools/frame_timeline.py $t 3
