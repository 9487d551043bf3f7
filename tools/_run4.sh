cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && DPVO_BENCH_NO_DROP_LEG=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/ks.out 2> /tmp/ks.err
cd $GRAFT_REPO_ROOT
tail -1 /tmp/ks.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
t=$(find /tmp/ks -name "*kernel_trace.csv" | head -1); python tools/frame_timeline.py $t 3 | cut -c1-110
echo ---- next frame; python tools/frame_timeline.py $t 2 | head -12 | cut -c1-110
