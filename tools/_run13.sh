python tools/_dbg_pm2.py 2>&1 | grep -v amdgpu
MODE=pm2 python tools/fu_trace.py 2>&1 | grep -v amdgpu | tail -22
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks2 && WHICH=pm2 REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python $GRAFT_REPO_ROOT/tools/update_bench.py > /tmp/ks2.out 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/ks2 -name "*kernel_stats.csv" | head -1); python tools/kstats.py $f 8
