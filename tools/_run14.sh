python tools/_dbg_pm2.py 2>&1 | grep -v amdgpu
for r in 1 2; do WHICH=pm2 REPS=30 python tools/update_bench.py 2>&1 | grep -E "pm2"; WHICH=fused REPS=30 python tools/update_bench.py 2>&1 | grep -E "fused"; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks2 && WHICH=pm2 REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python $GRAFT_REPO_ROOT/tools/update_bench.py > /tmp/ks2.out 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/ks2 -name "*kernel_stats.csv" | head -1); python tools/kstats.py $f 6
