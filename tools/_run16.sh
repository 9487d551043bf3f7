cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks3 -- python $GRAFT_REPO_ROOT/tools/gba_bench.py > /tmp/ks3.out 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/ks3 -name "*kernel_stats.csv" | head -1); python tools/kstats.py $f 10
