#!/bin/bash
# repeated default-bench runs: frames/sec, corr kernel time, state fingerprint and buffer addresses -- to see what the speed
# regimes correlate with
cd "${GRAFT_REPO_ROOT:-.}"
for r in $(seq 1 ${1:-8}); do
  DPVO_BENCH_DIAG=1 timeout 200 python bench.py --no-cpu-baseline --steps ${2:-400} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], json.dumps(d['diag']))"
done
