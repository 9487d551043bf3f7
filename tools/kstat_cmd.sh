#!/bin/bash
# rocprofv3 kernel stats of a command: name (40 chars), calls, avg us, min us, max us
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -- "$@" > /dev/null 2>&1
f=$(find /tmp/ks1 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    n=re.sub(r'\(anonymous namespace\)::|void ','',r['Name'])[:44]
    print(f"{n:46s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.1f} min {float(r['MinNs'])/1e3:9.1f} max {float(r['MaxNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
PY
