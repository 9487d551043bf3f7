#!/bin/bash
# tools/zz_repeats.sh <tag> [n=3]: the tracker-level checker (tests/test_zz_ref_pipeline.py: the reference's own tracker in lock step) n times in one call
tag=${1:-zz}; n=${2:-3}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
for i in $(seq 1 $n); do
  timeout 400 python -m pytest tests/test_zz_ref_pipeline.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -2 | tee -a $out/zz_repeats.txt
done
[ -x tools/probes/clock_probe.bin ] && tools/probes/clock_probe.bin 2>&1 | tail -2 >> $out/zz_repeats.txt
