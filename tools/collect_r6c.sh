#!/bin/bash
# tools/collect_r6c.sh <tag>: round 6, third GPU call -- the new parity tests (step-relative bounds, mid scale, negative control, fast.yaml,
# the reference's Python on libdpvo_hip.so through dpvo_amd/integration_stubs.py) and the phase traces of the 4-wave / 12-wave update kernels
tag=${1:-r6c}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 600 python -m pytest tests/test_gpu_integration_stubs.py tests/test_zz_ref_pipeline.py -q -x -s 2>&1 | grep -v "$F" > $out/pytest_parity.txt; tail -30 $out/pytest_parity.txt | cut -c1-600
for t in 0 12 28; do
  TILING=$t timeout 120 python tools/fu_trace.py 2>&1 | grep -v "$F" > $out/fu_trace_tiling_$t.txt
done
cat $out/fu_trace_tiling_12.txt
