#!/bin/bash
# tools/collect_r5g.sh <tag>: second GPU call of round 5's last session -- the row kernel's B / v part (bits, times, per-wave trace against
# the old form), the config-5 tests with the mirror checks on, the leg's A/B.  Needs tools/gba_bv_ab.sh build + the two trace libraries.
tag=${1:-r5g}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 400 bash tools/gba_bv_ab.sh > $out/gba_bv_ab.txt 2>&1; grep -n "IDENTICAL\|DIFFERENT" $out/gba_bv_ab.txt
for v in gbt gbt0; do GBA_TRACE=1 GBA_SIZES=100,130 DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_$v.so timeout 200 python tools/gba_bench.py 2>&1 | grep -v "$F" > $out/gba_trace_$v.txt; done
DPVO_CHECK_MIRROR=1 timeout 600 python -m pytest tests/test_gpu_dpvo.py tests/test_gpu_ba.py tests/test_gpu_ref.py tests/test_zz_ref_pipeline.py -m gpu -q -x -k "loop or closure or global or keyframe or bookkeeping or normalize" 2>&1 | grep -v "$F" > $out/pytest_lc.txt; tail -3 $out/pytest_lc.txt
timeout 600 python tools/lc_ab.py 2 > $out/lc_ab.txt 2>&1; tail -11 $out/lc_ab.txt
ls -la $out
