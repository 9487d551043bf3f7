#!/bin/bash
# tools/collect_r5i.sh <tag>: the row kernel's XCD-major workgroup numbering against pose-major (GBA_AB_DEFS=-DGBA_XCD=0 bash tools/gba_bv_ab.sh build
# first): bits, times, the config-5 leg under both, the global-BA tests.
tag=${1:-r5i}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
GBA_AB_DEFS="-DGBA_XCD=0" GBA_SIZES=50,100,130,200,400 timeout 400 bash tools/gba_bv_ab.sh > $out/gba_xcd_ab.txt 2>&1; grep -n "IDENTICAL\|DIFFERENT" $out/gba_xcd_ab.txt; sed -n '/== times: product/,$p' $out/gba_xcd_ab.txt
LC_AB_LIB_LABEL="row kernel numbered pose-major (GBA_XCD=0)" LC_AB_ONLY="product,GBA_XCD=0 (lib" timeout 400 python tools/lc_ab.py 3 > $out/lc_ab_xcd.txt 2>&1; tail -4 $out/lc_ab_xcd.txt
timeout 300 python -m pytest tests/test_gpu_ba.py tests/test_gpu_ref.py -m gpu -q -x -k "global" 2>&1 | grep -v "$F" | tail -2
