#!/bin/bash
# tools/collect_r5j.sh <tag>: the round's last GPU call -- the full GPU suite at the final code, then the scatter + patch launch and update()'s
# reproject / corr / plan order against their switches (GBA_AB_DEFS=-DGBA_FUSE_SP=0 bash tools/gba_bv_ab.sh build first): bits, the config-5 leg
tag=${1:-r5j}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
F='amdgpu\|Warning\|autocast\|warnings.warn'
timeout 200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -4 > $out/pytest_gpu.txt; tail -2 $out/pytest_gpu.txt
timeout 40 python tools/gba_bits.py 50,100 2>&1 | grep -v "$F" > $out/bits_product.txt
DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_bv0.so timeout 40 python tools/gba_bits.py 50,100 2>&1 | grep -v "$F" > $out/bits_two_launches.txt
cmp $out/bits_product.txt $out/bits_two_launches.txt && echo BIT-IDENTICAL | tee -a $out/bits_product.txt || echo DIFFERENT | tee -a $out/bits_product.txt
head -2 $out/bits_product.txt
LC_AB_LIB_LABEL="scatter and patch as two launches (GBA_FUSE_SP=0)" LC_AB_ONLY="product,PLAN_FIRST=1),(libdpvo_hip_bv0.so)" timeout 100 python tools/lc_ab.py 2 > $out/lc_ab.txt 2>&1; tail -5 $out/lc_ab.txt
