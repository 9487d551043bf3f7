#!/bin/bash
# per-kernel register / spill summary of one .hip file: tools/kres.sh dpvo_amd/csrc/update_fused.hip [extra flags]
# (packed-FP32 ops on, as the Makefile builds every translation unit but ba*.hip)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result \
  -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/kres.o 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|VGPRs Spill|ScratchSize" \
  | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - - | sed 's/Function Name: //'
