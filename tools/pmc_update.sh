#!/bin/bash
# rocprofv3 --pmc passes over the update operator alone (tools/update_bench.py, fused path): MFMA instruction counters and busy
# cycles per kernel, one counter group per pass (kernel-trace only, as the GPU pool requires for counter collection).
# Writes gpurun_out/pmc_update/<group>.txt (per-kernel means); copy the summaries to profiles/.
root=$(pwd); out=$root/gpurun_out/pmc_update; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  (cd $root && WHICH=fused REPS=5 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o pmc -- python tools/update_bench.py > $out/g$i.log 2>&1)
  f=$(find $out/g$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (cd $root && python tools/pmc_summary.py $f fu > $out/g$i.txt 2>&1); echo "== $grp"; cat $out/g$i.txt; else echo "== $grp: no counter file"; tail -3 $out/g$i.log; fi
done
