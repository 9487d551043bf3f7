#!/bin/bash
# rocprofv3 --pmc passes over the bench for corr_pyramid_kernel's shader-side counters (wait share, VALU / MFMA instruction
# counts), one group per pass, kernel-trace only; the bench runs with a sync per frame (see tools/pmc_corr.sh).
root=$(pwd); out=$root/gpurun_out/pmc_corr_sq; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  (cd $root && DPVO_BENCH_SYNC_EVERY_FRAME=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o pmc -- python bench.py --steps 8 --warmup 45 --no-cpu-baseline > $out/g$i.log 2>&1)
  f=$(find $out/g$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (cd $root && python tools/pmc_summary.py $f corr_pyramid > $out/g$i.txt 2>&1); echo "== $grp"; cat $out/g$i.txt; else echo "== $grp: no counter file"; tail -3 $out/g$i.log; fi
done
