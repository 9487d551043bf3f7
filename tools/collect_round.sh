#!/bin/bash
# tools/collect_round.sh <tag>: everything profiles/ keeps for a round, in one GPU call (~3 min): bench line, the same under
# rocprofv3 --kernel-trace --stats (per-kernel summary), PMC passes of the correlation kernel and of the update operator, frame
# phases, frame timeline, per-workgroup timeline of the fused update kernels.  Output: gpurun_out/<tag>/ (copy to profiles/).
tag=${1:-round}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.json; echo
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $root/bench.py > $out/bench_under_rocprof.json 2> /tmp/ks.err )
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; python tools/kstats.py $f 45 > $out/kernel_stats_short.txt
t=$(find /tmp/ks -name "*kernel_trace.csv" | head -1); python tools/frame_timeline.py $t 3 > $out/frame_timeline.txt
python tools/kernel_tail_avg.py $t corr_pyramid 20 > $out/corr_steady_state.txt
bash tools/pmc_corr.sh > $out/pmc_corr.log 2>&1; cp gpurun_out/corr_pmc.json $out/corr_pmc.json
bash tools/pmc_update.sh > $out/update_pmc.txt 2>&1
python tools/phase_times.py 2>&1 | grep -v amdgpu > $out/phases.txt
if [ -f dpvo_amd/libdpvo_hip_trace.so ]; then DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_trace.so MODE=seven python tools/fu_trace.py 2>&1 | grep -v amdgpu > $out/trace_seven.txt; fi
WHICH=both python tools/update_bench.py 2>&1 | grep -v amdgpu > $out/update_bench.txt
python tools/ba_bench.py 2>&1 | grep -v amdgpu > $out/ba_bench.txt
ls -la $out
