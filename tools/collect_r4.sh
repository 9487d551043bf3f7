#!/bin/bash
# tools/collect_r4.sh <tag>: the round-4 profile set in one GPU call (~6 min).  Output: gpurun_out/<tag>/ (copy to profiles/).
tag=${1:-r4}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json; echo
python bench.py --no-cpu-baseline --no-ref-baseline --config fast > $out/bench_fast.json 2>> $out/bench.err
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && DPVO_BENCH_NO_BOX=1 DPVO_BENCH_NO_DROP_LEG=1 DPVO_BENCH_NO_LC_LEG=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $root/bench.py --no-cpu-baseline --no-ref-baseline > $out/bench_under_rocprof.json 2> /tmp/ks.err )
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; python tools/kstats.py $f 45 > $out/kernel_stats_short.txt
t=$(find /tmp/ks -name "*kernel_trace.csv" | head -1); python tools/frame_timeline.py $t 3 > $out/frame_timeline.txt
python tools/kernel_tail_avg.py $t corr_pyramid 20 > $out/corr_steady_state.txt
python tools/stream_stamps.py 2>&1 | grep -v amdgpu > $out/stream_stamps.txt
bash tools/pmc_corr.sh > $out/pmc_corr.log 2>&1; cp gpurun_out/corr_pmc.json $out/corr_pmc.json
CONFIG=fast bash tools/pmc_corr.sh > $out/pmc_corr_fast.log 2>&1; cp gpurun_out/corr_pmc_fast.json $out/corr_pmc_fast.json
bash tools/pmc_update.sh > $out/update_pmc_sq.txt 2>&1
bash tools/pmc_update_mem.sh > $out/update_pmc_mem.txt 2>&1
WHICH=fused python tools/update_bench.py 2>&1 | grep -v amdgpu > $out/update_bench.txt
python tools/corr_bench.py 2>&1 | grep -v amdgpu > $out/corr_bench.txt
python tools/ba_bench.py 2>&1 | grep -v amdgpu > $out/ba_bench.txt
python tools/host_time.py tottime 2>&1 | grep -v amdgpu | head -30 > $out/host_profile.txt
python -m pytest tests/test_gpu_ref.py tests/test_gpu_ref_pipeline.py tests/test_gpu_update.py -q -s -m gpu 2>&1 | grep -v "amdgpu\|Warning\|autocast" > $out/ref_parity.txt
python tools/enc_bench.py 2>&1 | grep -v amdgpu > $out/enc_bench.txt
[ -f dpvo_amd/libdpvo_hip_kft.so ] && DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_kft.so python tools/kf_trace.py 2>&1 | grep -v amdgpu > $out/kf_phases.txt
[ -f dpvo_amd/libdpvo_hip_fst.so ] && DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_fst.so python tools/fs_trace.py 2>&1 | grep -v amdgpu > $out/frame_state_workgroups.txt
[ -f dpvo_amd/libdpvo_hip_bat.so ] && DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_bat.so python tools/ba_trace.py 2>&1 | grep -v amdgpu > $out/ba_patch_workgroups.txt
[ -x tools/probes/clock_probe.bin ] && tools/probes/clock_probe.bin > $out/clock_probe.txt 2>&1
ls -la $out
# (gpurun merges at most 64 MiB back: the raw counter / trace csvs stay on the box)
rm -rf $root/gpurun_out/pmc_corr $root/gpurun_out/pmc_corr_fast $root/gpurun_out/pmc_update $root/gpurun_out/pmc_update_mem
du -sh $root/gpurun_out
