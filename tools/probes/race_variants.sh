#!/bin/bash
# tools/probes/race_variants.sh out.log name...: ba_race3.py (copy mode) once per dpvo_amd/libdpvo_hip_<name>.so
out=$1; shift; mkdir -p $(dirname $out); : > $out
for v in "$@"; do
  echo -n "$v: " >> $out
  DPVO_HIP_LIB=$PWD/dpvo_amd/libdpvo_hip_$v.so REPS=${REPS:-300} timeout 200 python tools/ba_race3.py copy 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
