#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...> -- <cmd...>   (one rocprofv3 --pmc pass; kernel-trace only, per gpurun rules)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$out" -o pmc -- "$@" > "$out.log" 2>&1
