#!/bin/bash
# tools/kres.sh <file.hip> [extra hipcc flags]: per-kernel register / scratch / LDS table from the compiler's resource remarks (no GPU)
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root/dpvo_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres_$$.o 2>&1 | python3 "$root/tools/kres_parse.py"
rm -f /tmp/kres_$$.o
