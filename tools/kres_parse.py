"""stdin: hipcc -Rpass-analysis=kernel-resource-usage remarks -> one line per kernel (tools/kres.sh)"""
import re
import subprocess
import sys

rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line) or re.search(r"remark: [^ ]+ +Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"(SGPRs Spill|VGPRs Spill|SGPRs|VGPRs|AGPRs|ScratchSize|Occupancy|LDS Size)[^:]*: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
for r in rows:
    n = r["name"]
    try:
        n = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        pass
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    print(f"{n[:56]:56s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', -1):4d} scratch {r.get('ScratchSize', -1):5d} B/lane "
          f"spillV {r.get('VGPRs Spill', -1):4d} occ {r.get('Occupancy', -1)} LDS {r.get('LDS Size', -1)}")
