#!/usr/bin/env python
"""Build variants of one translation unit from EDITED device assembly (dev tool, runs in the build container).

    python tools/probes/asm_variants.py ba.hip "<extra hipcc flags>" out_dir  name=rule [name=rule ...]

hipcc -save-temps leaves the gfx950 assembly and the host assembly (with the device image as one .asciz blob); each
variant re-assembles an edited copy of the device assembly, links and bundles it the way hipcc does and swaps the blob in the
host assembly for an .incbin of the new bundle, then links dpvo_amd/libdpvo_hip_<name>.so from the other units' objects.
Rules edit only the body of the kernel named by KERNEL (default ba_pair_kernel):
    none                unchanged (control)
    after:<re>:<n>      s_nop <n> after every instruction matching <re>
    before:<re>:<n>     s_nop <n> before every instruction matching <re>
    range:<a>:<b>:<re>:<n>   as `after`, only for matching instructions number a..b-1 (bisection)
    unpack:<re>[:<a>:<b>]    every packed-FP32 instruction matching <re> (number a..b-1 of them) replaced by its two scalar
                             halves, computed into two spare registers and then moved (safe for any operand overlap)
"""
import os, re, subprocess, sys
LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(HERE, "dpvo_amd", "csrc")
unit, flags, out = sys.argv[1:4]
rules = dict(a.split("=", 1) for a in sys.argv[4:])
kernel = os.environ.get("KERNEL", "ba_pair_kernel")
stem = unit.split(".")[0]
os.makedirs(out, exist_ok=True)
run = lambda c, **k: subprocess.run(c, shell=True, check=True, cwd=out, **k)
run(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result {flags} -I{CSRC} -c {CSRC}/{unit} -o {stem}.o -save-temps 2>/dev/null")
dev = open(f"{out}/{stem}-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
host = open(f"{out}/{stem}-host-x86_64-unknown-linux-gnu.s").read().split("\n")
blob = [i for i, l in enumerate(host) if l.startswith("\t.asciz\t\"__CLANG_OFFLOAD_BUNDLE__")]
assert len(blob) == 1
start = next(i for i, l in enumerate(dev) if re.match(rf"^_Z\w*{kernel}\w*:", l))
end = next(i for i in range(start, len(dev)) if "s_endpgm" in dev[i])
print(f"{kernel}: device assembly lines {start}..{end}")
others = [o for o in "corr geom graph update update_fused ba ba_global frontend encoder capi".split() if o != stem]
def half(op, hi):
    m = re.match(r"^([vs])\[(\d+):(\d+)\]$", op)
    return f"{m.group(1)}{int(m.group(2)) + hi}" if m else op


def unpack(ins, t0, t1):
    """v_pk_{mul,add,fma}_f32 / v_pk_mov_b32 -> two VOP3 halves (op_sel / op_sel_hi / neg_lo / neg_hi honoured)."""
    m = re.match(r"^v_pk_(mul_f32|add_f32|fma_f32|mov_b32)\s+(.*)$", ins)
    op, rest = m.group(1), m.group(2)
    mods = dict((k, [int(x) for x in v.split(",")]) for k, v in re.findall(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([\d,]+)\]", rest))
    ops = [o.strip() for o in re.sub(r"\s*(op_sel|op_sel_hi|neg_lo|neg_hi):\[[\d,]+\]", "", rest).split(",")]
    dst, src = ops[0], ops[1:]
    n = len(src)
    sel_lo = mods.get("op_sel", [0] * n); sel_hi = mods.get("op_sel_hi", [1] * n)
    neg_lo = mods.get("neg_lo", [0] * n); neg_hi = mods.get("neg_hi", [0] * n)
    out = []
    if op == "mov_b32":          # D.lo = S0[op_sel[0]], D.hi = S1[op_sel[1]]
        out.append(f"\tv_mov_b32_e32 {t0}, {half(src[0], sel_lo[0])}")
        out.append(f"\tv_mov_b32_e32 {t1}, {half(src[1], sel_lo[1])}")
    else:
        mn = {"mul_f32": "v_mul_f32_e64", "add_f32": "v_add_f32_e64", "fma_f32": "v_fma_f32"}[op]
        for t, sel, neg in ((t0, sel_lo, neg_lo), (t1, sel_hi, neg_hi)):
            a = [("-" if neg[i] else "") + half(src[i], sel[i]) for i in range(n)]
            out.append(f"\t{mn} {t}, " + ", ".join(a))
    out.append(f"\tv_mov_b32_e32 {half(dst, 0)}, {t0}")
    out.append(f"\tv_mov_b32_e32 {half(dst, 1)}, {t1}")
    return out


for name, rule in rules.items():
    parts = rule.split(":")
    if parts[0] in ("unpack", "unpackx"):      # unpackx: all matching instructions EXCEPT number a..b-1
        nv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", "\n".join(dev[end:end + 200])).group(1))
        # two spare VGPRs below the AGPR block: the kernel's accum_offset is a multiple of 4 >= its VGPR count
        ao = int(re.search(r"\.amdhsa_accum_offset (\d+)", "\n".join(dev[end:end + 200])).group(1))
        vmax = max(int(x) for l in dev[start:end] for x in re.findall(r"\bv\[?(?:\d+:)?(\d+)\]?", l))
        assert vmax + 2 < ao, (vmax, ao)
        t0, t1 = f"v{vmax + 1}", f"v{vmax + 2}"
        if len(parts) > 3 and parts[-1].isdigit() and parts[-2].isdigit():
            lo, hi, parts = int(parts[-2]), int(parts[-1]), [parts[0], ":".join(parts[1:-2])]
        else:
            lo, hi, parts = 0, 1 << 30, [parts[0], ":".join(parts[1:])]
        lines, hits, seen = list(dev[:start]), 0, 0
        for l in dev[start:end + 1]:
            ins = l.strip()
            if l.startswith("\t") and ins.startswith("v_pk_") and re.search(parts[1], ins):
                if (lo <= seen < hi) != (rule.startswith("unpackx")):
                    lines.append("\t; unpacked: " + ins); lines += unpack(ins, t0, t1); hits += 1
                else:
                    lines.append(l)
                seen += 1
            else:
                lines.append(l)
        lines += dev[end + 1:]
        parts = ["done"]
    if parts[0] != "done":
        lines, hits, seen = list(dev[:start]), 0, 0
    for l in (dev[start:end + 1] if parts[0] != "done" else []):
        ins = l.strip()
        isins = l.startswith("\t") and not ins.startswith((".", ";"))
        if parts[0] == "none" or not isins:
            lines.append(l); continue
        if parts[0] in ("after", "before"):
            m = re.search(parts[1], ins) is not None
        else:
            m = re.search(parts[3], ins) is not None
            if m:
                m = int(parts[1]) <= seen < int(parts[2]); seen += 1
        n = int(parts[-1])
        if m and parts[0] == "before": lines.append(f"\ts_nop {n}")
        lines.append(l)
        if m and parts[0] != "before": lines.append(f"\ts_nop {n}")
        hits += bool(m)
    if parts[0] != "done":
        lines += dev[end + 1:]
    open(f"{out}/dev_{name}.s", "w").write("\n".join(lines))
    run(f"{LLVM}/clang -cc1as -triple amdgcn-amd-amdhsa -filetype obj -target-cpu gfx950 -mrelocation-model pic -o dev_{name}.o dev_{name}.s")
    run(f"{LLVM}/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o dev_{name}.out dev_{name}.o")
    run(f"{LLVM}/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=dev_{name}.out -output={name}.hipfb")
    h = list(host); h[blob[0]] = f"\t.incbin\t\"{out}/{name}.hipfb\""
    open(f"{out}/host_{name}.s", "w").write("\n".join(h))
    run(f"{LLVM}/clang -cc1as -triple x86_64-unknown-linux-gnu -filetype obj -target-cpu x86-64 -mrelocation-model pic -o {stem}_{name}.o host_{name}.s")
    objs = " ".join(f"{CSRC}/{o}.o" for o in others)
    run(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o {HERE}/dpvo_amd/libdpvo_hip_{name}.so {objs} {stem}_{name}.o")
    print(f"{name}: {rule}: {hits} sites -> dpvo_amd/libdpvo_hip_{name}.so")
