#!/usr/bin/env python
"""Refuse gfx950 code that contains the packed-FP32 operand-select form measured to fault on MI355X.

    python tools/isa_lint.py dpvo_amd/libdpvo_hip.so [more .so / .o files]

`v_pk_{mul,add,fma}_f32` with op_sel:[0,1,...] (LOW result = src0.lo (op) src1.HI) returns, in lanes 48-63, the result computed
with src1 = 0 whenever another wave on the same SIMD is issuing `v_mfma_f32_16x16x32_{f16,bf16}` with AGPR accumulators
(tools/probes/pk_opsel_probe.hip, profiles/r02_pk_opsel_probe.txt: 2e-4 of the executions beside the encoders, never alone).
The compiler picks that form freely when it packs scalar FP32 code, so every shipped code object is disassembled and checked;
a unit that trips this is compiled without packed-FP32 ops (csrc/Makefile, NOPK).  Exit status 1 on a hit.
"""
import os, re, subprocess, sys, tempfile
# LLVM tools: $DPVO_LLVM_BIN, else $ROCM_PATH/lib/llvm/bin, else /opt/rocm/lib/llvm/bin
LLVM = os.environ.get("DPVO_LLVM_BIN") or os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
BAD = re.compile(r"\bv_pk_(mul|add|fma)_f32\b.*\bop_sel:\[0,1")


def code_objects(path, tmp):
    fb = os.path.join(tmp, "fatbin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb], check=True)
    blob = open(fb, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        one, co = os.path.join(tmp, f"b{n}"), os.path.join(tmp, f"b{n}.co")
        open(one, "wb").write(blob[a:b])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={one}",
                        f"--output={co}", "--unbundle"], check=True)
        yield co


def lint(path):
    hits, n_pk, kernel = [], 0, "?"
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(path, tmp):
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    kernel = m.group(1)
                if "v_pk_" in line:
                    n_pk += 1
                    if BAD.search(line):
                        hits.append((kernel, line.split("//")[0].strip()))
    return n_pk, hits


if __name__ == "__main__":
    rc = 0
    missing = [t for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump") if not os.path.isfile(os.path.join(LLVM, t))]
    if missing:
        # exit status 2 = "could not check" (distinct from 1 = "the faulting form is present"); the Makefile treats both as fatal:
        # an unchecked library must not ship
        print(f"isa_lint: {', '.join(missing)} not found under {LLVM} (set DPVO_LLVM_BIN or ROCM_PATH)", file=sys.stderr)
        sys.exit(2)
    for p in sys.argv[1:]:
        n_pk, hits = lint(p)
        print(f"{p}: {n_pk} packed instructions, {len(hits)} of the faulting form")
        for k, ins in hits[:20]:
            print(f"   {k}: {ins}")
        rc |= bool(hits)
    sys.exit(rc)
