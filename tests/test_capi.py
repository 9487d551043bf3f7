"""The C-ABI shared library loads on a CPU-only host and exports every symbol include/dpvo_hip.h declares.
(No compute calls here: those need a GPU and live in the -m gpu tests.)"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="dpvo_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dpvo_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from dpvo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libdpvo_hip.so does not export {n}"
    assert sorted(_lib.SYMBOLS) == names, "dpvo_amd/_lib.py SYMBOLS out of sync with include/dpvo_hip.h"
    lib.dpvo_abi_version.restype = ctypes.c_int
    hdr = open(os.path.join(ROOT, "include", "dpvo_hip.h")).read()
    from dpvo_amd import _lib as L_
    assert lib.dpvo_abi_version() == L_.ABI_VERSION == int(re.search(r"#define DPVO_ABI_VERSION (\d+)", hdr).group(1))
    # the comparator library (two more implementations of the update operator: test / measurement partners) exports what its
    # own header declares, and the product library does NOT carry those entries
    cmp_names = _declared("dpvo_hip_cmp.h")
    cmp = ctypes.CDLL(_lib.CMP_LIB_PATH)
    assert sorted(_lib.CMP_SYMBOLS) == cmp_names
    for n in cmp_names:
        assert hasattr(cmp, n), f"libdpvo_hip_cmp.so does not export {n}"
    import subprocess
    exported = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    for n in cmp_names:
        assert f" {n}\n" not in exported, f"{n} (a comparator entry) is defined in the product library"


def test_layout_and_workspace_queries_are_host_only():
    """dpvo_plan_layout is pure host code: usable without a GPU."""
    from dpvo_amd import _lib as L
    lay = L.plan_layout(1000)
    offs = [lay.perm_k, lay.ku, lay.kx, lay.patch_off, lay.ix, lay.jx, lay.perm_p, lay.pu, lay.pair_off, lay.pair_ij,
            lay.counts, lay.total_ints]
    assert offs == sorted(offs) and lay.total_ints == 9 * 1000 + 2 * 1001 + 2 * 1000 - 2000 + 4 or lay.total_ints > 11000
    assert L.plan_layout(0).total_ints > 0
    assert L.lib().dpvo_ba_workspace_bytes(L.i64(45312), L.i32(10)) > 0
    assert L.lib().dpvo_ba_workspace_bytes(L.i64(45312), L.i32(21)) == 0      # dense path limit (N <= 20)


def test_no_fallback_without_gpu():
    """product ops refuse CPU tensors instead of silently computing somewhere else"""
    import torch
    from dpvo_amd import altcorr, _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DPVOHipError):
        altcorr.corr(torch.zeros(1, 2, 4, 3, 3), torch.zeros(1, 2, 4, 8, 8), torch.zeros(1, 1, 2, 3, 3),
                     torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long), 3)


def test_product_never_imports_oracle():
    """only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch oracle/"""
    pkg = os.path.join(ROOT, "dpvo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt, f


def test_environment_reads_are_on_the_allow_list():
    """the product reads NO measurement switch from the environment.  Allowed: DPVO_HIP_LIB / DPVO_HIP_CMP_LIB (which build
    of the two libraries to load: dpvo_amd/_lib.py, dpvo_amd/integration_stubs.py) and what torch.distributed's launcher hands a rank
    (dpvo_amd/multiseq.py).  The library itself (csrc) reads none at all."""
    allowed = {"DPVO_HIP_LIB", "DPVO_HIP_CMP_LIB", "MASTER_ADDR", "HSA_ENABLE_IPC_MODE_LEGACY", "WORLD_SIZE", "RANK", "LOCAL_RANK",
               "MASTER_PORT", "LOCAL_WORLD_SIZE"}
    pkg = os.path.join(ROOT, "dpvo_amd")
    seen = set()
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                txt = open(path).read()
                for m in re.finditer(r"environ(?:\.get|\.setdefault)?\s*[\(\[]\s*['\"]([A-Za-z0-9_]+)['\"]", txt):
                    seen.add(m.group(1))
                    assert m.group(1) in allowed, f"{os.path.relpath(path, ROOT)} reads {m.group(1)} from the environment"
                assert "getenv" not in txt, f
            elif f.endswith((".hip", ".h")):
                assert "getenv" not in open(path).read(), f"{f}: a library entry has no business reading the environment"
    assert "DPVO_HIP_LIB" in seen


def test_integration_md_quotes_the_stub_file():
    """INTEGRATION.md sections 1-3 quote dpvo_amd/integration_stubs.py verbatim: every ```python block of those sections is a substring of
    the file (the file is what tests/test_gpu_integration_stubs.py executes; the document must not drift from it)"""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(os.path.join(ROOT, "dpvo_amd", "integration_stubs.py")).read()
    sec = md[md.index("## 1-3. The three import sites"):md.index("## 4. ")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 4
    for b in blocks:
        assert b.strip() in src, b[:200]
    assert sum(len(b) for b in blocks) > 0.85 * len(src[src.index("import ctypes"):])


def test_argument_validation_is_host_only():
    """every entry validates its arguments before touching the device: error codes without a GPU, never a crash / exit
    (the reference calls exit(1) in block_e.cu:20-27 and ba.cpp:151-152)"""
    from dpvo_amd import _lib as L
    lib = L.lib()
    cmp = L.cmp_lib()
    null = ctypes.c_void_p(0)
    INVALID, UNSUPPORTED = -1, -2
    assert lib.dpvo_update_fused_workspace_bytes(L.i64(45312), L.i64(2300)) > 45312 * 384 * 2 * 5
    assert lib.dpvo_update_fused_workspace_bytes(L.i64(-1), L.i64(0)) == 0
    assert lib.dpvo_update_forward_fused(null, null, null, null, L.i64(0), null, L.i64(896), null, L.i64(1), L.i64(1), null, L.i32(3),
                                         null, null, null, null, L.i64(10), null, ctypes.c_size_t(0), null) == INVALID
    assert lib.dpvo_keyframe_step(null, null) == INVALID and lib.dpvo_frame_update(null, null) == INVALID
    assert cmp.dpvo_update_workspace_bytes(L.i64(45312), L.i64(2300)) > 45312 * 384 * 2 * 5
    assert cmp.dpvo_update_workspace_bytes(L.i64(-1), L.i64(0)) == 0
    assert cmp.dpvo_update_forward(null, null, null, null, L.i64(0), null, L.i64(896), null, L.i64(1), L.i64(1), null, L.i32(3),
                                   null, null, null, null, L.i64(10), null, ctypes.c_size_t(0), null) == INVALID
    assert lib.dpvo_plan_build_ranged(null, null, null, L.i64(-1), null, null, ctypes.c_size_t(0), L.i64(0), L.i64(0), null) == INVALID
    assert lib.dpvo_frame_patches(null, null, null, null, null, null, null, null, L.f32(4.0), null, null, null, null, null, null,
                                  null, null, L.i32(-1), L.i32(1), L.i32(1), L.i32(4), L.i32(4), L.i32(128), L.i32(384), L.i32(3),
                                  L.i64(0), L.i64(0), null) == INVALID
    assert lib.dpvo_frame_patches(null, null, null, null, null, null, null, null, L.f32(4.0), null, null, null, null, null, null,
                                  null, null, L.i32(4), L.i32(1), L.i32(1), L.i32(4), L.i32(4), L.i32(128), L.i32(384), L.i32(5),
                                  L.i64(0), L.i64(0), null) == UNSUPPORTED
    assert cmp.dpvo_heads_target(null, null, null, null, null, null, L.i32(3), null, null, null, L.i64(-3), L.i32(384), null) == INVALID
    assert cmp.dpvo_heads_target(null, null, null, null, null, null, L.i32(3), null, null, null, L.i64(0), L.i32(384), null) == 0
    assert cmp.dpvo_linear(null, L.i32(0), L.i64(384), null, null, L.i64(384), null, null, L.i64(384), null, L.i64(0), null, L.i64(0),
                           L.i32(0), L.i32(0), L.i64(0), L.i32(384), L.i32(384), null) == 0       # M = 0: nothing to do
    assert lib.dpvo_encoders_workspace_bytes(L.i32(480), L.i32(640)) > 0
    assert lib.dpvo_encoders_workspace_bytes(L.i32(481), L.i32(640)) == 0                          # H, W multiples of 16


def test_reference_import_names_resolve():
    """`from dpvo.dpvo import DPVO` etc. (demo.py:10-14, evaluate_euroc.py:14-19) resolve to this package after compat.install()"""
    import subprocess, sys
    code = ("import dpvo_amd.compat as c; c.install(); "
            "from dpvo.dpvo import DPVO; from dpvo.config import cfg; from dpvo.utils import Timer; "
            "from dpvo.net import VONet; from dpvo.patchgraph import PatchGraph; from dpvo import altcorr, fastba, lietorch; "
            "from dpvo.lietorch import SE3; import dpvo_amd.dpvo as d; assert DPVO is d.DPVO; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_ctypes_structs_match_the_header(tmp_path):
    """the ctypes mirrors of the header's structs (plan layout, update parameter table, frame state) have the C sizes and
    field offsets: compiled from include/dpvo_hip.h with the host C compiler"""
    import subprocess
    from dpvo_amd import _lib as L
    from dpvo_amd.net import _UpdParams, _UpdFusedParams
    probes = {"dpvo_plan_layout_t": (L.PlanLayout, ["perm_k", "counts", "total_ints", "flow"]),
              "dpvo_update_fused_params_t": (_UpdFusedParams, ["w", "b", "ln_g", "ln_b", "d_w", "w_b", "tiling", "start_skew"]),
              "dpvo_ring_t": (L.Ring, ["base", "slot_bytes", "ring"]),
              "dpvo_keyframe_step_t": (L.KeyframeStep, ["ii", "weight_b", "ii_inac", "inac_room", "flow4", "keep_rows", "result_host", "host_words", "ring", "n_ring",
                                                        "E", "n", "forced", "keyframe_thresh"]),
              "dpvo_frame_update_t": (L.FrameUpdate, ["kf", "fs", "ev_fs", "ev_enc", "fmap_spec", "ev_record", "ev_update_done", "plan_stream", "ev_plan_fork", "ev_plan_done", "fs_auto", "index_map", "net", "net_rows", "n_kept", "poses", "upd", "ws_ba", "ws_plan_bytes", "result_dev", "ev", "m", "n_buffer",
                                                      "P", "iterations", "lmbda", "mm_beta"]),
              "dpvo_update_params_t": (_UpdParams, ["c0_w", "g1_b2", "w_b"]),
              "dpvo_frame_state_t": (L.FrameState, ["fmap", "index_map", "poses", "ix", "frame_next", "n_new", "res", "mm_scale", "M",
                                                    "P", "mm_n", "D"])}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dpvo_hip_cmp.h"', 'int main(void) {']
    for cname, (_, fields) in probes.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fields:
            lines.append(f'  printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, (cls, fields) in probes.items():
        assert int(out[cname]) == ctypes.sizeof(cls), cname
        for f in fields:
            assert int(out[f"{cname}.{f}"]) == getattr(cls, f).offset, f"{cname}.{f}"


def test_plan_layout_flow_region():
    """dpvo_plan_layout: the flow-test list sits 16-byte aligned at the end of the plan, 4 + 2 x 256 ints, for every E"""
    from dpvo_amd import _lib as L
    for E in (0, 1, 2, 3, 5, 96, 1023, 45312, 47712):
        lay = L.plan_layout(E)
        assert lay.flow % 4 == 0 and lay.flow >= lay.counts + 4 and lay.total_ints == lay.flow + 4 + 2 * 256, E


def test_committed_pmc_pass_belongs_to_the_committed_correlation_kernel():
    """bench.py's roofline.traffic comes from the newest profiles/rNN_corr_pmc.json only if the SHA-256 recorded in it equals the
    fingerprint of dpvo_amd/csrc/corr.hip + corr_dev.h in the tree; otherwise the line falls back to the streaming figure and says so.
    A commit that changes the kernel without a fresh tools/pmc_corr.sh pass must be visible HERE, not only in the driver's bench line."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    traffic, source = bench.pmc_traffic("default")
    assert traffic is not None, f"stale PMC pass: {source} (run tools/pmc_corr.sh on a GPU box and commit profiles/rNN_corr_pmc.json)"
    # (0.98 GB on the SURVEY 8d stream of round 6, 1.65-1.68 GB on the 2-D crop stream of rounds 1-5: how many window bytes the per-XCD L2s serve
    #  depends on how coherent the reprojected coordinates are; compulsory 0.28 GB, streaming model 2.52 GB)
    assert 0.3e9 < traffic < 2.6e9 and source.endswith("_corr_pmc.json"), (traffic, source)
