"""The shipped library must not contain the packed-FP32 operand-select form that faults on MI355X (tools/isa_lint.py)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
isa_lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_lint)


def test_pattern_matches_only_the_faulting_selects():
    bad = ["v_pk_mul_f32 v[78:79], v[8:9], v[70:71] op_sel:[0,1] op_sel_hi:[1,0]",
           "v_pk_fma_f32 v[14:15], v[0:1], v[6:7], v[14:15] op_sel:[0,1,0] op_sel_hi:[1,0,1]",
           "v_pk_fma_f32 v[14:15], v[0:1], v[6:7], v[14:15] op_sel:[0,1,1] op_sel_hi:[1,0,0]",
           "v_pk_add_f32 v[26:27], v[28:29], v[26:27] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
           "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]"]
    good = ["v_pk_mul_f32 v[66:67], v[60:61], v[66:67] op_sel:[1,0] op_sel_hi:[0,1]",
            "v_pk_mul_f32 v[72:73], v[56:57], v[72:73] op_sel_hi:[0,1]",
            "v_pk_fma_f32 v[64:65], v[12:13], v[64:65], v[80:81] op_sel:[0,0,1] op_sel_hi:[1,1,0]",
            "v_pk_fma_f32 v[64:65], v[12:13], v[64:65], v[80:81] op_sel:[1,1,0] op_sel_hi:[0,0,1]",
            "v_pk_mov_b32 v[56:57], v[66:67], v[56:57] op_sel:[1,0]",     # measured clean (the probe's form 6)
            "v_pk_mov_b32 v[56:57], v[66:67], v[56:57] op_sel:[0,1]",
            "v_pk_mul_f32 v[0:1], v[2:3], v[4:5]",
            "v_pk_max_f16 v0, v1, v2 op_sel:[0,1] op_sel_hi:[1,0]"]
    assert all(isa_lint.BAD.search(x) for x in bad)
    assert not any(isa_lint.BAD.search(x) for x in good)


@pytest.mark.skipif(not os.path.exists(os.path.join(isa_lint.LLVM, "llvm-objdump")), reason="ROCm LLVM tools not installed")
@pytest.mark.parametrize("name", ["libdpvo_hip.so", "libdpvo_hip_cmp.so"])
def test_library_is_free_of_the_faulting_form(name):
    lib = os.path.join(ROOT, "dpvo_amd", name)
    assert os.path.exists(lib), "build the library first (__graft_entry__.build())"
    n_pk, hits = isa_lint.lint(lib)
    assert n_pk > (1000 if name == "libdpvo_hip.so" else 100)          # packed ops are in use (update operators), so the check is not vacuous
    assert hits == []


def test_missing_tools_are_reported_as_such():
    """exit status 2 ("could not check") is distinct from 1 ("the faulting form is present"): tools/isa_lint.py, csrc/Makefile"""
    import subprocess
    import sys
    env = dict(os.environ, DPVO_LLVM_BIN="/nonexistent/llvm/bin")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), os.path.join(ROOT, "dpvo_amd", "libdpvo_hip.so")],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "not found" in r.stderr
