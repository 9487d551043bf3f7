"""The C-ABI shared library loads on a CPU-only host and exports every symbol include/dpvo_hip.h declares.
(No compute calls here: those need a GPU and live in the -m gpu tests.)"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dpvo_hip.h")).read()
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
    assert lib.dpvo_abi_version() >= 1


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
