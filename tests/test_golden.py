"""Oracle (and, on the GPU box, the HIP kernels) against the golden fixtures under tests/golden/, which were produced
by importing the REFERENCE'S OWN PYTHON (tests/golden/make_golden.py): Update.forward / SoftAgg / GatedResidual,
pops.transform / flow_mag / point_cloud, altcorr.patchify's bilinear glue, reduce_edges and the Python bundle
adjustment dpvo/ba.py (an implementation independent of ba_cuda.cu).  The fixtures are f64
reference outputs, so the oracle must match them to ~1e-9 (f32 storage: 1e-5)."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def test_pops_golden(oracle):
    d = _load("pops")
    co = oracle.reproject(d["poses"], d["patches"], d["intr"], d["ii"], d["jj"], d["kk"])
    assert np.allclose(co, d["coords"], atol=1e-8, rtol=1e-10)
    fl, val = oracle.flow_mag(d["poses"], d["patches"], d["intr"], d["ii"], d["jj"], d["kk"], beta=0.5)
    assert np.allclose(fl, d["flow"], atol=1e-8) and np.array_equal(val, d["valid"])
    pts = oracle.point_cloud(d["poses"], d["patches"], d["intr"], np.arange(14 * 8) // 8)
    assert np.allclose(pts, d["points"], atol=1e-8)


def test_patchify_golden(oracle):
    d = _load("patchify")
    assert np.allclose(oracle.patchify(d["net"], d["coords"], 1), d["out"], atol=1e-12)


def test_reduce_edges_golden(oracle):
    from dpvo_amd.patchgraph import reduce_edges
    d = _load("reduce_edges")
    for cap, key in ((1000, "edges"), (5, "edges_cap5")):
        assert np.array_equal(oracle.reduce_edges(d["flow"], d["ii"], d["jj"], cap, 1), d[key])
        assert np.array_equal(reduce_edges(d["flow"], d["ii"], d["jj"], cap, 1), d[key])      # product host code


def _update_case():
    from dpvo_amd.net import Update
    d = _load("update")
    upd, net, inp, corr, ii, jj, kk = H.golden_update_case(Update)
    cs = H.state_checksums(upd.state_dict())
    assert list(d["ck_names"]) == sorted(cs)
    assert np.allclose([cs[k] for k in sorted(cs)], d["ck_vals"], rtol=1e-12), "weights differ from the fixture's"
    assert np.allclose([float(net.abs().sum()), float(inp.abs().sum()), float(corr.abs().sum())], d["in_ck"], rtol=1e-12)
    return d, upd, net, inp, corr, ii, jj, kk


def test_update_golden(oracle):
    """oracle/update_ref.py == the reference's Update.forward (f64, no autocast) on the fixture"""
    from oracle import update_ref
    d, upd, net, inp, corr, ii, jj, kk = _update_case()
    saved = (update_ref._h, update_ref._f)
    update_ref._h = lambda x: x.double(); update_ref._f = lambda x: x.double()
    try:
        rn, rd, rw = update_ref.update_forward(upd.state_dict(), net[0], inp[0], corr[0], ii, jj, kk, half_scatter=False)
    finally:
        update_ref._h, update_ref._f = saved
    assert np.allclose(rn.numpy(), d["net_out"], atol=2e-5) and np.allclose(rd.numpy(), d["delta"], atol=2e-5)
    assert np.allclose(rw.numpy(), d["weight"], atol=2e-5)


def _ba_steps():
    """the two single Gauss-Newton steps of the fixture: (start poses, start patches, targets, golden poses, golden patches)"""
    d = _load("ba")
    return d, ((d["poses"].astype(np.float64), d["patches"].astype(np.float64), d["target"], d["poses_it1"], d["patches_it1"]),
               (d["poses_it1"], d["patches_it1"], d["target2"], d["poses_it2"], d["patches_it2"]))


def _ba_compare(P, pat, P0, pat0, gP, gpat, kk, atol, rtol, tag):
    """The reference's Python BA (dpvo/ba.py:170-173) clamps EVERY inverse depth to [1e-3, 10]; cuda_ba touches only the
    patches that have edges and applies max(d, 1e-4), d > 20 -> 1 (ba_cuda.cu:218-221).  Outside the clamps the two are
    the same update, so: patches with edges and an unclamped golden depth must agree, the others must be untouched."""
    kx = np.unique(kk)
    rest = np.setdiff1d(np.arange(pat.shape[0]), kx)
    gd = gpat[kx, 2, 1, 1]
    inside = (gd > 1e-3) & (gd < 10.0)
    assert inside.sum() >= 40
    H.assert_close(P, gP, atol, rtol, f"poses vs the reference's Python BA [{tag}]")
    H.assert_close(pat[kx][inside], gpat[kx][inside], atol, rtol, f"patches vs the reference's Python BA [{tag}]")
    assert np.array_equal(pat[rest], pat0[rest].astype(pat.dtype)) and np.array_equal(pat[:, :2], pat0[:, :2].astype(pat.dtype))
    assert np.abs(P - P0).max() > 5e-3, "the step must move the poses"


def test_ba_golden(oracle):
    """oracle.ba (restatement of cuda_ba, ba_cuda.cu:232-582) == the reference's independent Python BA (dpvo/ba.py:86-182 with
    pops.transform's Jacobians, projective_ops.py:71-108) run with the CUDA path's damping and bounds, step by step"""
    d, steps = _ba_steps()
    for n, (P0, pat0, tgt, gP, gpat) in enumerate(steps):
        P, pat, info, _ = oracle.ba(P0, pat0, d["intr"], tgt, d["weight"], 1e-4, d["ii"], d["jj"], d["kk"], int(d["t0"]),
                                    int(d["t1"]), iterations=1)
        assert info == 0
        _ba_compare(P, pat.reshape(-1, 3, 3, 3), P0, pat0, gP, gpat, d["kk"], 1e-7, 1e-9, f"oracle f64, step {n + 1}")


@pytest.mark.gpu
def test_hip_ba_against_golden(dev):
    """fastba.BA (HIP, f32) against the reference's Python BA on the fixture (f32 Schur solve: atol 2e-4, rtol 2e-3)"""
    from dpvo_amd import fastba
    d, steps = _ba_steps()
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)
    i = lambda k: torch.from_numpy(d[k]).to(dev)
    for n, (P0, pat0, tgt, gP, gpat) in enumerate(steps):
        pd, ptd = f(P0), f(pat0)
        st = (pd.cpu().numpy(), ptd.cpu().numpy())
        ret = fastba.BA(pd.view(1, -1, 7), ptd.view(1, -1, 3, 3, 3), f(d["intr"]).view(1, -1, 4), f(tgt)[None], f(d["weight"])[None],
                        torch.as_tensor([1e-4], device=dev), i("ii"), i("jj"), i("kk"), int(d["t0"]), int(d["t1"]), M=6,
                        iterations=1, eff_impl=False)
        assert ret == []
        _ba_compare(pd.cpu().numpy(), ptd.cpu().numpy(), st[0], st[1], gP, gpat, d["kk"], 2e-4, 2e-3, f"HIP f32, step {n + 1}")


@pytest.mark.gpu
def test_hip_against_golden(dev):
    """the HIP kernels against the reference-generated fixtures directly (f16/f32 tolerances of the op tests)"""
    from dpvo_amd import altcorr
    from dpvo_amd import projective_ops as pops
    d = _load("pops")
    t = lambda k: torch.from_numpy(d[k]).to(dev)
    co = pops.transform_coords(t("poses"), t("patches"), t("intr"), t("ii"), t("jj"), t("kk"))
    H.assert_close(co[0].cpu().numpy(), d["coords"], 2e-3, 1e-5, "reproject vs reference golden")
    fl, val = pops.flow_mag(t("poses"), t("patches"), t("intr"), t("ii"), t("jj"), t("kk"), beta=0.5)
    H.assert_close(fl.cpu().numpy(), d["flow"].reshape(len(d["ii"]), -1).mean(1), 2e-3, 1e-4, "flow_mag vs golden")
    p = _load("patchify")
    out = altcorr.patchify(torch.from_numpy(p["net"]).float()[None].to(dev), torch.from_numpy(p["coords"]).float()[None].to(dev), 1)
    H.assert_close(out[0].cpu().numpy(), p["out"], 1e-5, 1e-5, "patchify vs golden")
    # update operator: f16 GEMM operands vs the reference's f64 run -> tolerance of the update tests
    d2, upd, net, inp, corr, ii, jj, kk = _update_case()
    upd = upd.float().to(dev)
    out, (dl, w, _) = upd(net.float().to(dev), inp.half().to(dev), corr.half().to(dev), None, ii.to(dev), jj.to(dev), kk.to(dev))
    H.assert_close(out[0].cpu().numpy(), d2["net_out"], 3e-2, 2e-2, "update net vs reference golden")
    H.assert_close(dl[0].cpu().numpy(), d2["delta"], 2e-2, 2e-2, "update delta vs reference golden")
    H.assert_close(w[0].cpu().numpy(), d2["weight"], 1e-2, 1e-2, "update weight vs reference golden")


def _graph_golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph.npz"))


def test_graph_oracle_matches_reference_dpvo_bookkeeping():
    """oracle/graph_ref.py (the integer restatement the GPU tracker is compared with bit for bit) against the state the
    REFERENCE'S OWN DPVO class produced (tests/golden/make_golden_graph.py: dpvo/dpvo.py:215-238,266-310,362-375,377-473 run on
    the CPU with the float pipeline replaced by the same scripted decisions): n, m, counter, the active and inactive edge lists
    and the timestamps after every one of the 46 frames, and the removed-frame links."""
    from oracle.graph_ref import GraphRef
    g = _graph_golden()
    M = int(g["dpvo_M"])
    ref = GraphRef(M=M, PATCH_LIFETIME=13, REMOVAL_WINDOW=22, BUFFER_SIZE=256)
    o = {k: 0 for k in ("E", "E_inac", "t")}
    for t, (accept, drop) in enumerate(g["dpvo_decisions"]):
        ref.frame(bool(accept), bool(drop))
        E, Ei, n = int(g["dpvo_E"][t]), int(g["dpvo_E_inac"][t]), int(g["dpvo_n"][t])
        assert (ref.n, ref.m, ref.counter) == (n, int(g["dpvo_m"][t]), int(g["dpvo_counter"][t])), t
        for k in ("ii", "jj", "kk"):
            assert np.array_equal(getattr(ref, k), g["dpvo_" + k][o["E"]:o["E"] + E]), (t, k)
            assert np.array_equal(getattr(ref, k + "_inac"), g["dpvo_" + k + "_inac"][o["E_inac"]:o["E_inac"] + Ei]), (t, k)
        assert np.array_equal(ref.tstamps_[:n], g["dpvo_tstamps"][o["t"]:o["t"] + n]), t
        o["E"] += E; o["E_inac"] += Ei; o["t"] += n
    assert sorted(ref.delta.keys()) == g["dpvo_delta_keys"].tolist()
    assert [ref.delta[k] for k in sorted(ref.delta.keys())] == g["dpvo_delta_t0"].tolist()


def _encoder_golden():
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encoder.npz"))
    sd = {"fnet": {}, "inet": {}}
    for k in g.files:
        if k.startswith(("fnet.", "inet.")):
            sd[k[:4]][k[5:]] = torch.from_numpy(g[k].astype(np.float32))
    return g, sd


def test_extractor_module_is_the_reference_architecture():
    """dpvo_amd.extractor.BasicEncoder4 loads the state dicts of the REFERENCE'S BasicEncoder4 (extractor.py:200-264; golden made by
    tests/golden/make_golden_encoder.py from the imported reference file) strictly -- same keys, same shapes -- and reproduces the
    reference's outputs in f32 on the CPU: the module the HIP encoders are tested against IS the reference architecture."""
    from dpvo_amd.extractor import BasicEncoder4
    g, sd = _encoder_golden()
    img = torch.from_numpy(g["image"].astype(np.float32))
    for name, dim, norm in (("fnet", 128, "instance"), ("inet", 384, "none")):
        m = BasicEncoder4(output_dim=dim, norm_fn=norm).eval()
        m.load_state_dict(sd[name], strict=True)
        with torch.no_grad():
            out = (m(img[None, None]) / 4.0)[0, 0].numpy()
        H.assert_close(out, g["fmap" if name == "fnet" else "imap"], 2e-5, 2e-5, name)
