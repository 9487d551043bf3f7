#!/usr/bin/env python
"""Generates tests/golden/graph.npz by running the REFERENCE'S OWN `DPVO` and `PatchGraph` classes (imported from
/root/reference/dpvo, nothing copied) on the CPU of this container:

  * `dpvo.dpvo.DPVO.__call__ / append_factors / remove_factors / keyframe / __edges_forw / __edges_back`
    (dpvo/dpvo.py:215-238,266-310,362-375,377-473) driven over 49 frames through a scripted list of the two data-dependent
    decisions (motion probe accepted?  keyframe dropped?) -- the integer state after every frame is the golden;
  * `dpvo.patchgraph.PatchGraph.edges_loop` (patchgraph.py:56-82, with the reference's `reduce_edges`) and `.normalize`
    (:84-95) on a synthetic loop-closure state.

What is replaced, and only that: the native extensions and absent third-party packages (the stubs of make_golden.py: lietorch /
altcorr / fastba backends on the CPU oracle, torch_scatter, numba, pypose, yacs), `device="cuda"` in tensor factories (mapped to
the CPU), the network (a stub whose `patchify` returns tensors of the right shapes: its values never reach an index) and
`DPVO.update` / `motion_probe` / `motionmag` (the float pipeline: replaced by the scripted decisions).  Run here:
    python tests/golden/make_golden_graph.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

from make_golden import install_stubs  # noqa: E402
from dpvo_amd import synthetic as S    # noqa: E402

# the decisions of tests/test_gpu_dpvo.py::test_bookkeeping_bit_exact_and_state_sane: (probe accepted, keyframe dropped)
DECISIONS = [(True, False)] * 3 + [(False, False)] * 2 + [(True, False)] * 9 + [(True, True)] * 3 + \
            [(True, False)] * 22 + [(True, True), (True, False), (True, True)] + [(True, False)] * 4


def cuda_to_cpu():
    """tensor factories called with device='cuda' build CPU tensors instead"""
    def wrap(fn):
        def f(*a, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return f
    for name in ("zeros", "ones", "empty", "arange", "as_tensor", "tensor", "randn", "rand", "full", "eye", "zeros_like",
                 "ones_like"):
        setattr(torch, name, wrap(getattr(torch, name)))
    torch.Tensor.cuda = lambda self, *a, **k: self
    _to = torch.Tensor.to

    def to(self, *a, **k):                  # .to("cuda") / .to(device="cuda") stay on the CPU
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        if "device" in k and str(k["device"]).startswith("cuda"):
            k["device"] = "cpu"
        return _to(self, *a, **k)
    torch.Tensor.to = to


class Cfg(types.SimpleNamespace):
    pass


def make_cfg(M, **over):
    c = Cfg(PATCHES_PER_FRAME=M, BUFFER_SIZE=256, MIXED_PRECISION=False, LOOP_CLOSURE=False, CLASSIC_LOOP_CLOSURE=False,
            MAX_EDGE_AGE=1000, KEYFRAME_INDEX=4, KEYFRAME_THRESH=12.5, REMOVAL_WINDOW=22, OPTIMIZATION_WINDOW=12,
            PATCH_LIFETIME=13, MOTION_MODEL="DAMPED_LINEAR", MOTION_DAMPING=0.5, CENTROID_SEL_STRAT="RANDOM",
            GLOBAL_OPT_FREQ=15, BACKEND_THRESH=64.0, GRADIENT_BIAS=False)
    for k, v in over.items():
        setattr(c, k, v)
    return c


class StubNet:
    DIM, RES, P = 384, 4, 3

    def cuda(self):
        return self

    def eval(self):
        return self

    def patchify(self, image, patches_per_image=80, centroid_sel_strat="RANDOM", return_color=False):
        M = patches_per_image
        h, w = image.shape[-2] // 4, image.shape[-1] // 4
        fmap = torch.zeros(1, 1, 128, h, w)
        gmap = torch.zeros(1, M, 128, 3, 3)
        imap = torch.zeros(1, M, self.DIM, 1, 1)
        patches = torch.ones(1, M, 3, 3, 3)
        clr = torch.zeros(1, M, 3)
        return fmap, gmap, imap, patches, None, clr


def run_dpvo(RDPVO, M=16, ht=96, wd=128):
    cfg = make_cfg(M)
    slam = RDPVO(cfg, StubNet(), ht=ht, wd=wd)
    state = {}

    def fake_update():
        E = slam.pg.ii.numel()
        slam.pg.target = torch.zeros(1, E, 2)
        slam.pg.weight = torch.zeros(1, E, 2)
    slam.update = fake_update
    slam.pg.target = torch.zeros(1, 0, 2)
    slam.pg.weight = torch.zeros(1, 0, 2)
    slam.motion_probe = lambda: 1e9 if state["accept"] else 0.0
    slam.motionmag = lambda i, j: 0.0 if state["drop"] else 4 * cfg.KEYFRAME_THRESH
    intr = torch.tensor([100.0, 100.0, wd / 2, ht / 2])
    rec = {k: [] for k in ("n", "m", "counter", "E", "E_inac")}
    cat = {k: [] for k in ("ii", "jj", "kk", "ii_inac", "jj_inac", "kk_inac", "tstamps")}
    for t, (accept, drop) in enumerate(DECISIONS):
        state["accept"], state["drop"] = accept, drop
        slam(float(t), torch.zeros(3, ht, wd), intr)
        rec["n"].append(slam.n); rec["m"].append(slam.m); rec["counter"].append(slam.counter)
        rec["E"].append(slam.pg.ii.numel()); rec["E_inac"].append(slam.pg.ii_inac.numel())
        for k in ("ii", "jj", "kk", "ii_inac", "jj_inac", "kk_inac"):
            cat[k].append(getattr(slam.pg, k).numpy().astype(np.int64).copy())
        cat["tstamps"].append(slam.pg.tstamps_[:slam.n].copy())
    out = {k: np.asarray(v, np.int64) for k, v in rec.items()}
    for k, v in cat.items():
        out[k] = np.concatenate(v) if v else np.zeros(0, np.int64)
    out["delta_keys"] = np.asarray(sorted(int(k) for k in slam.pg.delta.keys()), np.int64)
    out["delta_t0"] = np.asarray([int(slam.pg.delta[k][0]) for k in sorted(slam.pg.delta.keys())], np.int64)
    out["decisions"] = np.asarray(DECISIONS, np.int64)
    out["M"] = np.int64(M)
    return out


def run_patchgraph(RPG, RSE3):
    """edges_loop + normalize on a synthetic 60-frame state (M = 8 patches per frame)"""
    M, n = 8, 60
    cfg = make_cfg(M, LOOP_CLOSURE=True, REMOVAL_WINDOW=22, GLOBAL_OPT_FREQ=15, KEYFRAME_INDEX=4, MAX_EDGE_AGE=1000,
                   BACKEND_THRESH=64.0, BUFFER_SIZE=64)
    pg = RPG(cfg, 3, 384, 1000, device="cpu", dtype=torch.float)
    poses, patches, intr = S.make_scene(n, M=M, ht=48, wd=64, seed=21, noise=0.01)
    pg.n, pg.m = n, n * M
    pg.poses_[:n] = poses
    pg.patches_[:n] = patches.view(n, M, 3, 3, 3)
    pg.intrinsics_[:n] = intr
    for f in range(64):
        pg.index_[f] = f
    # the slow camera of make_scene keeps every old patch in view: a loop closure scenario (flow below BACKEND_THRESH)
    kk, jj = pg.edges_loop()
    out = dict(loop_kk=kk.numpy().astype(np.int64), loop_jj=jj.numpy().astype(np.int64), poses=poses.numpy(), patches=patches.numpy(),
               intr=intr.numpy(), n=np.int64(n), M=np.int64(M))
    pg.delta[7] = (6, RSE3(torch.tensor([[0.1, -0.2, 0.05, 0.0, 0.0, 0.0, 1.0]])))
    pg.normalize()
    out.update(norm_poses=pg.poses_[:n].numpy().copy(), norm_patches=pg.patches_[:n].numpy().copy(),
               norm_points=pg.points_[:n * M].numpy().copy(), norm_delta=pg.delta[7][1].data.numpy().copy())
    return out


def main():
    assert os.path.isdir(REF), "needs the reference checkout"
    install_stubs()
    cuda_to_cpu()
    class _NoAutocast:                     # usable as `with autocast(...)` and as `@autocast(...)`
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def __call__(self, fn): return fn
    torch.cuda.amp.autocast = _NoAutocast
    sys.path.insert(0, REF)
    from dpvo.dpvo import DPVO as RDPVO                           # reference classes
    from dpvo.patchgraph import PatchGraph as RPG
    from dpvo.lietorch import SE3 as RSE3
    out = {"dpvo_" + k: v for k, v in run_dpvo(RDPVO).items()}
    out.update({"pg_" + k: v for k, v in run_patchgraph(RPG, RSE3).items()})
    np.savez_compressed(os.path.join(HERE, "graph.npz"), **out)
    print({k: getattr(v, "shape", None) for k, v in out.items()})
    print("loop edges:", out["pg_loop_kk"].size, " final n/m/E:", out["dpvo_n"][-1], out["dpvo_m"][-1], out["dpvo_E"][-1])


if __name__ == "__main__":
    main()
