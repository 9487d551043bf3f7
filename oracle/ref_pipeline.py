"""TEST INFRASTRUCTURE ONLY -- the reference's OWN tracker (`dpvo.dpvo.DPVO`, dpvo/dpvo.py:20-473, with its own net.py / patchgraph.py /
projective_ops.py / blocks.py / lietorch Python and its own native kernels cuda_corr / cuda_ba compiled for gfx950) running on the
MI355X: the trajectory-level checker of tests/test_zz_ref_pipeline.py and the same-box `ref_baseline` of bench.py.

Where the pieces come from (oracle/build_ref.py, run where /root/reference exists; everything lands in the git-ignored oracle/_ref/,
which travels to the GPU box with the tree):
    oracle/_ref/ref_cuda_corr.so, ref_cuda_ba.so   the reference's extensions (correlation_kernel.cu, ba_cuda.cu, block_e.cu)
    oracle/_ref/pyref/dpvo_reference/              the reference's Python package, unmodified, under another package name
    oracle/ref_standins.py                         torch stand-ins for torch_scatter / lietorch_backends / numba (absent here)
Nothing under dpvo_amd/ imports this module, and this module imports nothing of dpvo_amd's native library: the two trackers share
weights (a state dict), frames and random draws (same torch seed before each call), nothing else.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PYREF = os.path.join(_HERE, "_ref", "pyref")
_loaded = None


def available():
    """True when the staged reference package and both native modules are present (built where /root/reference exists)"""
    return (os.path.isfile(os.path.join(_PYREF, "dpvo_reference", "dpvo.py"))
            and all(os.path.isfile(os.path.join(_HERE, "_ref", n + ".so")) for n in ("ref_cuda_corr", "ref_cuda_ba")))


def load(native=None):
    """import the staged reference package (needs a GPU: dpvo/dpvo.py:17 allocates a CUDA tensor at import).
    Returns a namespace with DPVO, VONet, SE3, pops (the reference's own classes / modules)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    from . import ref_native, ref_standins
    if native is None:
        native = ref_native()
        if native is None:
            raise FileNotFoundError("oracle/_ref/ref_cuda_*.so missing: run oracle/build_ref.py where /root/reference exists")
    ref_standins.install(native)
    if _PYREF not in sys.path:
        sys.path.insert(0, _PYREF)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")             # torch.cuda.amp.autocast deprecation, torch.meshgrid indexing
        import dpvo_reference.dpvo as rd
        import dpvo_reference.net as rn
        import dpvo_reference.projective_ops as rp
        from dpvo_reference.lietorch import SE3
    _loaded = types.SimpleNamespace(DPVO=rd.DPVO, VONet=rn.VONet, SE3=SE3, pops=rp, dpvo_module=rd, net_module=rn)
    return _loaded


class Cfg(types.SimpleNamespace):
    """attribute bag with the fields dpvo/config.py:1-38 defines (the tracker only reads attributes)"""


def make_cfg(src=None, **over):
    """a config for the reference tracker from a dpvo_amd CfgNode / dict (same field names: dpvo/config.py)"""
    base = dict(BUFFER_SIZE=4096, CENTROID_SEL_STRAT="RANDOM", PATCHES_PER_FRAME=80, REMOVAL_WINDOW=20, OPTIMIZATION_WINDOW=12,
                PATCH_LIFETIME=12, KEYFRAME_INDEX=4, KEYFRAME_THRESH=12.5, MOTION_MODEL="DAMPED_LINEAR", MOTION_DAMPING=0.5,
                MIXED_PRECISION=True, LOOP_CLOSURE=False, BACKEND_THRESH=64.0, MAX_EDGE_AGE=1000, GLOBAL_OPT_FREQ=15,
                CLASSIC_LOOP_CLOSURE=False, LOOP_CLOSE_WINDOW_SIZE=3, LOOP_RETR_THRESH=0.04)
    if src is not None:
        base.update({k: src[k] for k in base if k in src})
    base.update(over)
    return Cfg(**base)


class _Feed(torch.nn.Module):
    """stands where an encoder tower stood and returns a tensor handed in from outside (the OTHER tracker's encoder output x 4:
    Patchifier.forward divides by 4, net.py:113-114 -- exact in f16)"""

    def __init__(self):
        super().__init__()
        self.value = None

    def forward(self, images):
        return self.value


def make_tracker(cfg, state_dict, ht, wd, accept_probe=True, feed_encoders=False):
    """the reference's DPVO on `cfg` with the given VONet weights (strict load).  accept_probe: the initialisation motion probe
    (dpvo.py:441-444) is answered with 'enough motion' -- with random weights its median |delta| means nothing (bench.py does the
    same on the other side).  feed_encoders: the two encoder towers are replaced by feeds (set with `feed(slam, fmap, imap)`)."""
    R = load()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = R.VONet()
        missing = net.load_state_dict({k: v.detach().float() for k, v in state_dict.items()}, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        slam = R.DPVO(cfg, net, ht=ht, wd=wd)
    if accept_probe:
        slam.motion_probe = lambda: 1.0e9
    if feed_encoders:
        slam.network.patchify.fnet = _Feed()
        slam.network.patchify.inet = _Feed()
    return slam


def feed(slam, fmap_hwc, imap_hwc):
    """hand one frame's encoder outputs ([h,w,128] and [h,w,384], already divided by 4 as the trackers store them) to a tracker
    built with feed_encoders=True"""
    p = slam.network.patchify
    p.fnet.value = (fmap_hwc.permute(2, 0, 1)[None, None] * 4).contiguous()
    p.inet.value = (imap_hwc.permute(2, 0, 1)[None, None] * 4).contiguous()


def call(slam, tstamp, image, intrinsics):
    """slam(tstamp, image, intrinsics) with the reference's warnings silenced and gradients off (its scripts run under no_grad)"""
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        slam(tstamp, image, intrinsics)


def snapshot(slam):
    """the state a frame leaves behind, as numpy (integer bookkeeping + the float state BA works on)"""
    n, pg = slam.n, slam.pg
    cpu = lambda t: t.detach().cpu().numpy()
    return dict(n=n, m=slam.m, counter=slam.counter, ii=cpu(pg.ii).astype(np.int64), jj=cpu(pg.jj).astype(np.int64),
                kk=cpu(pg.kk).astype(np.int64), ii_inac=cpu(pg.ii_inac).astype(np.int64), jj_inac=cpu(pg.jj_inac).astype(np.int64),
                kk_inac=cpu(pg.kk_inac).astype(np.int64), tstamps=np.asarray(pg.tstamps_[:n]).copy(),
                poses=cpu(pg.poses_[:n]).astype(np.float64), patches=cpu(pg.patches_[:n]).astype(np.float64),
                intrinsics=cpu(pg.intrinsics_[:n]).astype(np.float64), colors=cpu(pg.colors_[:n]),
                delta_keys=sorted(int(k) for k in pg.delta.keys()))


def throughput(slam, frames, intrinsics, warm, timed, seed=0):
    """frames/sec of the reference tracker on a resident stream: `warm` untimed calls, then `timed` calls bracketed by device
    synchronisations (the same bracket bench.py puts around the HIP tracker)."""
    import time
    n_img = frames.shape[0]
    for t in range(warm):
        torch.manual_seed(seed + t)
        call(slam, float(t), frames[t % n_img], intrinsics)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(warm, warm + timed):
        torch.manual_seed(seed + t)
        call(slam, float(t), frames[t % n_img], intrinsics)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return timed / dt, dt
