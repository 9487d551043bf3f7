#!/usr/bin/env python
"""Per-edge timeline of corr_pyramid_kernel (trace build: `make -C dpvo_amd/csrc trace`): how long one wave spends in each
dependent phase of an edge, and how many edges are in flight.  Dev tool."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DPVO_HIP_LIB"] = os.path.join(ROOT, "dpvo_amd", "libdpvo_hip_trace.so")
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402

from dpvo_amd import _lib as L, altcorr, synthetic as S          # noqa: E402
from dpvo_amd import projective_ops as pops                      # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ii, jj, kk = (t.to(dev) for t in S.replay_graph(40))
    E = ii.numel()
    gmap, f0, f1, _ = S.make_features()
    g = gmap.permute(0, 2, 3, 1).reshape(-1, 9, 128).contiguous().to(dev)
    a = f0.permute(0, 2, 3, 1).contiguous().to(dev); b = f1.permute(0, 2, 3, 1).contiguous().to(dev)
    poses, patches, intr = (t.to(dev) for t in S.make_scene(40))
    coords = pops.transform_coords(poses, patches, intr, ii, jj, kk)
    us, vs = kk % 3456, jj % 36
    for _ in range(3):
        altcorr.corr_pyramid(g, a, b, coords, us, vs)
    buf = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert L.lib().dpvo_debug_corr_trace_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    altcorr.corr_pyramid(g, a, b, coords, us, vs)
    torch.cuda.synchronize()
    L.lib().dpvo_debug_corr_trace_buffer(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(65536, 8)[:min(E, 65536)].astype(np.int64)
    ok = (t[:, 0] > 0) & (t[:, 6] > 0) & (t[:, 2] > 0) & (t[:, 4] > 0)
    t = t[ok]
    t0 = t[:, 0].min()
    names = ["indices + coords + templates", "level 0: window tiles + MFMA", "level 0: blend", "level 1: window tiles + MFMA",
             "level 1: blend", "row store"]
    print(f"{ok.sum()} edges (single-pass path); kernel span {(t[:, 6].max() - t0) / 100:.1f} us; median edge lifetime "
          f"{np.median(t[:, 6] - t[:, 0]) / 100:.2f} us")
    for i, n in enumerate(names):
        d = (t[:, i + 1] - t[:, i]) / 100.0
        print(f"   {n:32s} median {np.median(d):6.2f} us   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}")
    # edges in flight over time
    ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 6], -np.ones(len(t))], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    infl = np.cumsum(ev[:, 1])
    print(f"   edges in flight: median {np.median(infl):.0f}, max {infl.max():.0f}  (256 CUs -> {np.median(infl) / 256:.1f} waves per CU)")


if __name__ == "__main__":
    main()
