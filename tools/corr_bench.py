#!/usr/bin/env python
"""corr_pyramid_kernel alone on the BASELINE config-2 tensors (E = 47 712 as inside update(), all edges in bounds pattern of
dpvo_amd.synthetic): HIP-event time per launch and a checksum (variants must agree bit for bit).  Dev tool: CORR_VARIANT=k times
the measurement kernel of tools/probes/corr_variant.hip instead (tools/corr_variants.sh builds the library that carries it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpvo_amd import altcorr, synthetic as S
from tests import helpers as H
dev = torch.device("cuda:0")
ii, jj, kk = S.replay_graph(40)
E = ii.numel()
gmap, f0, f1, _ = S.make_features()
coords = S.make_coords(E).to(dev)
us, vs = (kk % 3456).to(dev), (jj % 36).to(dev)
g, a, b = H.gmap_cl(gmap).to(dev), H.to_cl(f0).to(dev), H.to_cl(f1).to(dev)
out = torch.empty(E, 896, dtype=torch.float16, device=dev)
variant = os.environ.get("CORR_VARIANT")
if variant is not None:
    # the measurement kernel of tools/probes/corr_variant.hip (tools/corr_variants.sh builds the library that carries it)
    import ctypes
    from dpvo_amd import _lib as L
    fn = L.lib().dpvo_corr_pyramid_variant
    def run():
        L.check(fn(L.ptr(g), L.ptr(a), L.ptr(b), L.ptr(coords), L.ptr(us), L.ptr(vs), L.ptr(out), L.i64(896), L.i64(E), L.i64(3456),
                   L.i64(36), L.i32(a.shape[1]), L.i32(a.shape[2]), L.i32(b.shape[1]), L.i32(b.shape[2]), L.i32(int(variant)), L.stream()),
                "dpvo_corr_pyramid_variant")
else:
    run = lambda: altcorr.corr_pyramid(g, a, b, coords, us, vs, out=out)
for _ in range(5):
    run()
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "40"))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    run()
e.record(); torch.cuda.synchronize()
print(f"E={E} variant={variant if variant is not None else 'product'}: "
      f"{s.elapsed_time(e) / reps * 1e3:.1f} us per launch; checksum {out[:, :882].float().abs().sum().item():.6f} "
      f"{out[:, :882].view(torch.int16).to(torch.int64).sum().item()}")
