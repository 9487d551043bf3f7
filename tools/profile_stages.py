#!/usr/bin/env python
"""Per-stage HIP-event timings of the hot path at the BASELINE config-2 size (E = 45 312).  Dev tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import altcorr, fastba, synthetic as S          # noqa: E402
from dpvo_amd import net as N                                  # noqa: E402
from dpvo_amd import projective_ops as pops                    # noqa: E402
from dpvo_amd.graph import GraphPlan                           # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    dev = torch.device("cuda:0")
    ii, jj, kk = (t.to(dev) for t in S.replay_graph(40))
    E = ii.numel()
    gmap, f0, f1, imap = S.make_features()
    g = gmap.permute(0, 2, 3, 1).reshape(-1, 9, 128).contiguous().to(dev)
    a = f0.permute(0, 2, 3, 1).contiguous().to(dev); b = f1.permute(0, 2, 3, 1).contiguous().to(dev)
    poses, patches, intr = (t.to(dev) for t in S.make_scene(40))
    imap = imap.to(dev)
    us, vs = kk % 3456, jj % 36
    torch.manual_seed(0)
    upd = N.Update(3).to(dev); P = upd.pack()
    res = {}
    res["plan_build"] = timeit(lambda: GraphPlan(ii, jj, kk))
    plan = GraphPlan(ii, jj, kk)
    res["reproject"] = timeit(lambda: pops.transform_coords(poses, patches, intr, ii, jj, kk))
    coords = pops.transform_coords(poses, patches, intr, ii, jj, kk)
    res["corr_pyramid"] = timeit(lambda: altcorr.corr_pyramid(g, a, b, coords, us, vs))
    res["corr_pyramid(order=by pair)"] = timeit(lambda: altcorr.corr_pyramid(g, a, b, coords, us, vs, order=plan.perm_p))
    oj = torch.argsort(vs, stable=True).int()
    res["corr_pyramid(order=by target frame)"] = timeit(lambda: altcorr.corr_pyramid(g, a, b, coords, us, vs, order=oj))
    corr = altcorr.corr_pyramid(g, a, b, coords, us, vs)
    net = torch.randn(1, E, 384, device=dev)
    res["update_total"] = timeit(lambda: upd(net, imap[None], corr[None], None, ii, jj, kk, plan=plan, inp_rows=kk,
                                             inp_mod=3456, corr_is_padded=True))
    x = torch.randn(E, 384, device=dev); xh = x.half()
    res["linear 896->384 relu"] = timeit(lambda: N.linear(corr._base if corr._base is not None else corr, P["c0"][0], P["c0"][1], epilogue=N.EPI_RELU, K=896))
    res["linear 384->384 f16"] = timeit(lambda: N.linear(xh, P["c2"][0], P["c2"][1]))
    res["linear 384->384 f32A gather"] = timeit(lambda: N.linear(x, P["c1"][0], P["c1"][1], rows=plan.ix, epilogue=N.EPI_RELU))
    res["linear 384->768 f32A"] = timeit(lambda: N.linear(x, P["akk"][0], P["akk"][1]))
    fg = N.linear(x, P["akk"][0], P["akk"][1])
    res["linear resadd"] = timeit(lambda: N.linear(xh, P["c1"][2], P["c1"][3], out=x, epilogue=N.EPI_RESADD))
    res["layernorm"] = timeit(lambda: N.layernorm(x, P["norm"][0], P["norm"][1], y_f32=x, y_f16=xh))
    res["softagg kk"] = timeit(lambda: N.softagg(fg, plan.perm_k, plan.patch_off, plan.counts[0:1], plan.n_patches()))
    res["softagg ij"] = timeit(lambda: N.softagg(fg, plan.perm_p, plan.pair_off, plan.counts[1:2], plan.n_pairs()))
    res["heads"] = timeit(lambda: N.heads(x, P["d"][0], P["d"][1], P["w"][0], P["w"][1]))
    target = coords[0, :, :, 1, 1].contiguous() + 0.5 * torch.randn(E, 2, device=dev)
    weight = torch.rand(E, 2, device=dev)
    p0, pt0 = poses.clone(), patches.clone()
    def ba():
        poses.copy_(p0); patches.copy_(pt0)
        fastba.BA(poses, patches, intr, target, weight, 1e-4, ii, jj, kk, 30, 40, M=96, iterations=2, plan=plan)
    res["BA (2 iters, + 2 copies)"] = timeit(ba)
    for k, v in res.items():
        print(f"{k:36s} {v:9.4f} ms")
    gemm_flops = 2 * E * (896 * 384 + 16 * 384 * 384)
    print(f"update GEMM TFLOP/s (all 17 E-row GEMMs over update_total): {gemm_flops / res['update_total'] / 1e9:.1f}")
    print(f"corr GB/s (algorithmic 52884 B/edge): {E * 52884 / res['corr_pyramid'] / 1e6:.1f}")


if __name__ == "__main__":
    main()
