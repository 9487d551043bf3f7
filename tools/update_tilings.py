#!/usr/bin/env python
"""Update operator alone at E = 47 712 (the size it runs at inside update(): 45 312 + the newest frame's 2 400 edges), one HIP-event time
per tiling of dpvo_update_fused_params_t.tiling (include/dpvo_hip.h): bit 0 / 1 = chains / first kernel at 64-row tiles x 2 workgroups per
CU, bits 2 / 3 / 4 = last kernel / first kernel / chains in the 12-wave geometry.  Interleaved A/B/A/B rounds so that a clock drift of the
box shows up as a spread, not as a difference.

    python tools/update_tilings.py [tilings, default 1,5,9,13,29] [rounds, default 3] [start skews in us, default 0]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import synthetic as S          # noqa: E402
from dpvo_amd import net as N                 # noqa: E402
from dpvo_amd.graph import GraphPlan          # noqa: E402


def main():
    tilings = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,5,9,13,29").split(",")]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    skews = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
    tilings = [(t, k) for t in tilings for k in skews]
    dev = torch.device("cuda:0")
    i0, j0, k0 = S.replay_graph(40)
    cfg = S.GraphCfg()
    n, M, r = 41, cfg.M, cfg.PATCH_LIFETIME
    k1 = torch.arange(M * (n - r), M * (n - 1)); j1 = torch.full_like(k1, n - 1)
    k2 = torch.arange(M * (n - 1), M * n).repeat_interleave(r); j2 = torch.arange(n - r, n).repeat(M)
    kk = torch.cat([k0, k1, k2]); jj = torch.cat([j0, j1, j2]); ii = kk // M
    ii, jj, kk = ii.to(dev), jj.to(dev), kk.to(dev)
    E = ii.numel()
    torch.manual_seed(0)
    upd = N.Update(3).to(dev)
    upd.pack()
    plan = GraphPlan(ii, jj, kk)
    g = torch.Generator().manual_seed(1)
    imap = torch.randn(3456, 384, generator=g).half().to(dev)
    corr = torch.zeros(E, 896, dtype=torch.float16, device=dev)
    corr[:, :882] = torch.randn(E, 882, generator=g).half().to(dev)
    net = torch.randn(1, E, 384, generator=g).to(dev)
    kw = dict(plan=plan, inp_rows=kk, inp_mod=3456, corr_is_padded=True, fused=True)
    flops = 2 * E * (896 * 384 + 16 * 384 * 384)
    reps = int(os.environ.get("REPS", "30"))
    times = {t: [] for t in tilings}
    outs = {}
    for rd in range(rounds):
        for t in tilings:
            upd.tiling, upd.start_skew = t
            for _ in range(3):
                o = upd(net, imap[None], corr[None], None, ii, jj, kk, **kw)
            outs[t] = o
            torch.cuda.synchronize()
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                upd(net, imap[None], corr[None], None, ii, jj, kk, **kw)
            e.record(); torch.cuda.synchronize()
            times[t].append(s.elapsed_time(e) / reps * 1e3)
    ref = outs[tilings[0]]
    for t in tilings:
        ts = times[t]
        o = outs[t]
        print(f"E={E} tiling {t[0]:2d} skew {t[1]:2d}: " + " ".join(f"{v:7.1f}" for v in ts) + f" us   best {min(ts):7.1f} us = {flops / min(ts) / 1e6 / 2500:.3f} of the dense f16 peak"
              f"   net equal to tiling {tilings[0][0]}: {bool(torch.equal(o[0], ref[0]))}  |delta| diff {(o[1][0] - ref[1][0]).abs().max().item():.1e}  |weight| diff {(o[1][1] - ref[1][1]).abs().max().item():.1e}")


if __name__ == "__main__":
    main()
