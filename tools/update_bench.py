#!/usr/bin/env python
"""Update operator alone at the size it runs at in steady state (E = 45 312 after removal, 47 712 while update() runs):
fused (update_fused.hip, WHICH=fused) vs launch-by-launch (update.hip, WHICH=unfused), HIP-event time, TFLOP/s against the 2.5 PFLOP/s dense f16 peak,
and the difference of the two results.  Dev tool; run under `rocprofv3 --kernel-trace --stats` for the per-kernel table."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import synthetic as S          # noqa: E402
from dpvo_amd import net as N                 # noqa: E402
from dpvo_amd.graph import GraphPlan          # noqa: E402


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
    reps = int(os.environ.get("REPS", "20"))
    which = os.environ.get("WHICH", "both")
    for n_frames in (41, 40):
        ii, jj, kk = S.replay_graph(n_frames)
        if n_frames == 41:      # the 2400 edges of the newest frame are appended before update(), removal comes after
            i0, j0, k0 = S.replay_graph(40)
            cfg = S.GraphCfg()
            n = 41
            M, r = cfg.M, cfg.PATCH_LIFETIME
            k1 = torch.arange(M * (n - r), M * (n - 1)); j1 = torch.full_like(k1, n - 1)
            k2 = torch.arange(M * (n - 1), M * n).repeat_interleave(r); j2 = torch.arange(n - r, n).repeat(M)
            kk = torch.cat([k0, k1, k2]); jj = torch.cat([j0, j1, j2]); ii = kk // M
        ii, jj, kk = ii.to(dev), jj.to(dev), kk.to(dev)
        E = ii.numel()
        torch.manual_seed(0)
        upd = N.Update(3).to(dev)
        upd.pack()
        upd.tiling = int(os.environ.get("TILING", "-1"))        # (this tool's own switch: dpvo_update_fused_params_t.tiling)
        plan = GraphPlan(ii, jj, kk)
        g = torch.Generator().manual_seed(1)
        imap = torch.randn(3456, 384, generator=g).half().to(dev)
        corr = torch.zeros(E, 896, dtype=torch.float16, device=dev)
        corr[:, :882] = torch.randn(E, 882, generator=g).half().to(dev)
        net = torch.randn(1, E, 384, generator=g).to(dev)
        kw = dict(plan=plan, inp_rows=kk, inp_mod=3456, corr_is_padded=True)
        flops = 2 * E * (896 * 384 + 16 * 384 * 384)
        out = {}
        for name, fz in (("fused", True), ("unfused", False)):
            if which not in ("both", name):
                continue
            ms = timeit(lambda: upd(net, imap[None], corr[None], None, ii, jj, kk, fused=fz, **kw), reps=reps)
            out[name] = upd(net, imap[None], corr[None], None, ii, jj, kk, fused=fz, **kw)
            print(f"E={E} {name:8s} {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s (reference FLOPs)  "
                  f"{flops / ms / 1e9 / 2500:.3f} of dense f16 peak")
        for nm in ("fused",):
            if nm not in out or "unfused" not in out:
                continue
            a, b = out[nm], out["unfused"]
            print("   |net| diff max %.2e  delta %.2e  weight %.2e" % ((a[0] - b[0]).abs().max().item(),
                  (a[1][0] - b[1][0]).abs().max().item(), (a[1][1] - b[1][1]).abs().max().item()))


if __name__ == "__main__":
    main()
