#!/usr/bin/env python
"""Global bundle adjustment (BASELINE config 5: LOOP_CLOSURE, dpvo.py:312-326) at size: per-phase HIP-event times of one
`fastba.BA(..., eff_impl=True)` call with 2 iterations for growing numbers of free poses.  Synthetic sequences (M = 96 patches per
frame as in default.yaml, every edge kept = active + inactive, plus loop edges).  Dev tool -> profiles/rNN_gba.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import fastba, synthetic as S            # noqa: E402
from dpvo_amd.fastba import global_ba as G             # noqa: E402
from dpvo_amd.graph import GraphPlan                   # noqa: E402
from dpvo_amd import projective_ops as pops            # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M = int(os.environ.get("M", "96"))
    print(f"{'N free':>7s} {'edges':>8s} {'patches':>8s} | " + " | ".join(f"{n:>42s}" for n in
          ("linearise + Schur", "damping + Cholesky + substitutions (chol.hip)", "back-substitution + retraction")) + " | total ms (2 iterations)")
    sizes = tuple(int(v) for v in os.environ.get("GBA_SIZES", "50,100,200,400,800").split(","))
    for n in sizes:
        cfg = S.GraphCfg(M=M, REMOVAL_WINDOW=10 * n, PATCH_LIFETIME=13)
        ii, jj, kk = S.replay_graph(n, cfg)
        old = torch.arange(3, n - 40, max(1, (n - 43) // 12))[:12]
        ks = (old[:, None] * M + torch.arange(M)[None]).reshape(-1).repeat_interleave(3)
        js = torch.stack([n - 20 + (old % 7), n - 12 + (old % 5), n - 6 + (old % 3)], 1).repeat_interleave(M, 0).reshape(-1)
        ii = torch.cat([ii, ks // M]); jj = torch.cat([jj, js]); kk = torch.cat([kk, ks])
        poses, patches, intr = S.make_scene(n, M=M, seed=1)
        d = lambda t: t.to(dev)
        ii, jj, kk, poses, patches, intr = d(ii), d(jj), d(kk), d(poses), d(patches), d(intr)
        coords = pops.transform_coords(poses, patches, intr, ii, jj, kk)
        g = torch.Generator(device="cpu").manual_seed(0)
        target = coords[0, :, :, 1, 1].contiguous() + 0.5 * torch.randn(ii.numel(), 2, generator=g).to(dev)
        weight = torch.rand(ii.numel(), 2, generator=g).to(dev)
        plan = GraphPlan(ii, jj, kk)
        p0, pt0 = poses.clone(), patches.clone()
        acc = {}
        reps = 5
        for r in range(reps + 2):
            poses.copy_(p0); patches.copy_(pt0)
            G._PROFILE = [] if r >= 2 else None
            fastba.BA(poses, patches, intr, target, weight, 1e-4, ii, jj, kk, 1, n, M=M, iterations=2, eff_impl=True, plan=plan)
            torch.cuda.synchronize()
            if G._PROFILE:
                for name, a, b in G._PROFILE:
                    acc[name] = acc.get(name, 0.0) + a.elapsed_time(b) / reps
        G._PROFILE = None
        names = ("linearise + Schur", "damping + Cholesky + substitutions (chol.hip)", "back-substitution + retraction")
        print(f"{n - 1:7d} {ii.numel():8d} {plan.n_patches():8d} | " + " | ".join(f"{acc[k]:42.3f}" for k in names) +
              f" | {sum(acc.values()):.3f}")
        if os.environ.get("GBA_TRACE"):          # (library built by tools/gba_trace.sh: phase stamps of one wave of gba_row_kernel)
            import ctypes
            import numpy as np
            from dpvo_amd import _lib as L
            buf = (ctypes.c_ulonglong * 96)()
            assert L.lib().dpvo_debug_gba_trace(buf) == 0
            t = np.array(list(buf), dtype=np.int64)
            its = int(t[95]); t = t / 100.0
            step = [t[2 + i + 1] - t[2 + i] for i in range(min(its, 80) - 1)]
            print(f"    gba_row_kernel, wave 0 of the middle pose: B / v part {t[1] - t[0]:.1f} us, {its} source frames, per frame "
                  f"median {np.median(step):.1f} max {np.max(step):.1f} us, total {t[90] - t[0]:.1f} us")
            wb = (ctypes.c_ulonglong * (4096 * 3))()
            assert L.lib().dpvo_debug_gba_wg_trace(wb) == 0
            w = np.array(list(wb), dtype=np.int64).reshape(4096, 3)
            split = 4 if n - 1 <= 64 else (2 if n - 1 <= 128 else 1)
            nwg = min(4096, (n - 1) * split)
            w = w[:nwg]; t0_ = w[:, 0].min()
            dur = (w[:, 1] - w[:, 0]) / 100.0
            order = np.argsort(-dur)[:6]
            print(f"    {nwg} workgroups: start {(w[:, 0].max() - t0_) / 100.0:.1f} us apart, duration median {np.median(dur):.1f} max {dur.max():.1f} us, "
                  f"last end {(w[:, 1].max() - t0_) / 100.0:.1f} us; longest: " +
                  ", ".join(f"pose {i // split} part {i % split}: {dur[i]:.0f} us, {int(w[i, 2])} frames" for i in order))
            vb = (ctypes.c_uint * (1024 * 16 * 2))()
            assert L.lib().dpvo_debug_gba_wave_trace(vb) == 0
            v = np.array(list(vb), dtype=np.int64).reshape(1024, 16, 2)[:min(nwg, 1024)]
            wt, wb = v[:, :, 0] / 100.0, v[:, :, 1]
            print(f"    waves (after the B / v part): blocks per wave min {wb.min()} median {int(np.median(wb))} max {wb.max()}; time per wave median {np.median(wt):.1f} "
                  f"max {wt.max():.1f} us; us per block (waves with > 4 blocks) median {np.median((wt / np.maximum(wb, 1))[wb > 4]):.2f}; "
                  f"slowest wave: {wt.max():.0f} us with {wb.reshape(-1)[wt.argmax()]} blocks; corr(time, blocks) {np.corrcoef(wt.reshape(-1), wb.reshape(-1))[0, 1]:.2f}")


if __name__ == "__main__":
    main()
