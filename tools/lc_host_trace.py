#!/usr/bin/env python
"""Host-side timeline of the config-5 leg's frames (bench.loop_closure_leg's tracker: LOOP_CLOSURE=True): which Python calls a frame
with a global BA spends its host time in, and where the host waits.  Wraps the tracker's methods with perf_counter stamps.  Dev tool.
    python tools/lc_host_trace.py [frames to print, default 3]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                          # noqa: E402
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML   # noqa: E402
import dpvo_amd.dpvo as D                             # noqa: E402
import dpvo_amd.fastba as fastba                      # noqa: E402
import dpvo_amd.graph as G                            # noqa: E402
import dpvo_amd.patchgraph as PG                      # noqa: E402

LOG, DEPTH = [], [0]


def wrap(owner, name, label=None):
    f = getattr(owner, name)
    lab = label or name

    def g(*a, **k):
        t0 = time.perf_counter(); DEPTH[0] += 1
        try:
            return f(*a, **k)
        finally:
            DEPTH[0] -= 1
            LOG.append((t0, time.perf_counter(), DEPTH[0], lab))
    setattr(owner, name, g)


def main():
    n_print = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML); cfg.KEYFRAME_THRESH = -1.0
    cfg.LOOP_CLOSURE = True; cfg.BUFFER_SIZE = max(cfg.BUFFER_SIZE, 70 + 45 + 80)
    frames = bench.make_stream(64, 480, 640, dev)
    intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
    from dpvo_amd.net import VONet
    torch.manual_seed(1234)
    slam = D.DPVO(cfg, VONet(), ht=480, wd=640, device=dev)
    slam.motion_probe = lambda: 1.0e9
    for name in ("flush", "_keyframe_finish", "_keyframe_begin", "update", "_DPVO__run_global_BA", "plan", "_frame_update_call",
                 "_frame_update_finish", "remove_factors", "append_factors", "reproject", "corr", "_check_plan_bounds", "_pace_hold"):
        wrap(D.DPVO, name)
    wrap(PG.PatchGraph, "edges_loop"); wrap(PG.PatchGraph, "normalize")
    wrap(D.fastba, "BA", "fastba.BA")
    wrap(G.GraphPlan, "__init__", "GraphPlan()")
    for name in ("host", "reserve", "gather_into", "keep"):
        if hasattr(PG.EdgeStore, name):
            wrap(PG.EdgeStore, name, "EdgeStore." + name)
    wrap(torch.cuda.Event, "synchronize", "Event.synchronize")
    with torch.no_grad():
        for t in range(70):
            slam(float(t), frames[t % 64], intr, image_ready=False)
        slam.flush(); torch.cuda.synchronize()
        if slam._hip_enc is not None:
            wrap(type(slam._hip_enc), "__call__", "encoders()")
        LOG.clear()
        marks = []
        for t in range(70, 115):
            gb0 = int(slam.ran_global_ba.sum())
            t0 = time.perf_counter()
            slam(float(t), frames[t % 64], intr, image_ready=False)
            marks.append((t, t0, time.perf_counter(), int(slam.ran_global_ba.sum()) - gb0, int(slam._pg.edges.E)))
        slam.flush(); torch.cuda.synchronize()
        t_end = time.perf_counter()
    print(f"45 frames in {(t_end - marks[0][1]) * 1e3:.2f} ms = {45 / (t_end - marks[0][1]):.1f} frames/sec")
    print("frame  host ms in slam()  start-to-start ms  global BA  E")
    for i, (t, a, b, gb, E) in enumerate(marks):
        nxt = marks[i + 1][1] if i + 1 < len(marks) else t_end
        print(f"{t:5d} {1e3 * (b - a):10.3f} {1e3 * (nxt - a):14.3f} {gb:8d} {E:8d}")
    shown = 0
    for i, (t, a, b, gb, E) in enumerate(marks):
        if not gb or shown >= n_print:
            continue
        shown += 1
        for j in (i, i + 1):
            if j >= len(marks):
                continue
            tt, aa, bb, g2, _ = marks[j]
            print(f"\n-- frame {tt} ({'global BA' if g2 else 'no global BA'}): slam() {1e6 * (bb - aa):.0f} us of host time; calls (start us, duration us):")
            for (s0, s1, d, lab) in sorted(x for x in LOG if aa <= x[0] < bb):
                if s1 - s0 > 4e-6:
                    print(f"   {1e6 * (s0 - aa):8.0f} {1e6 * (s1 - s0):8.0f}  {'  ' * d}{lab}")


if __name__ == "__main__":
    main()
