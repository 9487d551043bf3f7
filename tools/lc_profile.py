#!/usr/bin/env python
"""Where a frame of BASELINE config 5 (LOOP_CLOSURE=True) spends its time on the bench stream: wall time of every timed frame with the
branch it took, and a cProfile of the frames that ran a global bundle adjustment.  Dev tool."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
from dpvo_amd.dpvo import DPVO
from dpvo_amd.net import VONet
dev = torch.device("cuda:0")
cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML); cfg.KEYFRAME_THRESH = -1.0
cfg.LOOP_CLOSURE = True; cfg.BUFFER_SIZE = max(cfg.BUFFER_SIZE, 70 + 45 + 80)
torch.manual_seed(2468)
slam = DPVO(cfg, VONet(), ht=480, wd=640, device=dev, defer_keyframe=True, overlap_encoders=True)
slam.motion_probe = lambda: 1.0e9
frames = bench.make_stream(64, 480, 640, dev)
intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
sync = bool(int(os.environ.get("LC_SYNC", "1")))
rows = []
pr = cProfile.Profile()
with torch.no_grad():
    for t in range(70):
        slam(float(t), frames[t % 64], intr, image_ready=False)
    slam.flush(); torch.cuda.synchronize()
    for t in range(70, 115):
        g0, pend = int(slam.ran_global_ba.sum()), slam._fu_pending
        t0 = time.perf_counter()
        pr.enable()
        slam(float(t), frames[t % 64], intr, image_ready=False)
        if sync:
            slam.flush(); torch.cuda.synchronize()
        pr.disable()
        dt = time.perf_counter() - t0
        rows.append((t, 1e3 * dt, int(slam.ran_global_ba.sum()) - g0, int(slam._fu_pending is not None and slam._fu_pending is not pend),
                     int(slam.pg.ii.numel()), int(slam.pg.ii_inac.numel()), slam.n))
print("frame  ms   globalBA one-call  E_active E_inactive n")
for r in rows:
    print(f"{r[0]:4d} {r[1]:7.3f}  {r[2]}  {r[3]}  {r[4]:7d} {r[5]:7d} {r[6]}")
gb = [r[1] for r in rows if r[2]]; oc = [r[1] for r in rows if r[3] and not r[2]]; other = [r[1] for r in rows if not r[2] and not r[3]]
print(f"global-BA frames: {len(gb)}, mean {sum(gb) / max(len(gb), 1):.3f} ms; one-call frames: {len(oc)}, mean {sum(oc) / max(len(oc), 1):.3f} ms; "
      f"other: {len(other)}, mean {sum(other) / max(len(other), 1):.3f} ms  (every frame followed by a device sync: {sync})")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(os.environ.get("LC_SORT", "cumulative")).print_stats(38)
print(s.getvalue()[:7000])
