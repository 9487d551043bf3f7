#!/usr/bin/env python
"""The keyframe step (dpvo_keyframe_step: decision + index lists + record, then the gathers) and the flow test alone at the bench
configuration's steady state, 200 calls each between two events.  Variants: result record copied to pinned host memory or not,
decision forced or from flow sums.  Dev tool (the step reads set A and writes set B: repeating it is harmless)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dpvo_amd import _lib as L, projective_ops as pops
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
from dpvo_amd.dpvo import DPVO
from dpvo_amd.net import VONet
from dpvo_amd.graph import GraphPlan

dev = torch.device("cuda:0")
cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML); cfg.KEYFRAME_THRESH = -1.0
torch.manual_seed(1234)
slam = DPVO(cfg, VONet(), ht=480, wd=640, device=dev, defer_keyframe=True, overlap_encoders=True)
slam.motion_probe = lambda: 1.0e9
frames = bench.make_stream(64, 480, 640, dev)
intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
with torch.no_grad():
    for t in range(70):
        slam(float(t), frames[t % 64], intr, image_ready=False)
    slam.flush(); torch.cuda.synchronize()
fu = slam._frame_update_buffers()
es, inac = slam.pg.edges, slam.pg.edges_inac
par = 0 if es.a is fu["sets"][0] else 1
kf = L.KeyframeStep.from_buffer_copy(fu["args"][par].kf)
room = min(es.E, 8 * slam.M * cfg.PATCH_LIFETIME)
inac.reserve(room)
I, o = inac.a, inac.E
kf.ii_inac, kf.jj_inac, kf.kk_inac = I["ii"].data_ptr() + 8 * o, I["jj"].data_ptr() + 8 * o, I["kk"].data_ptr() + 8 * o
kf.target_inac, kf.weight_inac, kf.inac_room = I["target"].data_ptr() + 8 * o, I["weight"].data_ptr() + 8 * o, room
kf.E, kf.n = es.E, slam.n
res = torch.zeros(16 + 4 + 4 * (es.cap // 1024 + 2), dtype=torch.float32, device=dev)
res[:4] = torch.tensor([100.0, 96.0, 100.0, 96.0])
host = torch.zeros(16, dtype=torch.float32).pin_memory()
kf.flow4, kf.result, kf.poses = res.data_ptr(), res.data_ptr() + 32, slam.pg.poses_.data_ptr()
print(f"E = {es.E}, n = {slam.n}")


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def step():
    L.check(L.lib().dpvo_keyframe_step(ctypes.byref(kf), L.stream()), "dpvo_keyframe_step")


for forced, hostcopy in ((0, False), (0, True), (-1, False), (-1, True), (1, False)):
    kf.forced = forced
    kf.result_host, kf.host_words = (host.data_ptr(), 8) if hostcopy else (None, 8)
    print(f"keyframe step, forced = {forced:2d}, host copy = {int(hostcopy)}: {timed(step):7.2f} us per call "
          f"(record {res.view(torch.int32)[8:14].tolist()})")
plan = GraphPlan(slam.pg.ii, slam.pg.jj, slam.pg.kk)
k = slam.n - cfg.KEYFRAME_INDEX
mm = lambda: pops.motionmag_pair(slam.poses, slam.patches, slam.intrinsics, slam.pg.ii, slam.pg.jj, slam.pg.kk, k - 1, k + 1, plan=plan, defer=True)
print(f"flow test alone (dpvo_motionmag_status, plan variant): {timed(mm):7.2f} us per call")
