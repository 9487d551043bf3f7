#!/usr/bin/env python
"""Track one image sequence with dpvo_amd and report the trajectory error -- the runner behind BASELINE configs 1 and 3
(demo.py:25-56 / evaluate_euroc.py:29-55,104-119 of the reference: read frames -> DPVO -> terminate() -> Sim(3)-aligned ATE RMSE),
without cv2 / evo: frames are read by dpvo_amd.stream (OpenCV if present, else Pillow / .npy), the ATE by dpvo_amd.traj.

    python tools/evaluate.py --network dpvo.pth --imagedir datasets/EUROC/MH_01_easy/mav0/cam0/data --calib calib/euroc.txt \
        --stride 2 --groundtruth datasets/euroc_groundtruth/MH_01_easy.txt --timestamps-from-names
    python tools/evaluate.py --random-weights --imagedir /tmp/frames --calib /tmp/calib.txt        # plumbing check, no checkpoint

`run()` has the reference's signature and return value, so `from tools.evaluate import run` is a drop-in for the scripts' own."""
import argparse
import json
import os
import sys
import time
from multiprocessing import Process, Queue

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dpvo_amd import traj as T                                    # noqa: E402
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML, FAST_YAML  # noqa: E402
from dpvo_amd.stream import image_stream, list_images, video_stream   # noqa: E402


@torch.no_grad()
def run(cfg, network, imagedir, calib, stride=1, skip=0, viz=False, timeit=False, device="cuda:0"):
    """demo.py:25-56: returns (poses [N,7], tstamps [N]), (points, colors, (fx, fy, cx, cy, H, W))"""
    from dpvo_amd.dpvo import DPVO
    from dpvo_amd.utils import Timer
    slam = None
    queue = Queue(maxsize=8)
    reader = Process(target=image_stream if os.path.isdir(imagedir) else video_stream, args=(queue, imagedir, calib, stride, skip))
    reader.start()
    dev = torch.device(device)
    H = W = intr_np = None
    up = torch.cuda.Stream(device=dev)                            # uploads on their own stream: the tracker waits for the
    while True:                                                   # frame's event only, not for the previous frame's kernels
        t, image, intr_np = queue.get()
        if t < 0:
            break
        with torch.cuda.stream(up):
            img = torch.from_numpy(image).permute(2, 0, 1).to(dev, non_blocking=False)
            intrinsics = torch.from_numpy(intr_np).to(dev, torch.float32)
            ready = torch.cuda.Event()
            ready.record(up)
        torch.cuda.current_stream(dev).wait_event(ready)
        if slam is None:
            _, H, W = img.shape
            slam = DPVO(cfg, network, ht=H, wd=W, viz=viz, device=dev)
        with Timer("SLAM", enabled=timeit):
            slam(t, img, intrinsics, image_ready=ready)
    reader.join()
    points = slam.pg.points_.cpu().numpy()[:slam.m]
    colors = slam.pg.colors_.view(-1, 3).cpu().numpy()[:slam.m]
    return slam.terminate(), (points, colors, (*intr_np, H, W))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--network", default="dpvo.pth")
    ap.add_argument("--random-weights", action="store_true", help="no checkpoint: random-init VONet (plumbing check only)")
    ap.add_argument("--imagedir", required=True)
    ap.add_argument("--calib", required=True)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--skip", type=int, default=0)
    ap.add_argument("--config", default="default", help="default | fast | path to a yaml file")
    ap.add_argument("--opts", nargs="+", default=[])
    ap.add_argument("--groundtruth", default=None, help="TUM file (t x y z qx qy qz qw)")
    ap.add_argument("--timestamps-from-names", action="store_true", help="frame timestamps = file names (EuRoC: ns -> s / 1e9 off)")
    ap.add_argument("--save-trajectory", default=None)
    ap.add_argument("--seed", type=int, default=1234)                     # evaluate_euroc.py:81
    args = ap.parse_args()

    cfg = base_cfg.clone()
    if args.config in ("default", "fast"):
        cfg.merge_from_dict(DEFAULT_YAML if args.config == "default" else FAST_YAML)
    else:
        cfg.merge_from_file(args.config)
    cfg.merge_from_list(args.opts)
    torch.manual_seed(args.seed)
    network = args.network
    if args.random_weights:
        from dpvo_amd.net import VONet
        network = VONet()
    elif not os.path.exists(network):
        sys.exit(f"{network} not found (pass --random-weights for a plumbing run)")
    t0 = time.perf_counter()
    (poses, tstamps), (points, colors, calib) = run(cfg, network, args.imagedir, args.calib, args.stride, args.skip)
    dt = time.perf_counter() - t0
    if args.timestamps_from_names and os.path.isdir(args.imagedir):
        names = [float(p.stem) for p in list_images(args.imagedir, args.stride, args.skip)]
        tstamps = np.asarray(names[:len(tstamps)], np.float64)
    out = {"frames": int(len(tstamps)), "seconds": round(dt, 3), "frames_per_sec_incl_io": round(len(tstamps) / dt, 2),
           "finite": bool(np.isfinite(poses).all()), "points": int(points.shape[0])}
    if args.save_trajectory:
        T.save_tum(args.save_trajectory, tstamps, poses)
        out["trajectory"] = args.save_trajectory
    if args.groundtruth:
        tr, pr = T.load_tum(args.groundtruth)
        ie, ir = T.associate(tstamps, tr)
        out["matched"] = int(ie.size)
        out["ate_rmse_m"] = T.ate_rmse(poses[ie, :3], pr[ir, :3], align=True, correct_scale=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
