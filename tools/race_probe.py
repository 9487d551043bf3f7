#!/usr/bin/env python
"""Per-frame bit checksums of the pipeline state (no host syncs in the loop), to locate run-to-run divergence:
   python tools/race_probe.py OUT.npy [frames] ; compare two OUT files with --diff A B"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

COLS = ["fmap1", "fmap2", "gmap", "imap", "poses", "patches", "net", "ii"]

if sys.argv[1] == "--diff":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = a != b
    if not d.any():
        print("identical"); sys.exit(0)
    f = np.nonzero(d.any(1))[0][0]
    print("first divergence at frame", f, "columns", [COLS[c] for c in np.nonzero(d[f])[0]])
    for g in range(f, min(f + 3, len(a))):
        print(" frame", g, [COLS[c] for c in np.nonzero(d[g])[0]])
    sys.exit(1)

import bench
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
from dpvo_amd.dpvo import DPVO
from dpvo_amd.net import VONet

nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML)
cfg.KEYFRAME_THRESH = -1.0
cfg.BUFFER_SIZE = max(cfg.BUFFER_SIZE, nfr + 16)
torch.manual_seed(1234)
slam = DPVO(cfg, VONet(), ht=480, wd=640, device=dev, defer_keyframe=bool(int(os.environ.get("DEFER_KEYFRAME", "1"))),      # (this tool's own switches)
            overlap_encoders=int(os.environ.get("OVERLAP_ENC", "1")))
slam.motion_probe = lambda: 1.0e9
frames = bench.make_stream(64, 480, 640, dev)
intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
out = torch.zeros(nfr, len(COLS), dtype=torch.int64, device=dev)


def bits(t):
    t = t.contiguous()
    v = t.view(torch.int16) if t.element_size() == 2 else (t.view(torch.int32) if t.element_size() == 4 else t)
    return v.long().sum()


with torch.no_grad():
    for t in range(nfr):
        slam(float(t), frames[t % 64], intr)
        n = slam.n - 1
        out[t, 0] = bits(slam._fmap1_cl[n % slam.mem]); out[t, 1] = bits(slam._fmap2_cl[n % slam.mem])
        out[t, 2] = bits(slam._gmap_cl[n % slam.pmem]); out[t, 3] = bits(slam.imap_[n % slam.pmem])
        out[t, 4] = bits(slam.pg.poses_[:slam.n]); out[t, 5] = bits(slam.pg.patches_[:slam.n])
        out[t, 6] = bits(slam.pg.edges.view("net")); out[t, 7] = bits(slam.pg.edges.view("ii"))
    slam.flush()
torch.cuda.synchronize()
np.save(sys.argv[1], out.cpu().numpy())
print("saved", sys.argv[1], "finite poses:", bool(torch.isfinite(slam.pg.poses_[:slam.n]).all()))
