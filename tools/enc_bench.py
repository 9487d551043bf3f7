#!/usr/bin/env python
"""The two HIP encoder towers alone (dpvo_encoders_forward, 480x640): HIP-event time per forward and a checksum of the outputs
(variants must agree bit for bit).  Run under `rocprofv3 --kernel-trace --stats` for the per-launch table.  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpvo_amd.extractor import BasicEncoder4
from dpvo_amd.encoders import HipEncoders
dev = torch.device("cuda:0")
torch.manual_seed(0)
fnet = BasicEncoder4(128, 'instance').to(dev).eval()
inet = BasicEncoder4(384, 'none').to(dev).eval()
enc = HipEncoders(fnet, inet)
g = torch.Generator().manual_seed(1)
img = (2 * torch.rand(3, 480, 640, generator=g) - 0.5).half().to(dev)
f = torch.empty(120, 160, 128, dtype=torch.float16, device=dev); i = torch.empty(120, 160, 384, dtype=torch.float16, device=dev)
for _ in range(5):
    enc(img, fmap_out=f, imap_out=i)
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "50"))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    enc(img, fmap_out=f, imap_out=i)
e.record(); torch.cuda.synchronize()
print(f"encoders: {s.elapsed_time(e) / reps * 1e3:.1f} us per forward; checksum {f.view(torch.int16).to(torch.int64).sum().item()} "
      f"{i.view(torch.int16).to(torch.int64).sum().item()}  finite {bool(torch.isfinite(f).all() and torch.isfinite(i).all())}")
