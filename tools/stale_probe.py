#!/usr/bin/env python
"""Cross-kernel visibility micro-test: kernel A (one workgroup) writes a small buffer, kernel B (many workgroups on all XCDs)
reads it -- same stream, back to back -- while another stream runs (a) nothing, (b) the HIP encoders, (c) torch matmuls.
Counts reads that did not see the latest write.  Dev tool."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd.encoders import HipEncoders
from dpvo_amd.net import VONet

dev = torch.device("cuda:0")
torch.manual_seed(0)
vo = VONet().to(dev)
enc = HipEncoders(vo.patchify.fnet, vo.patchify.inet)
img = (torch.randn(3, 480, 640, device=dev) / 2).half()
eo = (torch.empty(120, 160, 128, dtype=torch.float16, device=dev), torch.empty(120, 160, 384, dtype=torch.float16, device=dev))
mmA = torch.randn(4096, 4096, device=dev, dtype=torch.float16); mmC = torch.empty_like(mmA)
side = torch.cuda.Stream(device=dev)
X = torch.zeros(280, device=dev)                        # like poses_[40,7]
idx = torch.randint(0, 280, (1 << 20,), device=dev)
bad = torch.zeros(1, dtype=torch.int64, device=dev)
for mode in ("alone", "enc", "mm", "enc"):
    bad.zero_()
    torch.cuda.synchronize()
    for r in range(600):
        if mode != "alone" and r % 3 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    if mode == "enc":
                        enc(img, fmap_out=eo[0], imap_out=eo[1])
                    else:
                        torch.matmul(mmA, mmA, out=mmC)
        X.fill_(float(r))                               # kernel A
        Y = X[idx]                                      # kernel B
        bad += (Y != float(r)).sum()
    torch.cuda.synchronize()
    print(f"{mode:6s} stale reads: {int(bad.item())}")
