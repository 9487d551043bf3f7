#!/usr/bin/env python
"""Generates tests/golden/encoder.npz with the REFERENCE'S OWN BasicEncoder4 (imported from /root/reference/dpvo/extractor.py:200-264,
pure torch, CPU, f32): two towers as Patchifier builds them (net.py:98-99: fnet = 128 channels with InstanceNorm, inet = 384 channels
without normalisation), Kaiming-initialised under a fixed seed with the weights rounded to f16 (so that an f16 pipeline holds exactly
the same parameters) and non-zero biases; one 64 x 96 normalised image.  Stored: both state dicts (f16), the image, and the reference
outputs `fnet(image) / 4`, `inet(image) / 4` (net.py:116-117).  Run here:  python tests/golden/make_golden_encoder.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    sys.path.insert(0, os.path.join(REF, "dpvo"))
    import extractor as rext                                   # the reference file itself (it imports torch only)
    torch.manual_seed(20260924)
    fnet = rext.BasicEncoder4(output_dim=128, norm_fn="instance").eval()
    inet = rext.BasicEncoder4(output_dim=384, norm_fn="none").eval()
    out = {}
    with torch.no_grad():
        for name, m in (("fnet", fnet), ("inet", inet)):
            for p in m.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
                p.copy_(p.half().float())
            for k, v in m.state_dict().items():
                out[f"{name}.{k}"] = v.numpy().astype(np.float16)
        g = torch.Generator().manual_seed(3)
        img = (2 * (torch.randint(0, 256, (3, 64, 96), generator=g).float() / 255.0) - 0.5).half().float()
        out["image"] = img.numpy().astype(np.float16)
        out["fmap"] = (fnet(img[None, None]) / 4.0)[0, 0].numpy()
        out["imap"] = (inet(img[None, None]) / 4.0)[0, 0].numpy()
    np.savez_compressed(os.path.join(HERE, "encoder.npz"), **out)
    print(len(out), "arrays;", out["fmap"].shape, out["imap"].shape, "fmap rms %.3f imap rms %.3f" % (
        np.sqrt((out["fmap"] ** 2).mean()), np.sqrt((out["imap"] ** 2).mean())))


if __name__ == "__main__":
    main()
