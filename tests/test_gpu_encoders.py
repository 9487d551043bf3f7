"""HIP encoders (dpvo_amd/csrc/encoder.hip) vs the torch modules of dpvo_amd/extractor.py (same architecture and
state-dict layout as the reference's BasicEncoder4, extractor.py:200-264).

Reference = the torch towers evaluated in float32 on the f16-rounded weights and image (the "exact" result the f16
pipeline approximates).  Both MIOpen's f16 run and the HIP kernels round activations to f16 after every conv / norm;
stated tolerance on the final maps (|fmap| ~ 0.3, |imap| ~ 0.5 after the /4): atol 1.5e-2 + rtol 2e-2 worst element,
RMS error < 3e-3; the torch f16 (MIOpen) run itself differs from the f32 reference by the same amount."""
import pytest
import torch

from dpvo_amd.encoders import HipEncoders
from dpvo_amd.extractor import BasicEncoder4

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W", [(96, 128), (480, 640), (112, 176)])
def test_encoders_vs_torch(dev, H, W):
    torch.manual_seed(0)
    fnet = BasicEncoder4(128, 'instance').to(dev).eval()
    inet = BasicEncoder4(384, 'none').to(dev).eval()
    with torch.no_grad():
        for m in (fnet, inet):
            for p in m.parameters():
                p.copy_(p.half().float())
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p)).copy_(p.half().float())      # non-zero biases
    g = torch.Generator().manual_seed(1)
    img = (2 * (torch.randint(0, 256, (3, H, W), generator=g).float() / 255.0) - 0.5).half().to(dev)
    enc = HipEncoders(fnet, inet)
    fmap, imap = enc(img)
    assert fmap.shape == (H // 4, W // 4, 128) and imap.shape == (H // 4, W // 4, 384)
    with torch.no_grad():
        rf = (fnet(img.float()[None, None]) / 4.0)[0, 0].permute(1, 2, 0)
        ri = (inet(img.float()[None, None]) / 4.0)[0, 0].permute(1, 2, 0)
        hf = (fnet.half()(img[None, None]) / 4.0)[0, 0].permute(1, 2, 0).float()
        hi = (inet.half()(img[None, None]) / 4.0)[0, 0].permute(1, 2, 0).float()
    for name, out, ref, mi in (("fmap", fmap, rf, hf), ("imap", imap, ri, hi)):
        err = (out.float() - ref).abs()
        tol = 1.5e-2 + 2e-2 * ref.abs()
        rms = float(((out.float() - ref) ** 2).mean().sqrt())
        rms_mi = float(((mi - ref) ** 2).mean().sqrt())
        assert torch.isfinite(out).all()
        assert (err <= tol).all(), f"{name}: max err {float(err.max()):.3e}, rms {rms:.3e} (MIOpen f16 rms {rms_mi:.3e})"
        assert rms < max(3e-3, 2.0 * rms_mi), f"{name}: rms {rms:.3e} vs MIOpen f16 {rms_mi:.3e}"
    # deterministic
    f2, i2 = enc(img)
    assert torch.equal(f2, fmap) and torch.equal(i2, imap)
