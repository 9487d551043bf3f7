"""HIP path of the Patchifier's two encoders (dpvo_amd/csrc/encoder.hip): weight repacking + the C-ABI call.

The nn.Module towers in dpvo_amd/extractor.py remain the parameter containers (state-dict compatible with dpvo.pth);
`pack_tower` derives the f16 operand images once."""
import ctypes

import torch

from . import _lib as L
from . import workspace


def _conv(w):
    """[Cout, Cin, kh, kw] -> [K/32, Cout, 32] f16 with K = (kh, kw, cin): one MFMA k-step of all filters is contiguous, so
    a wave's fragment load (16 filters x 64 B) reads one 1 KB run instead of 16 half-used cache lines"""
    cout = w.shape[0]
    w2 = w.detach().permute(0, 2, 3, 1).reshape(cout, -1).to(torch.float16)
    return w2.view(cout, -1, 32).permute(1, 0, 2).contiguous()


def pack_tower(enc):
    """22 f16 tensors in the order dpvo_encoders_forward expects"""
    h = lambda t: t.detach().to(torch.float16).contiguous()
    w1 = enc.conv1.weight.detach()                                  # [32, 3, 7, 7]
    k = torch.zeros(32, 7, 3, 8, dtype=torch.float32, device=w1.device)
    k[:, :, :, :7] = w1.permute(0, 2, 1, 3)                         # (n, kh, c, kw)
    w1p = torch.zeros(32, 192, dtype=torch.float16, device=w1.device)
    w1p[:, :168] = k.reshape(32, 168).to(torch.float16)
    out = [w1p, h(enc.conv1.bias)]
    l1, l2 = enc.layer1, enc.layer2
    for conv in (l1[0].conv1, l1[0].conv2, l1[1].conv1, l1[1].conv2, l2[0].conv1, l2[0].conv2, l2[0].downsample[0],
                 l2[1].conv1, l2[1].conv2, enc.conv2):
        out += [_conv(conv.weight), h(conv.bias)]
    return out


class HipEncoders:
    def __init__(self, fnet, inet):
        self.tensors = pack_tower(fnet) + pack_tower(inet)           # keep alive
        self.ptrs = (ctypes.c_void_p * 44)(*[t.data_ptr() for t in self.tensors])
        self.dev = self.tensors[0].device

    def __call__(self, img16, fmap_out=None, imap_out=None, hold_event=None, hold_at=0):
        """img16 [3,H,W] f16 (normalised image) -> fmap [H/4,W/4,128], imap [H/4,W/4,384] (NHWC f16, already / 4).
        hold_event (a recorded torch.cuda.Event) / hold_at: the current stream waits for the event in front of launch hold_at."""
        assert img16.dtype == torch.float16 and img16.is_contiguous() and img16.dim() == 3
        _, H, W = img16.shape
        h, w = H // 4, W // 4
        if fmap_out is None:
            fmap_out = torch.empty(h, w, 128, dtype=torch.float16, device=self.dev)
        if imap_out is None:
            imap_out = torch.empty(h, w, 384, dtype=torch.float16, device=self.dev)
        nbytes = L.lib().dpvo_encoders_workspace_bytes(L.i32(H), L.i32(W))
        if nbytes == 0:
            raise L.DPVOHipError(f"encoders need H, W multiples of 16 (got {H}x{W})")
        ws = workspace.get(nbytes, self.dev, "enc")
        L.check(L.lib().dpvo_encoders_forward_hold(L.ptr(img16), self.ptrs, L.ptr(fmap_out), L.ptr(imap_out), L.i32(H), L.i32(W),
                                                   L.ptr(ws), ctypes.c_size_t(ws.numel()),
                                                   ctypes.c_void_p(hold_event.cuda_event if hold_event is not None else 0),
                                                   L.i32(hold_at), L.stream()), "dpvo_encoders_forward")
        return fmap_out, imap_out
