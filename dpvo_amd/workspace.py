"""Per-device, stream-ordered scratch buffers handed to the C ABI as `ws` (caller-owned, never retained)."""
import torch

_bufs = {}


def get(nbytes, device, tag="default"):
    key = (str(device), tag)
    buf = _bufs.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _bufs[key] = buf
    return buf
