"""Per-device, per-stream scratch buffers handed to the C ABI as `ws` (caller-owned, never retained by the library).

Keyed by (device, HIP stream, tag): two trackers on one GPU that run on different streams, or one tracker's main and side
streams, must not share scratch memory -- kernels of different streams may overlap in time."""
import torch

_bufs = {}
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_PER_STREAM = True


def get(nbytes, device, tag="default"):
    device = torch.device(device)
    if device.type == "cuda":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        stream = _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(idx).cuda_stream
    else:
        idx, stream = -1, 0
    key = (device.type, idx, stream if _PER_STREAM else 0, tag)
    buf = _bufs.get(key)
    if buf is None or buf.numel() < nbytes:
        # grown geometrically: the global BA's scratch (edge records, E blocks, the plan's sort buffers) grows with the inactive edge
        # store by a few KB EVERY frame that runs it, and a buffer sized exactly is a fresh hipMalloc of 15-100 MB each time -- the
        # caching allocator cannot serve a larger request from the block just released
        grow = 0 if buf is None else buf.numel() + buf.numel() // 2
        _bufs[key] = buf = None                 # (release first: the old block can then back the new request's neighbours)
        buf = torch.empty(max(int(nbytes), grow, 1 << 20), dtype=torch.uint8, device=device)
        _bufs[key] = buf
    return buf
