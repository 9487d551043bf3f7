"""Host-side helpers with the names the reference's inference path imports from `dpvo.utils` (Timer,
coords_grid_with_index, flatmeshgrid; behaviour of dpvo/utils.py:8-29,39-55,86-88), written for this code base:
the timer does not synchronise the device unless it has to report, and the coordinate grid is built by broadcasting
instead of `repeat`."""
import torch

all_times = []          # milliseconds of every finished Timer (the reference keeps the same module-level list)


class Timer:
    """`with Timer("BA", enabled=...)`: HIP events around the block on the current stream; prints `name ms` on exit.
    Usable as a decorator too (`@Timer("x")`)."""

    def __init__(self, name, enabled=True):
        self.name, self.enabled = name, bool(enabled)
        self._ev = None

    def __enter__(self):
        if self.enabled:
            self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            begin, end = self._ev
            end.record()
            end.synchronize()               # only this event, not the whole device
            ms = begin.elapsed_time(end)
            all_times.append(ms)
            print(f"{self.name} {ms:.03f}")
        return False

    def __call__(self, fn):
        def wrapped(*a, **k):
            with Timer(self.name, self.enabled):
                return fn(*a, **k)
        return wrapped


def coords_grid_with_index(d, **kwargs):
    """d [b,n,h,w] (a depth / disparity plane) -> (coords [b,n,3,h,w] = (x, y, d) per pixel, index [b,n,1,h,w] = frame number)"""
    b, n, h, w = d.shape
    xs = torch.arange(w, dtype=torch.float, **kwargs).view(1, 1, 1, w).expand(b, n, h, w)
    ys = torch.arange(h, dtype=torch.float, **kwargs).view(1, 1, h, 1).expand(b, n, h, w)
    coords = torch.stack((xs, ys, d), dim=2)
    index = torch.arange(n, dtype=torch.float, **kwargs).view(1, n, 1, 1, 1).expand(b, n, 1, h, w).contiguous()
    return coords, index


def flatmeshgrid(*axes, **kwargs):
    """torch.meshgrid with every output flattened"""
    return tuple(g.reshape(-1) for g in torch.meshgrid(*axes, **kwargs))
