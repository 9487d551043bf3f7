"""Small helpers of the reference's dpvo/utils.py that sit on the inference path."""
from contextlib import ContextDecorator

import torch

all_times = []


class Timer(ContextDecorator):
    """HIP-event timer (dpvo/utils.py:8-29)."""

    def __init__(self, name, enabled=True):
        self.name = name
        self.enabled = enabled
        if self.enabled:
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.enabled:
            self.start.record()

    def __exit__(self, type, value, traceback):
        if self.enabled:
            self.end.record()
            torch.cuda.synchronize()
            elapsed = self.start.elapsed_time(self.end)
            all_times.append(elapsed)
            print(f"{self.name} {elapsed:.03f}")


def coords_grid_with_index(d, **kwargs):
    """coordinate grid with frame index (dpvo/utils.py:39-55)"""
    b, n, h, w = d.shape
    x = torch.arange(0, w, dtype=torch.float, **kwargs)
    y = torch.arange(0, h, dtype=torch.float, **kwargs)
    y, x = torch.meshgrid(y, x, indexing="ij")
    y = y.view(1, 1, h, w).repeat(b, n, 1, 1)
    x = x.view(1, 1, h, w).repeat(b, n, 1, 1)
    coords = torch.stack([x, y, d], dim=2)
    index = torch.arange(0, n, dtype=torch.float, **kwargs)
    index = index.view(1, n, 1, 1, 1).repeat(b, 1, 1, h, w)
    return coords, index


def flatmeshgrid(*args, **kwargs):
    grid = torch.meshgrid(*args, **kwargs)
    return (x.reshape(-1) for x in grid)
