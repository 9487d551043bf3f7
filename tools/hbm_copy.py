#!/usr/bin/env python
"""Achievable HBM bandwidth on this box ("chip peaks must be re-measured"): device-to-device copy and a
read-only reduction over buffers far larger than the 256 MB Infinity Cache.  Prints GB/s (read+write counted for the copy)."""
import torch
dev = torch.device("cuda:0")
n = 2 * 1024 ** 3 // 4                      # 2 GiB of float32
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
dt = t(lambda: b.copy_(a))
print(f"copy 2 GiB -> 2 GiB : {2 * a.numel() * 4 / dt / 1e9:8.1f} GB/s (read + write)")
dt = t(lambda: a.sum())
print(f"read-only sum 2 GiB : {a.numel() * 4 / dt / 1e9:8.1f} GB/s")
h = torch.empty(n, dtype=torch.float16, device=dev)
dt = t(lambda: h.zero_())
print(f"write-only fill 1 GiB: {h.numel() * 2 / dt / 1e9:8.1f} GB/s")
p = torch.cuda.get_device_properties(0)
print("device:", p.name, "CUs", p.multi_processor_count, "clock MHz", getattr(p, "clock_rate", 0) / 1e3)
