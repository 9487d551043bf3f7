#!/usr/bin/env python
"""Micro-benchmark of dpvo_linear at the update operator's shapes (E = 47 712 rows).  Dev tool (rocprofv3 --pmc target)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import net as N

dev = torch.device("cuda:0")
E = 47712
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(0)
A = torch.randn(E, 384, generator=g).half().to(dev)
A9 = torch.randn(E, 896, generator=g).half().to(dev)
W = (torch.randn(384, 384, generator=g) / 20).half().to(dev)
W9 = (torch.randn(384, 896, generator=g) / 30).half().to(dev)
W7 = (torch.randn(768, 384, generator=g) / 20).half().to(dev)
b = torch.zeros(384).half().to(dev); b7 = torch.zeros(768).half().to(dev)
out = torch.empty(E, 384, dtype=torch.float16, device=dev); out7 = torch.empty(E, 768, dtype=torch.float16, device=dev)
def t(fn, flops):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return ms, flops / ms / 1e9
out32 = torch.zeros(E, 384, dtype=torch.float32, device=dev); gate = torch.rand(E, 384, generator=g).half().to(dev)
o16 = torch.empty(E, 384, dtype=torch.float16, device=dev)
for name, fn, fl in (("384x384 f16", lambda: N.linear(A, W, b, out=out), 2 * E * 384 * 384),
                     ("384x384 relu", lambda: N.linear(A, W, b, out=out, epilogue=N.EPI_RELU), 2 * E * 384 * 384),
                     ("384x384 resadd", lambda: N.linear(A, W, b, out=out32, epilogue=N.EPI_RESADD, out16=o16), 2 * E * 384 * 384),
                     ("384x384 gated", lambda: N.linear(A, W, b, out=out32, epilogue=N.EPI_GATED, gate=gate, out16=o16), 2 * E * 384 * 384),
                     ("896x384 f16", lambda: N.linear(A9, W9, b, out=out, K=896), 2 * E * 896 * 384),
                     ("384x768 f16", lambda: N.linear(A, W7, b7, out=out7), 2 * E * 384 * 768)):
    ms, tf = t(fn, fl)
    print(f"{name:14s} {ms*1e3:8.1f} us  {tf:7.1f} TFLOP/s")
