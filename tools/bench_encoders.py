#!/usr/bin/env python
"""Time the two torch/MIOpen encoders (BasicEncoder4) in f16: NCHW vs channels_last, cudnn.benchmark on/off.  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd.extractor import BasicEncoder4
dev = torch.device("cuda:0")
torch.manual_seed(0)
fnet = BasicEncoder4(128, 'instance').to(dev).half().eval()
inet = BasicEncoder4(384, 'none').to(dev).half().eval()
x = torch.randn(1, 1, 3, 480, 640, device=dev).half()
def run(cl):
    xi = x
    if cl:
        xi = x.view(1, 3, 480, 640).contiguous(memory_format=torch.channels_last).view(1, 1, 3, 480, 640)
    with torch.no_grad():
        return fnet(xi), inet(xi)
def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    print("benchmark", bench, "NCHW %.3f ms" % timeit(lambda: run(False)))
for m in (fnet, inet):
    m.to(memory_format=torch.channels_last)
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    print("benchmark", bench, "channels_last %.3f ms" % timeit(lambda: run(True)))
