#!/usr/bin/env python
"""dpvo_gba_solve (dpvo_amd/csrc/chol.hip) alone: HIP-event time per solve at the sizes of the global BA (n = 6 N free poses) and the
relative error against an f64 solve.  Dev tool; under `rocprofv3 --kernel-trace --stats` it gives the per-kernel table."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpvo_amd import _lib as L
dev = torch.device("cuda:0")
for n in [int(x) for x in os.environ.get("NS", "294,630,1194,2394,4794").split(",")]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n + 8)); S = (A @ A.T / (n + 8) + np.eye(n)).astype(np.float32); y = rng.standard_normal(n).astype(np.float32)
    Sd, yd = torch.from_numpy(S).to(dev), torch.from_numpy(y).to(dev)
    x = torch.empty(n, device=dev)
    nbytes = L.lib().dpvo_gba_solve_workspace_bytes(L.i32(n)); ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    run = lambda: L.check(L.lib().dpvo_gba_solve(L.ptr(Sd), L.ptr(yd), L.i32(n), L.ptr(x), L.ptr(ws), ctypes.c_size_t(nbytes), L.stream()), "solve")
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10; s.record()
    for _ in range(reps): run()
    e.record(); torch.cuda.synchronize()
    S64 = S.astype(np.float64); d = np.diag(S).astype(np.float32); S64[np.arange(n), np.arange(n)] = (d + (d * np.float32(1e-4) + np.float32(1.0))).astype(np.float64)
    ref = np.linalg.solve(S64, y.astype(np.float64))
    rel = np.linalg.norm(x.cpu().numpy() - ref) / np.linalg.norm(ref)
    print(f"n = {n:5d} ({(n + 63) // 64:3d} panels): {s.elapsed_time(e) / reps * 1e3:8.1f} us per solve, |x - x64| / |x64| = {rel:.2e}")
