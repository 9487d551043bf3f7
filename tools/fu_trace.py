#!/usr/bin/env python
"""Per-workgroup timeline of the fused update kernels (update_fused.hip built with -DFU_TRACE: `make -C dpvo_amd/csrc trace`).
Prints, per kernel, the median duration of every phase between two stamps and when workgroups start / end relative to the
first one (which shows the rounds).  Dev tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DPVO_HIP_LIB"] = os.path.join(ROOT, "dpvo_amd", "libdpvo_hip_trace.so")
sys.path.insert(0, ROOT)
import ctypes          # noqa: E402

import numpy as np     # noqa: E402
import torch           # noqa: E402

from dpvo_amd import _lib as L                # noqa: E402
from dpvo_amd import net as N                 # noqa: E402
from dpvo_amd import synthetic as S           # noqa: E402
from dpvo_amd.graph import GraphPlan          # noqa: E402

NAMES = {
    0: ("K1 corr+norm", ["first chunk staged", "GEMM 896", "to_lds + GEMM c2", "LN + relu + to_lds", "net req + GEMM c5",
                         "inp + add + LN", "img store + to_lds + rows"]),
    1: ("K2 c1", ["gather issued", "gather landed", "GEMM a", "W/img req + to_lds", "GEMM b", "round + img add/store",
                  "to_lds", "rows out"]),
    2: ("K3 c2 + fg", ["gather issued", "gather landed", "GEMM a", "W/img req + to_lds", "GEMM b", "round + img add/store",
                       "to_lds", "GEMM f", "f rows out", "GEMM g", "g rows out"]),
    3: ("K5 h + fg", ["gather issued", "gather landed", "-", "-", "GEMM h", "round + img add/store", "to_lds", "GEMM f",
                      "f rows out", "GEMM g", "g rows out"]),
    4: ("K7 h + gru + heads", ["gather landed", "GEMM h", "img add + LN0", "W req + to_lds", "GEMM gate0", "park gate + GEMM res0",
                               "to_lds + GEMM res2", "gate*res + LN1", "W req + to_lds", "GEMM gate1", "park + GEMM res0",
                               "to_lds + GEMM res2", "gate*res", "net out + heads"]),
}


def main():
    dev = torch.device("cuda:0")
    i0, j0, k0 = S.replay_graph(40)
    cfg = S.GraphCfg(); n = 41; M, r = cfg.M, cfg.PATCH_LIFETIME
    k1 = torch.arange(M * (n - r), M * (n - 1)); j1 = torch.full_like(k1, n - 1)
    k2 = torch.arange(M * (n - 1), M * n).repeat_interleave(r); j2 = torch.arange(n - r, n).repeat(M)
    kk = torch.cat([k0, k1, k2]).to(dev); jj = torch.cat([j0, j1, j2]).to(dev); ii = kk // M
    E = ii.numel()
    torch.manual_seed(0)
    upd = N.Update(3).to(dev); upd.pack()
    upd.tiling = int(os.environ.get("TILING", "-1"))         # (this tool's own switch: dpvo_update_fused_params_t.tiling)
    plan = GraphPlan(ii, jj, kk)
    g = torch.Generator().manual_seed(1)
    imap = torch.randn(3456, 384, generator=g).half().to(dev)
    corr = torch.zeros(E, 896, dtype=torch.float16, device=dev); corr[:, :882] = torch.randn(E, 882, generator=g).half().to(dev)
    net = torch.randn(1, E, 384, generator=g).to(dev)
    run = lambda: upd(net, imap[None], corr[None], None, ii, jj, kk, plan=plan, inp_rows=kk, inp_mod=3456, corr_is_padded=True, fused=True)
    setter = L.lib().dpvo_debug_fu_trace_buffer
    for _ in range(3):
        run()
    buf = torch.zeros(8 * 1024 * 4 * 16, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert setter(ctypes.c_void_p(buf.data_ptr())) == 0
    run()
    torch.cuda.synchronize()
    setter(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(8, 1024, 4, 16).astype(np.int64)
    nb = (E + 95) // 96
    for k, (name, phases) in NAMES.items():
        nb = min(nb, 1024)
        a = t[k, :nb]                                  # [block, wave, stamp]
        used = [i for i in range(16) if (a[:, 0, i] != 0).any()]
        if not used:
            continue
        t0 = a[:, :, 0][a[:, :, 0] > 0].min()
        start = (a[:, 0, 0] - t0) / 100.0             # us
        end = (a[:, 0, used[-1]] - t0) / 100.0
        print(f"== {name}: {nb} workgroups; starts: median {np.median(start):.1f} us, 2nd half from {np.sort(start)[nb // 2]:.1f} us; "
              f"last end {end.max():.1f} us; median workgroup lifetime {np.median(end - start):.1f} us")
        for w in (0,):
            prev = used[0]
            for i in used[1:]:
                d = (a[:, w, i] - a[:, w, prev]) / 100.0
                first = d[np.argsort(start)[:nb // 2]]; second = d[np.argsort(start)[nb // 2:]]
                label = phases[i - 1] if i - 1 < len(phases) else "?"
                print(f"   [{prev:2d}->{i:2d}] {label:28s} median {np.median(d):7.2f} us   (1st round {np.median(first):7.2f}, 2nd round {np.median(second):7.2f}, max {d.max():7.2f})")
                prev = i


if __name__ == "__main__":
    main()
