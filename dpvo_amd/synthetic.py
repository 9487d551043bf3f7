"""Seeded synthetic workloads for the DPVO hot path.

Everything is generated on the CPU with a fixed ``torch.Generator`` so that tests, bench and the
CPU oracle see bit-identical inputs on every machine; callers move tensors to the device.

* ``replay_graph``  -- replays the reference's edge bookkeeping (dpvo/dpvo.py:362-375 edges_forw /
  edges_back, :305 removal) for ``n`` frames without keyframe drops.  ``n=40`` with default.yaml
  gives the steady state ``E = 45 312`` edges, 2 112 patches with edges, 472 frame pairs.
* ``make_scene``    -- poses / patches / intrinsics of a smooth trajectory over a random depth field.
* ``make_features`` -- fp16 feature ring buffers (gmap, two pyramid levels, imap).
"""
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class GraphCfg:
    M: int = 96                 # PATCHES_PER_FRAME (config/default.yaml:4)
    REMOVAL_WINDOW: int = 22    # :5
    OPTIMIZATION_WINDOW: int = 10
    PATCH_LIFETIME: int = 13    # :7
    mem: int = 36               # dpvo/dpvo.py:58
    pmem: int = 36


def replay_graph(n_frames=40, cfg=None):
    """Returns (ii, jj, kk) int64 CPU tensors in the exact order the reference would hold them."""
    cfg = cfg or GraphCfg()
    M, r = cfg.M, cfg.PATCH_LIFETIME
    ii = torch.zeros(0, dtype=torch.long)
    jj = torch.zeros(0, dtype=torch.long)
    kk = torch.zeros(0, dtype=torch.long)
    for n in range(1, n_frames + 1):
        # edges_forw (dpvo.py:362-368): patches of frames n-r..n-2 -> frame n-1
        t0, t1 = M * max(n - r, 0), M * max(n - 1, 0)
        k1 = torch.arange(t0, t1)
        j1 = torch.full_like(k1, n - 1)
        # edges_back (dpvo.py:370-375): patches of frame n-1 -> frames max(n-r,0)..n-1, kk-major
        k2 = torch.arange(M * max(n - 1, 0), M * n).repeat_interleave(n - max(n - r, 0))
        j2 = torch.arange(max(n - r, 0), n).repeat(M)
        for k, j in ((k1, j1), (k2, j2)):
            kk = torch.cat([kk, k]); jj = torch.cat([jj, j]); ii = torch.cat([ii, k // M])
        # removal (dpvo.py:305): source frame < n - REMOVAL_WINDOW
        keep = ~(ii < n - cfg.REMOVAL_WINDOW)
        ii, jj, kk = ii[keep], jj[keep], kk[keep]
    return ii, jj, kk


def _quat_mul(a, b):
    ax, ay, az, aw = a.unbind(-1); bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx,
                        aw * bw - ax * bx - ay * by - az * bz], -1)


def make_scene(n_frames=40, M=96, P=3, ht=120, wd=160, seed=1234, noise=0.02, buffer=None):
    """poses [N,7] (world->camera, t then q_xyzw), patches [N*M,3,P,P], intrinsics [N,4]; float32."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    N = buffer or n_frames
    t = torch.arange(N, dtype=torch.float64)
    # smooth trajectory: 2 cm/frame along x, small sinusoidal y/z, +-0.2 deg/frame yaw
    trans = torch.stack([0.02 * t, 0.01 * torch.sin(0.2 * t), 0.005 * t], -1)
    yaw = torch.deg2rad(torch.tensor(0.2, dtype=torch.float64)) * t
    q = torch.stack([torch.zeros_like(t), torch.sin(yaw / 2), torch.zeros_like(t), torch.cos(yaw / 2)], -1)
    xi = noise * torch.randn(N, 6, generator=g, dtype=torch.float64)
    # small perturbation quaternion (first order) applied on the left
    dq = torch.cat([0.5 * xi[:, 3:], torch.ones(N, 1, dtype=torch.float64)], -1)
    dq = dq / dq.norm(dim=-1, keepdim=True)
    q = _quat_mul(dq, q)
    q = q / q.norm(dim=-1, keepdim=True)
    poses = torch.cat([trans + xi[:, :3], q], -1).float()
    poses[0] = torch.tensor([0, 0, 0, 0, 0, 0, 1.0])

    cx = torch.randint(1, wd - 1, (N * M,), generator=g).float()
    cy = torch.randint(1, ht - 1, (N * M,), generator=g).float()
    d = 0.2 + 1.3 * torch.rand(N * M, generator=g)
    off = torch.arange(P, dtype=torch.float32) - P // 2
    px = cx[:, None, None] + off[None, None, :].expand(1, P, P)
    py = cy[:, None, None] + off[None, :, None].expand(1, P, P)
    pd = d[:, None, None].expand(-1, P, P)
    patches = torch.stack([px.expand(-1, P, P), py.expand(-1, P, P), pd], 1).contiguous()
    intr = torch.tensor([wd / 2.0, wd / 2.0, wd / 2.0, ht / 2.0]).repeat(N, 1)   # tartan/4 = (80,80,80,60)
    return poses, patches, intr


def make_features(n_patches_slots=3456, mem=36, C=128, DIM=384, P=3, ht=120, wd=160, seed=1234,
                  dtype=torch.float16):
    """gmap [slots,C,P,P], fmap1 [mem,C,ht,wd], fmap2 = avg_pool2d(fmap1,4) (dpvo.py:437-438), imap [slots,DIM].

    Returned in the reference's logical (NCHW) shapes; callers choose the memory layout."""
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    gmap = (torch.randn(n_patches_slots, C, P, P, generator=g) / 4).to(dtype)
    fmap1 = (torch.randn(mem, C, ht, wd, generator=g) / 4).to(dtype)
    fmap2 = torch.nn.functional.avg_pool2d(fmap1.float(), 4, 4).to(dtype)
    imap = torch.randn(n_patches_slots, DIM, generator=g).to(dtype)
    return gmap, fmap1, fmap2, imap


def make_coords(E, P=3, ht=120, wd=160, seed=1234, oob_frac=0.02):
    """coords [E,2,P,P] float32: integer centre + unit pixel grid + sub-pixel shift; a fraction fully OOB."""
    g = torch.Generator(device="cpu").manual_seed(seed + 2)
    cx = torch.randint(8, wd - 8, (E,), generator=g).float()
    cy = torch.randint(8, ht - 8, (E,), generator=g).float()
    s = torch.rand(E, 2, generator=g)
    off = torch.arange(P, dtype=torch.float32) - P // 2
    x = cx[:, None, None] + off[None, None, :] + s[:, 0, None, None]
    y = cy[:, None, None] + off[None, :, None] + s[:, 1, None, None]
    coords = torch.stack([x.expand(-1, P, P), y.expand(-1, P, P)], 1).contiguous()
    n_oob = int(E * oob_frac)
    if n_oob:
        idx = torch.randperm(E, generator=g)[:n_oob]
        coords[idx] += 10000.0
    return coords


def numpy_rng_like(seed):
    return np.random.default_rng(seed)
