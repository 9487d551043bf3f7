"""End-to-end CPU restatement of the reference's per-frame pipeline -- TEST INFRASTRUCTURE ONLY.

DPVO.__call__ / update / keyframe (dpvo/dpvo.py:328-473 of the reference) chained from the oracle's own pieces:
oracle.patchify / reproject / corr_pyramid / ba (C, float64), update_ref.update_forward (torch CPU with autocast's
rounding points), GraphRef (integer bookkeeping), the lietorch SE3 exp/log of the motion model.  Float state is stored
in float32 exactly where the reference stores it (poses_, patches_, intrinsics_; half for the feature ring buffers).

Inputs per frame are the two feature maps (the encoders have their own parity test), the patch coordinates and the
random depths; the two data-dependent decisions (motion probe, keyframe flow test) are inputs, as in GraphRef.
The GPU pipeline driven with the same inputs must produce the same trajectory up to rounding (tests/test_gpu_trajectory.py).
"""
import numpy as np
import torch

from . import patchify, reproject, corr_pyramid, ba, se3_exp, se3_log, se3_inv, se3_mul
from . import update_ref
from .graph_ref import GraphRef


def _h(x):
    """round to float16 and back (where the reference holds half tensors)"""
    return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)


class DPVORef:
    def __init__(self, sd, ht, wd, M=16, BUFFER_SIZE=256, PATCH_LIFETIME=13, REMOVAL_WINDOW=22, OPTIMIZATION_WINDOW=10,
                 KEYFRAME_INDEX=4, MOTION_DAMPING=0.5, mem=36):
        self.sd = sd                                     # Update state dict (keys without the "update." prefix)
        self.M, self.N = M, BUFFER_SIZE
        self.OW, self.MD = OPTIMIZATION_WINDOW, MOTION_DAMPING
        self.mem = self.pmem = mem
        self.h, self.w = ht // 4, wd // 4
        self.g = GraphRef(M=M, PATCH_LIFETIME=PATCH_LIFETIME, REMOVAL_WINDOW=REMOVAL_WINDOW, KEYFRAME_INDEX=KEYFRAME_INDEX,
                          BUFFER_SIZE=BUFFER_SIZE)
        self.poses = np.zeros((self.N, 7), np.float32); self.poses[:, 6] = 1.0
        self.patches = np.zeros((self.N, M, 3, 3, 3), np.float32)
        self.intr = np.zeros((self.N, 4), np.float32)
        self.imap = np.zeros((self.pmem, M, 384))                       # f16-representable values
        self.gmap = np.zeros((self.pmem, M, 128, 3, 3))
        self.fmap1 = np.zeros((self.mem, 128, self.h, self.w))
        self.fmap2 = np.zeros((self.mem, 128, self.h // 4, self.w // 4))
        self.net = np.zeros((0, 384), np.float32)
        self.tlist = []
        ys, xs = np.meshgrid(np.arange(self.h, dtype=np.float64), np.arange(self.w, dtype=np.float64), indexing="ij")
        self.grid = np.stack([xs, ys, np.ones_like(xs)], 0)             # coords_grid_with_index of ones (utils.py)

    # ---- float twins of GraphRef's structural operations -------------------------------------------------
    def _append(self, kk, jj):
        self.g._append(kk, jj)
        self.net = np.concatenate([self.net, np.zeros((kk.size, 384), np.float32)])

    def _remove(self, m, store):
        self.g._remove(m, store)
        self.net = self.net[~m]

    @property
    def n(self):
        return self.g.n

    # ---- DPVO.__call__ (dpvo.py:377-473) -----------------------------------------------------------------
    def frame(self, tstamp, fmap, imap, coords, depth, intrinsics, accept=True, drop_keyframe=False):
        g, M = self.g, self.M
        n = g.n
        self.tlist.append(tstamp)
        g.tstamps_[n] = g.counter
        self.intr[n] = (np.asarray(intrinsics, np.float32) / np.float32(4.0))
        fmap, imap = np.asarray(fmap, np.float64), np.asarray(imap, np.float64)      # [128,h,w], [384,h,w] half values
        coords = np.asarray(coords, np.float64).reshape(M, 2)
        # Patchifier.forward (net.py:136-147): bilinear gathers; outputs are half (autocast module outputs)
        gm = _h(patchify(fmap, coords, 1))                                            # [M,128,3,3]
        im = _h(patchify(imap, coords, 0)).reshape(M, 384)
        pt = patchify(self.grid, coords, 1).astype(np.float32)                        # [M,3,3,3] float
        g.index_[n + 1] = n + 1
        if n > 1:                                                                     # damped linear motion model (:410-421)
            *_, a, b, c = [1] * 3 + self.tlist
            fac = (c - b) / (b - a)
            P1, P2 = self.poses[n - 1].astype(np.float64)[None], self.poses[n - 2].astype(np.float64)[None]
            xi = self.MD * fac * se3_log(se3_mul(P1, se3_inv(P2)))
            self.poses[n] = se3_mul(se3_exp(xi), P1)[0].astype(np.float32)
        pt[:, 2] = np.asarray(depth, np.float32).reshape(M, 1, 1)                     # :427
        self.patches[n] = pt
        if g.is_initialized:                                                          # :430-432, torch.median = lower median
            v = np.sort(self.patches[n - 3:n, :, 2].reshape(-1))
            self.patches[n, :, 2] = v[(v.size - 1) // 2]
        self.imap[n % self.pmem] = im
        self.gmap[n % self.pmem] = gm
        self.fmap1[n % self.mem] = fmap
        f4 = fmap.astype(np.float32).reshape(128, self.h // 4, 4, self.w // 4, 4)
        self.fmap2[n % self.mem] = _h(f4.sum(axis=(2, 4), dtype=np.float32) / np.float32(16.0))   # avg_pool2d(fmap, 4, 4) in half
        g.counter += 1
        if n > 0 and not g.is_initialized and not accept:                             # :441-444
            g.delta[g.counter - 1] = g.counter - 2
            return "skipped"
        g.n += 1; g.m += M
        n = g.n
        t0, t1 = M * max(n - g.r, 0), M * max(n - 1, 0)                               # edges_forw (:362-368)
        k1 = np.arange(t0, t1, dtype=np.int64)
        self._append(k1, np.full_like(k1, n - 1))
        k2 = np.repeat(np.arange(M * max(n - 1, 0), M * n, dtype=np.int64), n - max(n - g.r, 0))   # edges_back (:370-375)
        j2 = np.tile(np.arange(max(n - g.r, 0), n, dtype=np.int64), M)
        self._append(k2, j2)
        if n == 8 and not g.is_initialized:
            g.is_initialized = True
            for _ in range(12):
                self.update()
            return "initialized"
        if g.is_initialized:
            self.update()
            self.keyframe(drop_keyframe)
            return "tracked"
        return "buffered"

    # ---- DPVO.update (dpvo.py:328-360) -------------------------------------------------------------------
    def update(self):
        g, M = self.g, self.M
        ii, jj, kk = g.ii, g.jj, g.kk
        pat = self.patches.reshape(-1, 3, 3, 3)
        coords = reproject(self.poses, pat, self.intr, ii, jj, kk).astype(np.float32)            # float32 tensor in the reference
        corr = _h(corr_pyramid(self.gmap.reshape(-1, 128, 3, 3), (self.fmap1, self.fmap2), coords, kk % (M * self.pmem),
                               jj % self.mem))
        inp = self.imap.reshape(-1, 384)[kk % (M * self.pmem)]
        net, delta, weight = update_ref.update_forward(self.sd, torch.from_numpy(self.net), torch.from_numpy(inp),
                                                       torch.from_numpy(corr), torch.from_numpy(ii), torch.from_numpy(jj),
                                                       torch.from_numpy(kk))
        self.net = net.numpy().astype(np.float32)
        target = (coords[:, :, 1, 1].astype(np.float32) + delta.numpy().astype(np.float32)).astype(np.float32)
        weight = weight.numpy().astype(np.float32)
        t0 = max(g.n - self.OW if g.is_initialized else 1, 1)
        poses, patches, _, _ = ba(self.poses, pat, self.intr, target, weight, 1e-4, ii, jj, kk, t0, g.n, iterations=2)
        self.poses = np.asarray(poses, np.float32).reshape(self.N, 7)
        self.patches = np.asarray(patches, np.float32).reshape(self.N, M, 3, 3, 3)

    # ---- DPVO.keyframe (dpvo.py:266-310) -----------------------------------------------------------------
    def keyframe(self, drop):
        g, M = self.g, self.M
        if drop:
            k = g.n - g.KI
            g.delta[int(g.tstamps_[k])] = int(g.tstamps_[k - 1])
            self._remove((g.ii == k) | (g.jj == k), store=False)
            g.kk[g.ii > k] -= M
            g.ii[g.ii > k] -= 1
            g.jj[g.jj > k] -= 1
            for i in range(k, g.n - 1):
                g.tstamps_[i] = g.tstamps_[i + 1]
                self.poses[i] = self.poses[i + 1]
                self.patches[i] = self.patches[i + 1]
                self.intr[i] = self.intr[i + 1]
                self.imap[i % self.pmem] = self.imap[(i + 1) % self.pmem]
                self.gmap[i % self.pmem] = self.gmap[(i + 1) % self.pmem]
                self.fmap1[i % self.mem] = self.fmap1[(i + 1) % self.mem]
                self.fmap2[i % self.mem] = self.fmap2[(i + 1) % self.mem]
            g.n -= 1; g.m -= M
        self._remove(g.ix[g.kk] < g.n - g.R, store=True)
