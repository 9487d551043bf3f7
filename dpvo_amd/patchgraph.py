"""PatchGraph: state container of the reference (dpvo/patchgraph.py:11-111) with the same attribute names.

Differences that do not change results: `net` is kept in float32 (the reference's torch.cat promotes it to
float32 after the first update anyway, dpvo.py:220 + net.py:78); index tensors stay int64 on the device.
`reduce_edges` (numba in the reference, loop_closure/optim_utils.py:23-60) is restated in plain numpy/Python."""
import numpy as np
import torch

from . import projective_ops as pops
from .lietorch import SE3
from .utils import flatmeshgrid


def reduce_edges(flow_mag, ii, jj, max_num_edges, nms):
    """Greedy NMS over candidate loop-closure edges (optim_utils.py:23-60); integer result is bit-exact with the
    reference for the same np.argsort tie-breaking."""
    es = []
    if ii.size == 0:
        return np.zeros((0, 2), dtype=np.int64)
    Ni, Nj = int(ii.max() + 1), int(jj.max() + 1)
    ignore = np.zeros((Ni, Nj), dtype=bool)
    for idx in np.argsort(flow_mag):
        if len(es) + 1 > max_num_edges:
            break
        i, j, mag = int(ii[idx]), int(jj[idx]), flow_mag[idx]
        if (j - i) < 30:
            continue
        if mag >= 1000:
            continue
        if ignore[i, j]:
            continue
        es.append((i, j))
        for di in range(-nms, nms + 1):
            i1 = i + di
            if 0 <= i1 < Ni:
                ignore[i1, j] = True
    return np.asarray(es, dtype=np.int64).reshape(-1, 2)


class PatchGraph:
    """Dataclass for storing variables"""

    def __init__(self, cfg, P, DIM, pmem, **kwargs):
        self.cfg = cfg
        self.P = P
        self.pmem = pmem
        self.DIM = DIM
        dev = kwargs.get("device", "cuda")

        self.n = 0      # number of frames
        self.m = 0      # number of patches

        self.M = self.cfg.PATCHES_PER_FRAME
        self.N = self.cfg.BUFFER_SIZE

        self.tstamps_ = np.zeros(self.N, dtype=np.int64)
        self.poses_ = torch.zeros(self.N, 7, dtype=torch.float, device=dev)
        self.patches_ = torch.zeros(self.N, self.M, 3, self.P, self.P, dtype=torch.float, device=dev)
        self.intrinsics_ = torch.zeros(self.N, 4, dtype=torch.float, device=dev)

        self.points_ = torch.zeros(self.N * self.M, 3, dtype=torch.float, device=dev)
        self.colors_ = torch.zeros(self.N, self.M, 3, dtype=torch.uint8, device=dev)

        self.index_ = torch.zeros(self.N, self.M, dtype=torch.long, device=dev)
        self.index_map_ = torch.zeros(self.N, dtype=torch.long, device=dev)

        # initialize poses to identity matrix
        self.poses_[:, 6] = 1.0

        # store relative poses for removed frames
        self.delta = {}

        ### edge information ###
        self.net = torch.zeros(1, 0, DIM, dtype=torch.float, device=dev)
        self.ii = torch.as_tensor([], dtype=torch.long, device=dev)
        self.jj = torch.as_tensor([], dtype=torch.long, device=dev)
        self.kk = torch.as_tensor([], dtype=torch.long, device=dev)
        self.weight = torch.zeros(1, 0, 2, dtype=torch.float, device=dev)
        self.target = torch.zeros(1, 0, 2, dtype=torch.float, device=dev)

        ### inactive edge information (i.e., no longer updated, but useful for BA) ###
        self.ii_inac = torch.as_tensor([], dtype=torch.long, device=dev)
        self.jj_inac = torch.as_tensor([], dtype=torch.long, device=dev)
        self.kk_inac = torch.as_tensor([], dtype=torch.long, device=dev)
        self.weight_inac = torch.zeros(1, 0, 2, dtype=torch.float, device=dev)
        self.target_inac = torch.zeros(1, 0, 2, dtype=torch.float, device=dev)

    def edges_loop(self):
        """Adding edges from old patches to new frames (patchgraph.py:56-82)"""
        lc_range = self.cfg.MAX_EDGE_AGE
        l = self.n - self.cfg.REMOVAL_WINDOW  # l is the upper bound for "old" patches
        dev = self.poses_.device
        if l <= 0:
            return torch.empty(2, 0, dtype=torch.long, device=dev)

        # create candidate edges
        jj, kk = flatmeshgrid(
            torch.arange(self.n - self.cfg.GLOBAL_OPT_FREQ, self.n - self.cfg.KEYFRAME_INDEX, device=dev),
            torch.arange(max(l - lc_range, 0) * self.M, l * self.M, device=dev), indexing='ij')
        ii = self.ix[kk]

        # Remove edges which have too large flow magnitude (centre pixel only: patches[...,1,1] in the reference)
        c = self.P // 2
        centre = self.patches[..., c, c].reshape(1, -1, 3, 1, 1)
        flow_mg, nval = pops.flow_mag(self.poses, centre, self.intrinsics, ii, jj, kk, beta=0.5)
        val = (nval > 0.5).float()
        flow_mg_sum = (flow_mg * val).view(-1, self.M).sum(dim=1).float()
        num_val = val.view(-1, self.M).sum(dim=1).clamp(min=1)
        flow_mag = torch.where(num_val > (self.M * 0.75), flow_mg_sum / num_val,
                               torch.full_like(num_val, float("inf")))

        mask = (flow_mag < self.cfg.BACKEND_THRESH)
        es = reduce_edges(flow_mag[mask].cpu().numpy(), ii[::self.M][mask].cpu().numpy(),
                          jj[::self.M][mask].cpu().numpy(), max_num_edges=1000, nms=1)

        edges = torch.as_tensor(es, device=dev).reshape(-1, 2)
        ii = edges[:, 0][:, None].expand(-1, self.M)
        jj = edges[:, 1][:, None].expand(-1, self.M)
        kk = ii.mul(self.M) + torch.arange(self.M, device=dev)
        return kk.flatten(), jj.flatten()

    def normalize(self):
        """normalize depth and poses (patchgraph.py:84-95)"""
        s = self.patches_[:self.n, :, 2].mean()
        self.patches_[:self.n, :, 2] /= s
        self.poses_[:self.n, :3] *= s
        for t, (t0, dP) in self.delta.items():
            self.delta[t] = (t0, dP.scale(s))
        self.poses_[:self.n] = (SE3(self.poses_[:self.n]) * SE3(self.poses_[[0]]).inv()).data

        points = pops.point_cloud(self.poses, self.patches[:, :self.m], self.intrinsics, self.ix[:self.m])
        self.points_[:len(points)] = points[:]

    @property
    def poses(self):
        return self.poses_.view(1, self.N, 7)

    @property
    def patches(self):
        return self.patches_.view(1, self.N * self.M, 3, 3, 3)

    @property
    def intrinsics(self):
        return self.intrinsics_.view(1, self.N, 4)

    @property
    def ix(self):
        return self.index_.view(-1)
