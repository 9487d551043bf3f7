"""Device-resident patch-graph plan (`dpvo_plan_build`): the sorted/grouped index structures that the
reference recomputes with host round trips every update -- fastba.neighbors (dpvo/fastba/ba.cpp:59-97),
torch::_unique in cuda_ba (dpvo/fastba/ba_cuda.cu:447-449) and torch.unique in SoftAgg (dpvo/blocks.py:41).
"""
import ctypes


import torch

from . import _lib as L
from . import workspace

_PLAN_WIDE = True       # tools/lc_ab.py sets it False for measurements against the radix build


class GraphPlan:
    """All views are int32 device tensors into one buffer; `counts` = [n_patches, n_pairs, 0, 0] stays on the
    device (no synchronisation); `n_patches()` / `n_pairs()` synchronise and are for tests / host logic only."""

    def __init__(self, ii, jj, kk, n_patches_ub=None, n_pairs_ub=None, n_frames=0, n_patch_ids=0, window=None, flow_pair=None,
                 wide=None):
        L.require_cuda(ii, jj, kk)
        assert ii.dtype == jj.dtype == kk.dtype == torch.long
        E = ii.numel()
        assert jj.numel() == E and kk.numel() == E
        self.E = E
        lay = L.plan_layout(E)
        self.buf = torch.empty(lay.total_ints, dtype=torch.int32, device=ii.device)
        self._lay = lay         # the views (perm_k, ku, kx, patch_off, ix, jx, perm_p, pu, pair_off, pair_ij, counts) are
        #                         created on first use: the hot path only hands `buf` to the C ABI
        ii, jj, kk = ii.contiguous(), jj.contiguous(), kk.contiguous()
        self._keep = (ii, jj, kk)
        nbytes = L.lib().dpvo_plan_workspace_bytes(L.i64(E))
        # wide = (n_frames, n_patch_ids): the caller guarantees ii, jj < n_frames and kk < n_patch_ids with n_frames of the order of the
        # tracker's frame count (not BUFFER_SIZE) -> bins-in-memory counting build (dpvo_plan_build_wide: 6 launches instead of the
        # radix build's 16-48); ranges it does not take fall back to the radix build
        wide_bytes = 0
        if wide is not None and window is None and E > 0 and _PLAN_WIDE:
            wide_bytes = L.lib().dpvo_plan_wide_workspace_bytes(L.i64(E), L.i64(wide[0]), L.i64(wide[1]))
        ws = workspace.get(max(nbytes, wide_bytes), ii.device, "plan")
        # window = (frame_lo, n_frames_win, patch_lo, n_patches_win): the caller guarantees that all frame / patch ids lie in
        # these windows -> counting-sort build (dpvo_plan_build_window); falls back when the windows are too large for it
        rc = -2
        if window is not None:
            # flow_pair = (qi, qj): the edges qi -> qj and qj -> qi are listed in the plan's `flow` region on the way (what the
            # keyframe flow test of dpvo.py:257-270 reads; dpvo_frame_update does this for its own plan)
            qi, qj = flow_pair if flow_pair is not None else (-1, -1)
            rc = L.lib().dpvo_plan_build_window_flow(L.ptr(ii), L.ptr(jj), L.ptr(kk), L.i64(E), L.ptr(self.buf), L.ptr(ws),
                                                     ctypes.c_size_t(ws.numel()), L.i64(window[0]), L.i64(window[1]), L.i64(window[2]),
                                                     L.i64(window[3]), L.i64(qi), L.i64(qj), L.stream())
            if rc != -2:
                L.check(rc, "dpvo_plan_build_window")
        self.wide = False
        if rc == -2 and wide_bytes:
            rc = L.lib().dpvo_plan_build_wide(L.ptr(ii), L.ptr(jj), L.ptr(kk), L.i64(E), L.ptr(self.buf), L.ptr(ws),
                                              ctypes.c_size_t(ws.numel()), L.i64(wide[0]), L.i64(wide[1]), L.stream())
            if rc != -2:
                L.check(rc, "dpvo_plan_build_wide")
                self.wide = True
        # n_frames / n_patch_ids: optional bounds on the index values (BUFFER_SIZE, BUFFER_SIZE * PATCHES_PER_FRAME): 32-bit keys
        if rc == -2:
            L.check(L.lib().dpvo_plan_build_ranged(L.ptr(ii), L.ptr(jj), L.ptr(kk), L.i64(E), L.ptr(self.buf), L.ptr(ws),
                                                   ctypes.c_size_t(ws.numel()), L.i64(n_frames), L.i64(n_patch_ids),
                                                   L.stream()), "dpvo_plan_build")

        # Launch sizes of the group-level kernels.  Callers that can bound the group counts from their own bookkeeping
        # (DPVO: patches / frame pairs inside the removal window) pass upper bounds and NO host synchronisation happens:
        # the kernels read the exact counts from `counts` on the device and surplus blocks exit.  Otherwise one
        # read-back per plan (the reference synchronises ~10x per update for the same information).
        if n_patches_ub is not None and n_pairs_ub is not None:
            self.n_patches_host, self.n_pairs_host = int(min(max(n_patches_ub, 1), max(E, 1))), int(min(max(n_pairs_ub, 1), max(E, 1)))
            self.exact = False
        else:
            c = self.counts[:2].tolist()
            self.n_patches_host, self.n_pairs_host = int(c[0]), int(c[1])
            self.exact = True

    _VIEWS = {"perm_k": 1, "ku": 1, "kx": 1, "patch_off": (1, 1), "ix": 1, "jx": 1, "perm_p": 1, "pu": 1, "pair_off": (1, 1),
              "pair_ij": 2}

    def __getattr__(self, name):
        if name == "counts":
            v = self.buf[self._lay.counts:self._lay.counts + 4]
        elif name == "flow":         # {qi, qj, n_ij, n_ji} + 2 x 256 patch ids (dpvo_plan_layout_t.flow)
            v = self.buf[self._lay.flow:self._lay.flow + 4 + 2 * 256]
        elif name in GraphPlan._VIEWS:
            n, mul = max(self.E, 1), GraphPlan._VIEWS[name]
            cnt = n + 1 if isinstance(mul, tuple) else mul * n
            off = getattr(self._lay, name)
            v = self.buf[off:off + cnt]
        else:
            raise AttributeError(name)
        setattr(self, name, v)
        return v

    def n_patches(self):
        return self.n_patches_host if self.exact else int(self.counts[0].item())

    def n_pairs(self):
        return self.n_pairs_host if self.exact else int(self.counts[1].item())
