"""DPVO front-end state machine with the reference's API (dpvo/dpvo.py:20-473): `DPVO(cfg, network, ht, wd, viz)`,
`slam(tstamp, image, intrinsics)`, `slam.terminate()`, `.pg`, `.n`, `.m`, so demo.py / evaluate_*.py stay drop-in.

MI355X-first differences (results unchanged):
  * feature ring buffers are stored channels-last ([mem,H,W,128], [pmem,M,3,3,128]) -- what the MFMA correlation
    kernel reads with coalesced 16-byte fragments; `fmap1_`, `fmap2_`, `gmap_`, `pyramid`, `gmap` expose the
    reference's NCHW shapes as permuted views, so external code indexing them keeps working;
  * `update()` = 1 reproject kernel + 1 fused two-level correlation kernel + the HIP update operator + the HIP
    BA, sharing ONE device-built graph plan per frame (instead of ~150 launches and ~10 host syncs);
  * the per-edge hidden state `pg.net` is float32 from the start.
"""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L
from . import altcorr, fastba, lietorch
from . import projective_ops as pops
from .graph import GraphPlan
from .lietorch import SE3
from .net import VONet
from .patchgraph import PatchGraph
from .utils import Timer, flatmeshgrid

autocast = torch.autocast
# Hooks for tests and measurement tools: MODULE ATTRIBUTES, set after import (tests monkeypatch them, tools assign them) -- not environment
# switches.  The only environment variables this package reads are DPVO_HIP_LIB / DPVO_HIP_CMP_LIB (development builds of the two
# libraries, dpvo_amd/_lib.py; tests/test_capi.py enforces the list).  Everything rounds 2-5 switched through the environment for an
# A/B measurement is fixed at the setting that won (DESIGN.md 3.7 keeps the numbers): one C-ABI call per steady-state frame with the
# keyframe step on the device; the next frame's encoder launches held behind the update operator, the host issuing that wait late
# (_pace_hold); the result record's event recorded inside the call; sleep-then-wait on it; reproject / corr in front of the plan and the
# global BA's edge lists as views on the call-by-call path.
_CHECK_MIRROR = False       # tests: assert on every removal that the host mirror of the edge arrays equals the device arrays
_FRAME_CALL = True          # tests: False = every frame on the call-by-call path (Python-paced launches, host-side keyframe decision)
_PLAN_ASIDE = False         # tests: the plan's launches on the encoders' stream (dpvo_frame_update_t.plan_stream; measured: no gain, +300 us of host CPU)
_PLAN_OWN_STREAM = False    # ... on a third stream instead
_STAMPS = False             # tools/stream_stamps.py: stream-ordered wall-clock stamps
_HOST_TRACE = None          # tools/fu_host_trace.py: a list that collects (label, perf_counter()) around the frame call
_PROFILE_EVERY = 1          # bench.py: HIP events around the correlation / the update operator on every k-th frame only (~5 us per marker)
_PROFILE_POOL = False       # bench.py: create the pool of timing events with the frame buffers (warm-up) instead of at its first use
_MAX_SLEEP_S = 2.0e-3      # no single pacing sleep is longer than this, whatever the running mean says
_MAX_FRAME_S = 4.0e-3      # a wait longer than this is not a frame's GPU time (first frames, a paused caller): clamped in the mean


class DPVO:

    def __init__(self, cfg, network, ht=480, wd=640, viz=False, device="cuda", defer_keyframe=True, overlap_encoders=True):
        """The reference's constructor (dpvo/dpvo.py:22: cfg, network, ht, wd, viz) plus two host-level options, both ON by default since
        round 6 -- a drop-in caller (demo.py:46, evaluate_*.py) gets the pipeline bench.py times:
        defer_keyframe: resolve the keyframe decision of frame t (its one host read-back, dpvo.py:266-310) at the
        start of the call for frame t+1, after that frame's encoders have been enqueued, so that the GPU never waits
        for the host.  Same operations in the same order, same bits.  A caller that reads the tracker's state between two calls
        (`slam.pg.*`, `slam.n`, `slam.m`: the reference's viewer does) sees the CURRENT state all the same: the `pg` accessor resolves
        a pending record first (`flush()`), terminate() does too.  False: every call returns with its record resolved."""
        self.cfg = cfg
        self._busy = 0              # > 0 inside the tracker's own entry points: `pg` then hands out the patch graph without flushing
        self._pg = None
        self.defer_keyframe = bool(defer_keyframe)
        # overlap_encoders: run the next frame's image normalisation + encoders on a second HIP stream, so that they fill
        # the gaps of the previous frame's update / BA kernels (only useful together with defer_keyframe, which lets the
        # host get that far ahead).  The caller's image must be complete when __call__ is entered (it is read on the side
        # stream without waiting for earlier work of the current stream).
        self.overlap_encoders = bool(overlap_encoders)
        self._enc_stream = None
        self._fp_done = None
        self._img_ring = None       # _upload_image: pinned / device slots for images handed over in host memory
        self._loop_pre = None       # (pinned buffer, event) of the loop-closure candidate test the last frame call ran in its tail
        self._fs = None             # dpvo_frame_state_t, reused
        self._kf_pending = None
        self._fu = None             # buffers + dpvo_frame_update_t of the one-call frame path
        self._fu_pending = None
        self._mm_host = None
        # keyframe_override: None, or a callable(frame counter) -> bool that REPLACES the outcome of the flow test of
        # dpvo.py:266-270 (True = drop keyframe n - KEYFRAME_INDEX); the test kernel and its read-back still run.  For workloads
        # whose weights are random (bench.py --drop-every, the bookkeeping tests): the flow magnitude means nothing there.
        self.keyframe_override = None
        self._loop_pairs_total = 0  # edges ever appended from outside the tracker's own bookkeeping (bounds the pair count of the global plan)
        self._iota = None           # 0, 1, 2, ... on the device (identity gather of __run_global_BA)
        self._bound_watch = []      # plans built with host-side bounds whose exact counts are on their way back (_watch_plan_bounds)
        self._lr_active = 0         # long-range (loop-closure) edges in the active list: > 0 => update() owes a global BA (dpvo.py:348)
        self._enc_done_ev = None    # the side stream's "encoders done" events (two, alternating)
        self._rng_done_ev = None    # the side stream's "random draws done" events (two, alternating)
        self._fs_join = None        # (encoders-done event handle, speculative feature-map slot) handed to the next frame call
        self._loop_try = None       # this frame's PatchGraph.edges_loop() result when it was evaluated ahead of the frame call
        self.last_keyframe = None   # (decision, (sum_ij, count_ij, sum_ji, count_ji)) of the last resolved keyframe test
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:     # "cuda" -> "cuda:<current>": tensors carry an index, and
            self.device = torch.device("cuda", torch.cuda.current_device())   # `t.device != torch.device("cuda")` is always True
        self.load_weights(network)
        self.is_initialized = False
        self.enable_timing = False
        torch.set_num_threads(2)

        self.M = self.cfg.PATCHES_PER_FRAME
        self.N = self.cfg.BUFFER_SIZE

        self.ht = ht    # image height
        self.wd = wd    # image width

        DIM = self.DIM
        RES = self.RES

        ### state attributes ###
        self.tlist = []
        self.counter = 0

        # keep track of global-BA calls
        self.ran_global_ba = np.zeros(100000, dtype=bool)

        ht = ht // RES
        wd = wd // RES

        # dummy image for visualization
        self.image_ = torch.zeros(self.ht, self.wd, 3, dtype=torch.uint8, device="cpu")

        ### network attributes ###
        if self.cfg.MIXED_PRECISION:
            self.kwargs = kwargs = {"device": self.device, "dtype": torch.half}
        else:
            self.kwargs = kwargs = {"device": self.device, "dtype": torch.float}

        ### frame memory size ###
        self.pmem = self.mem = 36  # 32 was too small given default settings
        if self.cfg.LOOP_CLOSURE:
            self.last_global_ba = -1000  # keep track of time since last global opt
            self.pmem = self.cfg.MAX_EDGE_AGE  # patch memory

        P = self.P
        self.imap_ = torch.zeros(self.pmem, self.M, DIM, **kwargs)
        # channels-last storage, reference-shaped views
        self._gmap_cl = torch.zeros(self.pmem, self.M, P, P, 128, **kwargs)
        self.gmap_ = self._gmap_cl.permute(0, 1, 4, 2, 3)                       # [pmem, M, 128, P, P]

        self.pg = PatchGraph(self.cfg, self.P, self.DIM, self.pmem, **kwargs)

        # classic backend
        if self.cfg.CLASSIC_LOOP_CLOSURE:
            self.load_long_term_loop_closure()

        self._fmap1_cl = torch.zeros(self.mem, ht // 1, wd // 1, 128, **kwargs)
        self._fmap2_cl = torch.zeros(self.mem, ht // 4, wd // 4, 128, **kwargs)
        self.fmap1_ = self._fmap1_cl.permute(0, 3, 1, 2)[None]                  # [1, mem, 128, h, w]
        self.fmap2_ = self._fmap2_cl.permute(0, 3, 1, 2)[None]

        # feature pyramid
        self.pyramid = (self.fmap1_, self.fmap2_)

        self._plan = None          # GraphPlan of the active edge list (rebuilt when edges change)
        self._deferred_removals = 0
        self._imap_full = None
        self._corr_buf = None

        self.viewer = None
        if viz:
            self.start_viewer()

    def load_long_term_loop_closure(self):
        # classical loop closure (DBoW2 / DISK+LightGlue / pypose PGO) is out of scope
        self.cfg.CLASSIC_LOOP_CLOSURE = False
        print("WARNING: CLASSIC_LOOP_CLOSURE is not available in dpvo_amd; continuing without it")

    def load_weights(self, network):
        # load network from checkpoint file
        if isinstance(network, str):
            from collections import OrderedDict
            state_dict = torch.load(network, map_location="cpu")
            new_state_dict = OrderedDict()
            for k, v in state_dict.items():
                if "update.lmbda" not in k:
                    new_state_dict[k.replace('module.', '')] = v
            self.network = VONet()
            self.network.load_state_dict(new_state_dict)
        else:
            self.network = network

        # steal network attributes
        self.DIM = self.network.DIM
        self.RES = self.network.RES
        self.P = self.network.P

        self.network.to(self.device)
        self.network.eval()
        self.network.update.pack()
        # MIXED_PRECISION: the reference re-casts every conv weight to f16 through autocast on every frame
        # (dpvo.py:391); the encoders are cast once here and run in f16 directly (same kernels, same arithmetic).
        self._enc_half = bool(self.cfg.MIXED_PRECISION)
        self._hip_enc = None
        if self._enc_half:
            self.network.patchify.fnet.half()
            self.network.patchify.inet.half()
            from .encoders import HipEncoders
            self._hip_enc = HipEncoders(self.network.patchify.fnet, self.network.patchify.inet)

    def start_viewer(self):
        raise NotImplementedError("DPViewer (Pangolin) is out of scope; run headless")

    @property
    def poses(self):
        return self.pg.poses_.view(1, self.N, 7)

    @property
    def patches(self):
        return self.pg.patches_.view(1, self.N * self.M, 3, 3, 3)

    @property
    def intrinsics(self):
        return self.pg.intrinsics_.view(1, self.N, 4)

    @property
    def ix(self):
        return self.pg.index_.view(-1)

    @property
    def imap(self):
        return self.imap_.view(1, self.pmem * self.M, self.DIM)

    @property
    def gmap(self):
        return self._gmap_cl.view(self.pmem * self.M, self.P, self.P, 128).permute(0, 3, 1, 2)[None]

    @property
    def pg(self):
        """the patch graph (dpvo/dpvo.py:44).  Read from OUTSIDE a tracker call while a deferred keyframe record is pending, it is
        brought up to date first -- so `slam.pg.points_`, `slam.n`, `slam.m` between two calls (demo.py:52-56) mean what they mean in
        the reference whatever `defer_keyframe` is."""
        if self._busy == 0 and (self._fu_pending is not None or self._kf_pending is not None):
            self.flush()
        return self._pg

    @pg.setter
    def pg(self, val):
        self._pg = val

    @property
    def n(self):
        return self.pg.n

    @n.setter
    def n(self, val):
        self.pg.n = val

    @property
    def m(self):
        return self.pg.m

    @m.setter
    def m(self, val):
        self.pg.m = val

    def get_pose(self, t):
        if t in self.traj:
            return SE3(self.traj[t])
        t0, dP = self.pg.delta[t]
        return dP * self.get_pose(t0)

    def terminate(self):
        self.flush()
        if self.cfg.LOOP_CLOSURE:
            self.append_factors(*self.pg.edges_loop())

        for _ in range(12):
            self.ran_global_ba[self.n] = False
            self.update()
        self._check_plan_bounds(wait=True)

        """ interpolate missing poses """
        self.traj = {}
        for i in range(self.n):
            self.traj[self.pg.tstamps_[i]] = self.pg.poses_[i]

        # iterative version of the reference's recursion (get_pose), avoids Python recursion limits
        poses = []
        cache = {}
        for t in range(self.counter):
            chain = []
            s = t
            while s not in self.traj and s not in cache:
                t0, dP = self.pg.delta[s]
                chain.append((s, dP))
                s = t0
            base = SE3(self.traj[s]) if s in self.traj else cache[s]
            for s2, dP in reversed(chain):
                base = dP * base
                cache[s2] = base
            poses.append(base)
        poses = lietorch.stack(poses, dim=0)
        poses = poses.inv().data.cpu().numpy()
        tstamps = np.array(self.tlist, dtype=np.float64)

        # Poses: x y z qx qy qz qw
        return poses, tstamps

    # ------------------------------------------------------------------------------------------ hot path pieces
    def corr(self, coords, indicies=None):
        """ local correlation volume (dpvo.py:200-207): fused two-level MFMA kernel -> [1, E, 882] f16 """
        ii, jj = indicies if indicies is not None else (self.pg.kk, self.pg.jj)
        E = ii.numel()
        if self._gmap_cl.dtype == torch.float16:
            # (ii % (M * pmem), jj % mem: the fused kernel reduces the indices modulo the buffer sizes itself)
            out = altcorr.corr_pyramid(self._gmap_cl.view(self.pmem * self.M, self.P * self.P, 128), self._fmap1_cl,
                                       self._fmap2_cl, coords, ii, jj, radius=3)
            return out.unsqueeze(0)
        ii1 = ii % (self.M * self.pmem)
        jj1 = jj % (self.mem)
        corr1 = altcorr.corr(self.gmap, self.pyramid[0], coords / 1, ii1, jj1, 3)
        corr2 = altcorr.corr(self.gmap, self.pyramid[1], coords / 4, ii1, jj1, 3)
        return torch.stack([corr1, corr2], -1).view(1, E, -1)

    def reproject(self, indicies=None):
        """ reproject patch k from i -> j (dpvo.py:209-213): one fused kernel -> coords [1,E,2,P,P] """
        (ii, jj, kk) = indicies if indicies is not None else (self.pg.ii, self.pg.jj, self.pg.kk)
        return pops.transform_coords(self.poses, self.patches, self.intrinsics, ii, jj, kk)

    def append_factors(self, ii, jj, lr_count=None):
        """generic append (loop-closure edges): ii = patch ids, jj = target frames (dpvo.py:215-221).  lr_count: how many of them are
        long-range by update()'s test (PatchGraph.edges_loop knows: loop_lr_count); None = counted on the device (one read-back)"""
        src = self.ix[ii]
        self.pg.edges.append(src, jj, ii)
        self._plan = None
        # edges from outside the tracker's own bookkeeping (loop closure): no window / bound assumptions while they are active; the
        # keyframe step reports when the last of them has left the active list (_lr_active back to 0).  _lr_active is EXACTLY the
        # number of active edges with ii < n - REMOVAL_WINDOW - 1: update() decides on it instead of asking the device (dpvo.py:348)
        if lr_count is None:
            lr_count = int((src < self.n - self.cfg.REMOVAL_WINDOW - 1).sum().item()) if ii.numel() else 0
        self._lr_active += int(lr_count)
        self._loop_pairs_total += int(ii.numel())          # (an upper bound on the frame pairs these edges add to any plan)

    def append_frame_factors(self):
        """append_factors(*edges_forw) + append_factors(*edges_back) (dpvo.py:458-459) as one kernel"""
        self.pg.edges.append_frame(self.ix, self.n, self.M, self.cfg.PATCH_LIFETIME)
        self._plan = None

    def _stage_removal(self, m, store):
        """host mask -> (rem, keep) index tensors on the device (async copies through pinned memory) + keep on the host"""
        es = self.pg.edges
        rem_h, keep_h = np.flatnonzero(m), np.flatnonzero(~m)
        rem = es.stage_indices(rem_h) if store and rem_h.size else None
        return rem, es.stage_indices(keep_h), keep_h

    def remove_factors(self, m, store: bool, staged=None):
        """dpvo.py:223-238.  m: bool mask over the active edges (a device tensor as in the reference, or a host numpy
        mask computed from the edge store's mirror: then no read-back is needed; `staged` = _stage_removal(m, store)
        done ahead of time)."""
        es = self.pg.edges
        if staged is not None:
            rem, keep, keep_h = staged
        elif isinstance(m, np.ndarray):
            rem, keep, keep_h = self._stage_removal(m, store)
        else:
            rem = m.nonzero().squeeze(1) if store else None
            keep, keep_h = (~m).nonzero().squeeze(1), None
        if store and rem is not None and rem.numel():
            es.keep(keep, keep_h, also=(rem, self.pg.edges_inac))      # both gathers in one launch
        else:
            es.keep(keep, keep_h)
        self._deferred_removals += int(es.net_pending is not None)
        self._plan = None

    def _removal_mask(self, h):
        """edges falling outside the optimization window (dpvo.py:305-310), from the host mirror (ix[kk] == kk // M)"""
        to_remove = (h["kk"] // self.M) < self.n - self.cfg.REMOVAL_WINDOW
        if self.cfg.LOOP_CLOSURE:
            # ...unless they are being used for loop closure
            lc_edges = ((h["jj"] - h["ii"]) > 30) & (h["jj"] > (self.n - self.cfg.OPTIMIZATION_WINDOW))
            to_remove = to_remove & ~lc_edges
        return to_remove

    def motion_probe(self):
        """ kinda hacky way to ensure enough motion for initialization (dpvo.py:240-255) """
        kk = torch.arange(self.m - self.M, self.m, device=self.device)
        jj = self.n * torch.ones_like(kk)
        ii = self.ix[kk]

        net = torch.zeros(1, len(ii), self.DIM, dtype=torch.float, device=self.device)
        coords = self.reproject(indicies=(ii, jj, kk))
        corr = self.corr(coords, indicies=(kk, jj))
        net, (delta, weight, _) = self.network.update(
            net, self.imap, corr, None, ii, jj, kk, inp_rows=kk, inp_mod=self.M * self.pmem,
            corr_is_padded=(corr.stride(1) == 896))
        return torch.quantile(delta.norm(dim=-1).float(), 0.5)

    def motionmag(self, i, j):
        k = (self.pg.ii == i) & (self.pg.jj == j)
        ii = self.pg.ii[k]
        jj = self.pg.jj[k]
        kk = self.pg.kk[k]
        flow, _ = pops.flow_mag(self.poses, self.patches, self.intrinsics, ii, jj, kk, beta=0.5)
        return flow.mean().item()

    def _keyframe_begin(self):
        """enqueue the flow test of dpvo.py:266-269 and everything of the removal step that does not need its answer"""
        i = self.n - self.cfg.KEYFRAME_INDEX - 1
        j = self.n - self.cfg.KEYFRAME_INDEX + 1
        # m = self.motionmag(i, j) + self.motionmag(j, i): one kernel + one read-back (was 2 x ~12 launches + 2 syncs);
        # the read-back goes through pinned memory + an event, so waiting for it does not wait for later launches
        if self._mm_host is None:
            self._mm_host = [torch.empty(8, dtype=torch.float32).pin_memory() for _ in range(2)]
            self._mm_flip = 0
        host = self._mm_host[self._mm_flip]
        self._mm_flip ^= 1
        m_pending = pops.motionmag_pair(self.poses, self.patches, self.intrinsics, self.pg.ii, self.pg.jj, self.pg.kk,
                                        i, j, beta=0.5, plan=self._plan, defer=True, host_buf=host)
        # while the GPU is still busy with this frame: the removal mask of the common case (keyframe kept), on the host
        es = self.pg.edges
        h = es.host()
        to_remove = self._removal_mask(h)
        staged = self._stage_removal(to_remove, True)
        forced = None if self.keyframe_override is None else bool(self.keyframe_override(self.counter))
        # ... and the number of long-range edges that removal leaves active (what _keyframe_finish used to count AFTER the removal,
        # replaying the mirror's compaction -- three 50 k-entry gathers on the host -- between the flow test's read-back and the next
        # frame's first launch, with the GPU idle)
        lr_after = None
        if self.cfg.LOOP_CLOSURE or self._lr_active > 0:
            lr_after = int(np.count_nonzero((h["ii"] < self.n - self.cfg.REMOVAL_WINDOW) & ~to_remove))
        return m_pending, to_remove, staged, forced, lr_after

    def flush(self):
        """apply a deferred keyframe decision (no-op otherwise)"""
        self._busy += 1
        try:
            if self._kf_pending is not None:
                pending, self._kf_pending = self._kf_pending, None
                self._keyframe_finish(*pending)
            if self._fu_pending is not None:
                pending, self._fu_pending = self._fu_pending, None
                self._frame_update_finish(*pending)
            if self._bound_watch:
                self._check_plan_bounds()
        finally:
            self._busy -= 1

    def keyframe(self):
        pending = self._keyframe_begin()
        if self.defer_keyframe:
            self._kf_pending = pending
        else:
            self._keyframe_finish(*pending)

    def _keyframe_finish(self, m_pending, to_remove, staged, forced=None, lr_after=None):
        es = self.pg.edges
        m_ij, m_ji = m_pending()            # the one host read-back of the frame
        m = m_ij + m_ji
        st = getattr(m_pending, "plan_status", None)
        if st is not None and st[3] != 0 and not getattr(self, "_plan_exact", False):
            # the plan was built from bounds (frame / patch id windows, group counts) that this edge list violated: its groups
            # were clamped for the frame just done.  From now on plans are built exactly (one read-back each), and say so.
            import warnings
            warnings.warn("dpvo_amd: an edge fell outside the window the graph plan was sized for "
                          f"(counters {st}); switching to exact plans", RuntimeWarning)
            self._plan_exact = True
            self._plan = None

        drop = (m / 2 < self.cfg.KEYFRAME_THRESH) if forced is None else forced
        self.last_keyframe = (int(bool(drop)), (m_ij, 1.0, m_ji, 1.0))
        if drop:
            k = self.n - self.cfg.KEYFRAME_INDEX
            t0 = self.pg.tstamps_[k - 1]
            t1 = self.pg.tstamps_[k]

            dP = SE3(self.pg.poses_[k]) * SE3(self.pg.poses_[k - 1]).inv()
            self.pg.delta[t1] = (t0, dP)

            h = es.host()
            self.remove_factors((h["ii"] == k) | (h["jj"] == k), store=False)

            self.pg.kk[self.pg.ii > k] -= self.M
            self.pg.ii[self.pg.ii > k] -= 1
            self.pg.jj[self.pg.jj > k] -= 1
            h = es.host()                   # same renumbering on the mirror (in-place views)
            h["kk"][h["ii"] > k] -= self.M
            h["ii"][h["ii"] > k] -= 1
            h["jj"][h["jj"] > k] -= 1

            # shift the ring buffers down by one slot (the reference does this with a Python loop of
            # device-to-device copies, dpvo.py:289-299; tstamps_ is a host array)
            for i in range(k, self.n - 1):
                self.pg.tstamps_[i] = self.pg.tstamps_[i + 1]
                self.pg.colors_[i] = self.pg.colors_[i + 1]
                self.pg.poses_[i] = self.pg.poses_[i + 1]
                self.pg.patches_[i] = self.pg.patches_[i + 1]
                self.pg.intrinsics_[i] = self.pg.intrinsics_[i + 1]

                self.imap_[i % self.pmem] = self.imap_[(i + 1) % self.pmem]
                self._gmap_cl[i % self.pmem] = self._gmap_cl[(i + 1) % self.pmem]
                self._fmap1_cl[i % self.mem] = self._fmap1_cl[(i + 1) % self.mem]
                self._fmap2_cl[i % self.mem] = self._fmap2_cl[(i + 1) % self.mem]

            self.n -= 1
            self.m -= self.M
            self._plan = None
            to_remove, staged, lr_after = self._removal_mask(es.host()), None, None

        self.remove_factors(to_remove, store=True, staged=staged)
        if self.cfg.LOOP_CLOSURE or self._lr_active > 0:
            # what the device-side step reports in its result word [5]: the edges the rule of dpvo.py:307-308 kept alive.  With the
            # next frame counted (n + 1) they satisfy ii < n - REMOVAL_WINDOW - 1 of dpvo.py:348.  (Also without LOOP_CLOSURE once a
            # caller has appended old patches through append_factors(): the removal above has dropped them, the count returns to 0
            # and with it the window plan / the one-call path -- ADVICE r4: it used to stick)
            self._lr_active = lr_after if lr_after is not None else int(np.count_nonzero(es.host()["ii"] < self.n - self.cfg.REMOVAL_WINDOW))
            if _CHECK_MIRROR:
                assert self._lr_active == int(np.count_nonzero(es.host()["ii"] < self.n - self.cfg.REMOVAL_WINDOW)), "long-range count diverged"
        if _CHECK_MIRROR:       # tests: the host mirror must track the device arrays exactly
            h = es.host()
            for k in ("ii", "jj", "kk"):
                assert np.array_equal(h[k], getattr(self.pg, k).cpu().numpy()), f"edge mirror diverged ({k})"

    # ------------------------------------------------------------------------------------------ one-call frame path
    def _stamp(self, i):
        """tools/stream_stamps.py (_STAMPS): stream-ordered wall-clock stamps [frame counter % 256][8] on the current stream"""
        if _STAMPS:
            if getattr(self, "_stamp_buf", None) is None:
                self._stamp_buf = torch.zeros(256, 8, dtype=torch.int64, device=self.device)
            L.lib().dpvo_debug_stamp(ctypes.c_void_p(self._stamp_buf.data_ptr() + 8 * (8 * (self.counter % 256) + i)), L.stream())

    def _frame_call_ok(self, n=None):
        from . import net as net_mod
        n = self.n if n is None else n
        # LOOP_CLOSURE (BASELINE config 5): the one-call path serves every frame whose update() takes the LOCAL BA branch of
        # dpvo.py:351-354 -- no long-range edge active (reported by the previous keyframe step) and none appended for this frame
        return (_FRAME_CALL and self.is_initialized and self._lr_active == 0 and self._hip_enc is not None and self.P == 3
                and self._gmap_cl.dtype == torch.float16 and net_mod.FUSED_DEFAULT
                and n - self.cfg.KEYFRAME_INDEX >= 1 and self.cfg.OPTIMIZATION_WINDOW <= 20)

    def _frame_update_buffers(self):
        """persistent scratch + the two pre-filled argument blocks (one per ping-pong parity of the edge store) of
        dpvo_frame_update, sized for the edge store's capacity"""
        es = self.pg.edges
        fu = self._fu
        if fu is not None and fu["cap"] == es.cap and (es.a is fu["sets"][0] or es.a is fu["sets"][1]):
            return fu
        dev, cap, lib, cfg = self.device, es.cap, L.lib(), self.cfg
        nf = cfg.REMOVAL_WINDOW + 2
        maxg = max(nf * self.M, nf * (2 * cfg.PATCH_LIFETIME + 2))
        i32, f32 = torch.int32, torch.float32
        prof_on = _PROFILE_POOL
        nb = lambda n: torch.empty(max(int(n), 16), dtype=torch.uint8, device=dev)
        fu = {"cap": cap, "sets": (es.a, es.b),
              "coords": torch.empty(cap, 2, self.P, self.P, dtype=f32, device=dev),
              "corr": torch.zeros(cap, 896, dtype=torch.float16, device=dev),
              "delta": torch.empty(cap, 2, dtype=f32, device=dev),
              "plan": torch.empty(L.plan_layout(cap).total_ints, dtype=i32, device=dev),
              "ws_plan": nb(lib.dpvo_plan_workspace_bytes(L.i64(cap))),
              "ws_update": nb(lib.dpvo_update_fused_workspace_bytes(L.i64(cap), L.i64(maxg))),
              "ws_ba": nb(lib.dpvo_ba_workspace_bytes(L.i64(cap), L.i32(20))),
              "keep": torch.empty(cap, dtype=i32, device=dev), "rem": torch.empty(cap, dtype=i32, device=dev),
              "keep_rows": torch.empty(cap, dtype=torch.int64, device=dev), "evpos": 0,
              # timing events for bench.py's roofline legs (HIP events around the correlation kernel / the update operator
              # inside the call): created once, re-recorded in place -- creating events per frame costs the host ~20 us
              "evpool": [torch.cuda.Event(enable_timing=True) for _ in range(512)] if prof_on else None,
              "result": torch.zeros(16 + 4 + 4 * (cap // 1024 + 2), dtype=f32, device=dev),      # (zeroed once: the look-back scratch counts its own generations)
              "host": [torch.zeros(16, dtype=f32).pin_memory() for _ in range(2)],
              "dpose": torch.zeros(2, 7, dtype=f32, device=dev),
              "ev": [torch.cuda.Event() for _ in range(2)], "wake": 0.0, "side": 0.0,
              "args": [L.FrameUpdate(), L.FrameUpdate()]}
        fu["host_i"] = [h.view(torch.int32).numpy() for h in fu["host"]]       # (views of the same pinned memory)
        fu["host_f"] = [h.numpy() for h in fu["host"]]
        for e_ in (fu["evpool"] or ()):
            e_.record()                 # (creates the HIP event handle)
        dp = lambda t: t.data_ptr()
        upd = self.network.update
        fp = (upd._packed or upd.pack())["_fparams"]
        rings = ((self.pg.colors_, 0), (self.pg.poses_, 0), (self.pg.patches_, 0), (self.pg.intrinsics_, 0), (self.imap_, self.pmem),
                 (self._gmap_cl, self.pmem), (self._fmap1_cl, self.mem), (self._fmap2_cl, self.mem))
        hh, ww = self._fmap1_cl.shape[1:3]
        for par in (0, 1):
            a = fu["args"][par]
            kf = a.kf
            A, B = fu["sets"][par], fu["sets"][par ^ 1]
            # (the hidden state is not moved by the keyframe step: net / net_b stay NULL, the kept rows' old numbers go to
            #  keep_rows and the next update operator gathers them in its first kernel, dpvo_update_forward_fused_rows)
            kf.ii, kf.jj, kf.kk, kf.target, kf.weight = dp(A["ii"]), dp(A["jj"]), dp(A["kk"]), dp(A["target"]), dp(A["weight"])
            kf.ii_b, kf.jj_b, kf.kk_b, kf.target_b, kf.weight_b = dp(B["ii"]), dp(B["jj"]), dp(B["kk"]), dp(B["target"]), dp(B["weight"])
            kf.delta_pose = dp(fu["dpose"]) + 28 * par
            kf.keep_idx, kf.rem_idx, kf.result_host = dp(fu["keep"]), dp(fu["rem"]), fu["host"][par].data_ptr()
            kf.keep_rows = dp(fu["keep_rows"])
            a.index_map = dp(self.pg.index_map_)
            for r, (t, ring) in enumerate(rings):
                kf.ring[r].base, kf.ring[r].slot_bytes, kf.ring[r].ring = dp(t), t.stride(0) * t.element_size(), ring
            kf.n_ring, kf.M, kf.D = len(rings), self.M, es.D
            kf.keyframe_index, kf.removal_window, kf.loop_closure = cfg.KEYFRAME_INDEX, cfg.REMOVAL_WINDOW, int(bool(cfg.LOOP_CLOSURE))
            kf.optimization_window, kf.keyframe_thresh = cfg.OPTIMIZATION_WINDOW, cfg.KEYFRAME_THRESH
            a.poses, a.patches, a.intrinsics, a.points, a.ix = (dp(self.pg.poses_), dp(self.pg.patches_), dp(self.pg.intrinsics_),
                                                                dp(self.pg.points_), dp(self.pg.index_))
            a.gmap, a.fmap1, a.fmap2, a.imap = dp(self._gmap_cl), dp(self._fmap1_cl), dp(self._fmap2_cl), dp(self.imap_)
            a.upd = ctypes.addressof(fp)
            a.coords, a.corr, a.delta, a.plan = dp(fu["coords"]), dp(fu["corr"]), dp(fu["delta"]), dp(fu["plan"])
            a.ws_plan, a.ws_update, a.ws_ba = dp(fu["ws_plan"]), dp(fu["ws_update"]), dp(fu["ws_ba"])
            a.ws_plan_bytes, a.ws_update_bytes, a.ws_ba_bytes = fu["ws_plan"].numel(), fu["ws_update"].numel(), fu["ws_ba"].numel()
            a.result_dev = dp(fu["result"])
            a.n_buffer = self.N
            a.P, a.pmem, a.mem, a.H0, a.W0 = self.P, self.pmem, self.mem, hh, ww
            a.H1, a.W1 = self._fmap2_cl.shape[1:3]
            a.patch_lifetime, a.ba_window, a.iterations, a.lmbda, a.mm_beta = cfg.PATCH_LIFETIME, cfg.OPTIMIZATION_WINDOW, 2, 1e-4, 0.5
        fu["fparams"] = fp
        self._fu = fu
        return fu

    def _frame_update_call(self, fs=None):
        """DPVO.update() + DPVO.keyframe() (dpvo.py:328-360,266-310) of a steady-state frame as one library call (with `fs`, a
        filled dpvo_frame_state_t, also the new frame's state stores and edges in front of it); the keyframe decision is taken
        and executed on the device, its result is consumed by flush() -- at the start of the next call."""
        if _HOST_TRACE is not None: _HOST_TRACE.append(("fuc", __import__("time").perf_counter()))
        from . import net as net_mod
        from .altcorr import correlation as corr_mod
        es, inac, cfg = self.pg.edges, self.pg.edges_inac, self.cfg
        fu = self._frame_update_buffers()
        par = 0 if es.a is fu["sets"][0] else 1
        E, n = es.E, self.n
        room = min(E, 8 * self.M * cfg.PATCH_LIFETIME)
        inac.reserve(room)
        a = fu["args"][par]
        kf = a.kf
        I, o = inac.a, inac.E
        kf.ii_inac, kf.jj_inac, kf.kk_inac = I["ii"].data_ptr() + 8 * o, I["jj"].data_ptr() + 8 * o, I["kk"].data_ptr() + 8 * o
        kf.target_inac, kf.weight_inac, kf.inac_room = I["target"].data_ptr() + 8 * o, I["weight"].data_ptr() + 8 * o, room
        kf.E, kf.n = E, n
        kf.forced = -1 if self.keyframe_override is None else int(bool(self.keyframe_override(self.counter)))
        upd = self.network.update
        fu["fparams"].tiling, fu["fparams"].start_skew = upd.tiling, upd.start_skew
        a.net = es.a["net"].data_ptr()
        if es.net_pending is not None:
            a.net_rows, a.n_kept = es.net_pending[0].data_ptr(), es.net_pending[1]
        else:
            a.net_rows, a.n_kept = None, 0
        # bench.py: HIP events around the correlation kernel / the update operator.  Every hipEventRecord is a marker the stream stalls
        # on for ~5 us, so they are taken on every _PROFILE_EVERY-th frame only, and the end of the correlation doubles as the
        # start of the update operator (3 records instead of 4)
        sample = _PROFILE_EVERY <= 1 or (self.counter % _PROFILE_EVERY) == 0
        if not corr_mod.PROFILE and not net_mod.PROFILE:
            fu["evpos"] = 0                     # fresh (or no) PROFILE lists: nothing refers to the pool's events any more
        for i, lst in enumerate((corr_mod.PROFILE, net_mod.PROFILE)):
            if lst is not None and sample:
                pool = fu["evpool"]     # a pool whose handles exist already (the call re-records them in place)
                if pool is None:
                    pool = fu["evpool"] = [torch.cuda.Event(enable_timing=True) for _ in range(512)]
                if not pool[0].cuda_event:
                    for e_ in pool:
                        e_.record()
                pos = fu["evpos"]
                if pos + 2 > len(pool):         # never hand out an event a PROFILE entry still refers to: grow instead of wrapping
                    more = [torch.cuda.Event(enable_timing=True) for _ in range(256)]
                    for e_ in more:
                        e_.record()
                    pool.extend(more)
                fu["evpos"] = pos + 2
                e0, e1 = pool[pos], pool[pos + 1]
                if i == 1 and corr_mod.PROFILE is not None:
                    e0 = fu["ev_corr_end"]                      # (recorded once, read by both)
                a.ev[2 * i], a.ev[2 * i + 1] = e0.cuda_event, e1.cuda_event
                if i == 0:
                    fu["ev_corr_end"] = e1
                else:
                    fu["ev_upd_end"] = e1
                lst.append((e0, e1, E))
            elif a.ev[2 * i] or a.ev[2 * i + 1]:
                a.ev[2 * i] = a.ev[2 * i + 1] = None
        a.m = self.m
        a.ev_enc = a.fmap_spec = None
        if fs is not None and self._fs_join is not None:
            a.ev_enc, a.fmap_spec = self._fs_join
            self._fs_join = None
        if fs is not None:
            # ev_fs ("the frame state has read the encoder outputs"): what the NEXT frame's side-stream batch waits for before it
            # overwrites them -- unless that batch is held behind this call's update operator anyway (one marker less in the stream)
            a.fs, a.ev_fs, a.fs_auto = ctypes.addressof(fs), None, 1
        else:
            a.fs = a.ev_fs = None
        # the event behind the update operator: what the next frame's encoder launches are held behind (§3.7)
        if getattr(self, "_upd_done", None) is None:
            self._upd_done = torch.cuda.Event()
            self._upd_done.record()
        if a.ev[3]:                 # the profiling event behind the update operator is the same point in the stream: one record
            a.ev_update_done, self._hold_event = None, fu["ev_upd_end"]
        else:
            a.ev_update_done, self._hold_event = self._upd_done.cuda_event, self._upd_done
        # the plan's five launches go to the side stream (behind whatever the encoders have queued there) when there is one: only
        # the update operator's second kernel needs them (dpvo_frame_update_t.plan_stream)
        if _PLAN_ASIDE and self._enc_stream is not None:
            evs = fu.get("ev_plan")
            if evs is None:
                evs = fu["ev_plan"] = [torch.cuda.Event(), torch.cuda.Event()]
                for e_ in evs:
                    e_.record()         # (creates the handles)
            ps = self._enc_stream
            if _PLAN_OWN_STREAM:
                ps = fu.get("plan_stream")
                if ps is None:
                    ps = fu["plan_stream"] = torch.cuda.Stream(device=self.device)
            a.plan_stream, a.ev_plan_fork, a.ev_plan_done = ps.cuda_stream, evs[0].cuda_event, evs[1].cuda_event
        else:
            a.plan_stream = a.ev_plan_fork = a.ev_plan_done = None
        # LOOP_CLOSURE: if the NEXT frame will ask PatchGraph.edges_loop for candidates ((n' + 1) - last_global_ba >= GLOBAL_OPT_FREQ with
        # n' = n or n - 1, whichever this call's keyframe step decides: the test below is the superset), their flow test runs in the tail
        # of this call, on the device, for the frame count the decision leaves behind (dpvo_frame_update_t.loop_out) -- the next frame
        # finds the magnitudes in pinned memory instead of paying a launch and a round trip in front of its own frame call
        a.loop_out = a.ev_loop = None
        self._loop_pre = None
        if cfg.LOOP_CLOSURE and (n + 1) - self.last_global_ba >= cfg.GLOBAL_OPT_FREQ:
            lp = fu.get("loop")
            if lp is None:
                cap = 2 + max(cfg.GLOBAL_OPT_FREQ - cfg.KEYFRAME_INDEX, 0) * min(cfg.MAX_EDGE_AGE, self.N)
                lp = fu["loop"] = [(torch.zeros(cap, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(2)]
                for _, e_ in lp:
                    e_.record()         # (creates the handles)
            host_l, ev_l = lp[par]
            a.loop_out, a.ev_loop = host_l.data_ptr(), ev_l.cuda_event
            a.loop_freq, a.loop_max_age = cfg.GLOBAL_OPT_FREQ, cfg.MAX_EDGE_AGE
            self._loop_pre = (host_l, ev_l)
        self._stamp(3)
        if _HOST_TRACE is not None: _HOST_TRACE.append(("call", __import__("time").perf_counter()))
        # the event the host waits for is recorded INSIDE the call, as soon as the keyframe step's result record is final -- the
        # point cloud and the gathers that execute the decision (~36 us) then run while the host reads the record and enqueues the
        # next frame (round 3 recorded it here, behind the whole call: that stretch was an idle bubble of the same length)
        ev = fu["ev"][par]
        if not ev.cuda_event:
            ev.record()                 # (creates the handle)
        a.ev_record = ev.cuda_event
        L.check(L.lib().dpvo_frame_update(ctypes.byref(a), L.stream()), "dpvo_frame_update")
        if _HOST_TRACE is not None: _HOST_TRACE.append(("ret", __import__("time").perf_counter()))
        self._stamp(4)
        es.net_pending = None           # (gathered by the operator's first kernel, rewritten compact by its last one)
        self._plan = None
        self._fu_pending = (ev, fu["host"][par], par, n, E, __import__("time").perf_counter())

    def _pace_hold(self, hold_ev):
        """The wait of the side stream for `hold_ev` (behind the update operator of the frame in flight) costs a HIP runtime thread
        its CPU time for as long as it is PENDING (0.7 ms per frame when issued right behind the frame call), so the host issues
        it late: fu["side"] seconds after that call was enqueued.  Steered by what the host finds when it gets there -- the event
        already complete: the encoders could have started earlier, issue 60 us sooner next time; still pending: 15 us later, but
        never closer than 200 us to the expected result record -- instead of by a running mean of the frame duration, which the
        delay it causes feeds back into."""
        if self._fu is None or self._fu_pending is None:
            return
        import time
        fu = self._fu
        rest = min(self._fu_pending[5] + fu["side"] - time.perf_counter(), _MAX_SLEEP_S)
        if rest > 6e-5 and not hold_ev.query():
            time.sleep(rest - 3e-5)
        if hold_ev.query():
            fu["side"] = max(0.0, fu["side"] - 6e-5)
        else:
            fu["side"] = max(0.0, min(fu["side"] + 1.5e-5, fu["wake"] - 2e-4))

    def _frame_update_finish(self, ev, host, flip, n, E, t_enq):
        """the host side of the keyframe step whose device side dpvo_keyframe_step has already executed (dpvo.py:266-310)"""
        # The result lands when the GPU has finished the frame, ~1 ms after it was enqueued; the host gets here after ~0.3 ms.
        # Sleep through most of the expected rest (running mean of the last waits), then wait on the event: a spinning wait from
        # the start would burn a host core per tracker for nothing.
        import time
        fu = self._fu
        # Pacing without a running mean of frame durations (rounds 2-3 used one; it fed back on itself: a late wake-up or a late encoder
        # batch lengthens the very duration it is derived from).  fu["wake"] = how long after the call was enqueued this wait wakes
        # up, steered by what it finds: the record already there -> wake 80 us earlier next time; otherwise half of the time it then
        # spends in the event wait beyond a 100 us margin is added.  Nothing is learnt from a frame the caller arrived late for, and
        # no single sleep exceeds _MAX_SLEEP_S (a paused caller must not teach the tracker to sleep through frames).
        if not ev.query():
            rest = min(fu["wake"] - (time.perf_counter() - t_enq), _MAX_SLEEP_S)
            if rest > 6e-5:
                time.sleep(rest - 3e-5)
            t_wake = time.perf_counter()
            late = ev.query()
            ev.synchronize()
            if late:
                fu["wake"] = max(0.0, fu["wake"] - 8e-5)
            else:
                fu["wake"] = min(max(0.0, fu["wake"] + 0.5 * (time.perf_counter() - t_wake - 1e-4)), _MAX_FRAME_S)
        else:
            ev.synchronize()
        if _HOST_TRACE is not None: _HOST_TRACE.append(("sync", time.perf_counter()))
        # (numpy views of the pinned record: this stretch of host code runs while the GPU has nothing to do)
        hi = fu["host_i"][flip]
        decision, n_keep, n_rem, e_in, overflow = int(hi[8]), int(hi[9]), int(hi[10]), int(hi[11]), int(hi[12])
        # what the device decided on: the flow test's sums and counts for (i -> j) and (j -> i) (dpvo.py:266-270), for diagnostics
        self.last_keyframe = (decision, tuple(fu["host_f"][flip][0:4].tolist()))
        self._lr_active = int(hi[13])
        if e_in != E or overflow:
            raise L.DPVOHipError(f"dpvo_keyframe_step: inconsistent result {hi[8:13].tolist()} for E = {E}")
        if fu["host_f"][flip][7] != 0 and not getattr(self, "_plan_exact", False):
            st = [int(v) for v in host[4:8].tolist()]
            import warnings
            warnings.warn("dpvo_amd: an edge fell outside the window the graph plan was sized for "
                          f"(counters {st}); switching to exact plans", RuntimeWarning)
            self._plan_exact = True
        es, inac = self.pg.edges, self.pg.edges_inac
        if decision:
            k = n - self.cfg.KEYFRAME_INDEX
            t0 = self.pg.tstamps_[k - 1]
            t1 = self.pg.tstamps_[k]
            self.pg.delta[t1] = (t0, SE3(self._fu["dpose"][flip].clone()))
            self.pg.tstamps_[k:n - 1] = self.pg.tstamps_[k + 1:n]
            self.n -= 1
            self.m -= self.M
        es.a, es.b = es.b, es.a
        es.a["net"], es.b["net"] = es.b["net"], es.a["net"]       # the state stays where it is ...
        es.net_pending = (fu["keep_rows"], n_keep)                 # ... until the next update operator gathers it (or a reader asks)
        es.E = n_keep
        inac.E += n_rem
        es.invalidate_host()
        self._plan = None
        if _HOST_TRACE is not None: _HOST_TRACE.append(("fin", __import__("time").perf_counter()))

    def __run_global_BA(self):
        """ Global bundle adjustment
         Includes both active and inactive edges """
        # full_* = torch.cat((inactive, active)) of dpvo.py:315-319 without the five copies of the inactive store: the active edges are
        # copied behind the inactive ones in the inactive store's own buffers (free space there; one launch), the lists are views
        es, inac = self.pg.edges, self.pg.edges_inac
        Ea, Ei = es.E, inac.E
        inac.reserve(Ea)
        if self._iota is None or self._iota.numel() < Ea:
            self._iota = torch.arange(max(2 * Ea, 1 << 16), dtype=torch.int64, device=self.device)
        es.gather_into(self._iota[:Ea], inac.a, Ei, skip_net=True)
        full_ii, full_jj, full_kk = (inac.a[k][:Ei + Ea] for k in ("ii", "jj", "kk"))
        full_target, full_weight = inac.a["target"][None, :Ei + Ea], inac.a["weight"][None, :Ei + Ea]

        self.pg.normalize()
        t0 = int(self.pg.edges.host()["ii"].min()) if self.pg.edges.mirror else self.pg.ii.min().item()      # (the host mirror: no device wait)
        # The plan of active + inactive edges with BOUNDS on its group counts and on the frame range instead of read-backs (the kernels
        # read the exact counts on the device, the bounds size launches and workspaces): patches <= n M; frame pairs <= the tracker's
        # own (every frame with its 2 PATCH_LIFETIME + 2 neighbours) + every loop-closure pair ever appended; frames [0, n).  With the
        # long-range test and t0 answered from the host's bookkeeping the host does not wait for the device anywhere in a global-BA
        # frame: it used to four times, each time with the GPU idle behind it.
        E_all = int(full_ii.numel())
        ub_p = min(E_all, self.n * self.M)
        ub_g = min(E_all, self.n * (2 * self.cfg.PATCH_LIFETIME + 2) + self._loop_pairs_total + self.n)
        plan = GraphPlan(full_ii.contiguous(), full_jj.contiguous(), full_kk.contiguous(), n_patches_ub=ub_p, n_pairs_ub=ub_g,
                         n_frames=self.N, n_patch_ids=self.N * self.M, wide=(self.n + 1, (self.n + 1) * self.M))
        self._watch_plan_bounds(plan, "global BA plan (active + inactive edges)")
        if _CHECK_MIRROR:
            c = plan.counts.cpu().tolist()
            assert c[0] <= ub_p and c[1] <= ub_g, ("global plan bounds", c, ub_p, ub_g)
        fastba.BA(self.poses, self.patches, self.intrinsics,
                  full_target, full_weight, 1e-4, full_ii, full_jj, full_kk, t0, self.n, M=self.M, iterations=2,
                  eff_impl=True, plan=plan, f0=0, n_frames=self.n)
        self.ran_global_ba[self.n] = True

    def _watch_plan_bounds(self, plan, what):
        """A plan sized by BOUNDS from the host's bookkeeping instead of a read-back (long-range edges active, global BA): the kernels
        read the exact group counts on the device and launches / workspaces are sized by the bounds, so a violated bound would
        truncate silently.  The exact counts therefore follow the plan to the host through a non-blocking copy, and
        _check_plan_bounds() -- at the start of the following frames, in flush-free code, and in terminate() -- RAISES if a bound
        did not hold (this was only asserted under DPVO_CHECK_MIRROR)."""
        if plan is None or plan.exact:
            return
        # counts = {patches, frame pairs, 0, id-range flag}: [2] of the copy = the flag of the window / wide builds (an id outside the range
        # the caller promised was clamped: the plan's groups are then wrong for this call)
        host = torch.empty(3, dtype=torch.int32, pin_memory=True)
        host[:2].copy_(plan.counts[:2], non_blocking=True)
        host[2:3].copy_(plan.counts[3:4], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._bound_watch.append((host, ev, plan.n_patches_host, plan.n_pairs_host, what, self.n))

    def _check_plan_bounds(self, wait=False):
        while self._bound_watch and (wait or len(self._bound_watch) > 4 or self._bound_watch[0][1].query()):
            host, ev, ub_p, ub_g, what, n = self._bound_watch.pop(0)
            ev.synchronize()
            c = host.tolist()
            if c[0] > ub_p or c[1] > ub_g or (len(c) > 2 and c[2] != 0):
                # every later plan is built exactly (one read-back each) and the cached one is dropped BEFORE raising: a caller that
                # catches this continues with exact plans, and the flag stays up
                self._plan_exact, self._plan, self.plan_bound_violated = True, None, True
                raise L.DPVOHipError(f"dpvo_amd: {what} built at frame {n} has {c[0]} patches / {c[1]} frame pairs, more than the "
                                     f"bounds {ub_p} / {ub_g} its launches and workspaces were sized for (id-range flag {c[2] if len(c) > 2 else 0}): results since then are invalid "
                                     "(please report this; `slam._plan_exact = True` switches to exact plans)")

    def plan(self):
        """the graph plan of the active edge list (rebuilt when the edges changed)"""
        if self._plan is None or self._plan.E != self.pg.ii.numel():
            ub_p = ub_g = window = None
            exact = getattr(self, "_plan_exact", False)
            if self._lr_active > 0 and not exact:
                # long-range edges active: no window, but still bounds (sources inside the removal window + one patch / one frame pair
                # per long-range edge at most + the targets of foreign edges that are not long-range by the test): no read-back
                nf = min(self.n, self.cfg.REMOVAL_WINDOW + 2)
                E_ = int(self.pg.ii.numel())
                ub_p = min(E_, nf * self.M + self._lr_active)
                ub_g = min(E_, nf * (2 * self.cfg.PATCH_LIFETIME + 2) + self._lr_active + self.n)
            if self._lr_active == 0 and not exact:
                # every active edge has its source frame in [n - REMOVAL_WINDOW - 1, n) and its target within
                # PATCH_LIFETIME frames of the source: bounds on #patches / #frame pairs, no device read-back needed
                nf = min(self.n, self.cfg.REMOVAL_WINDOW + 2)
                ub_p = nf * self.M
                ub_g = nf * (2 * self.cfg.PATCH_LIFETIME + 2)
                # ... and every frame / patch id lies in a window of REMOVAL_WINDOW + PATCH_LIFETIME (+ slack) frames:
                # counting-sort plan build (falls back to the radix build by itself when the window is too wide for it)
                flo = max(0, self.n - (self.cfg.REMOVAL_WINDOW + self.cfg.PATCH_LIFETIME + 3))
                window = (flo, self.n - flo, flo * self.M, (self.n - flo) * self.M)
            # (no window -- long-range edges active, or exact plans: the ids are still below the frame count: wide counting build)
            self._plan = GraphPlan(self.pg.ii, self.pg.jj, self.pg.kk, n_patches_ub=ub_p, n_pairs_ub=ub_g,
                                   n_frames=self.N, n_patch_ids=self.N * self.M, window=window,
                                   wide=None if window is not None else (self.n + 1, (self.n + 1) * self.M))
            if window is None and ub_p is not None:
                self._watch_plan_bounds(self._plan, "plan of the active edges while long-range edges are active")
        return self._plan

    def update(self):
        with Timer("other", enabled=self.enable_timing):
            # reprojection and correlation first: neither reads the plan, and on this call-by-call path (initialisation, frames
            # with long-range edges active: the global-BA frames of config 5) the frame's start is paced by the host -- the plan's
            # host work (buffer, two C calls, the bound watch) then runs while the GPU is busy with the correlation instead of in
            # front of it
            coords = self.reproject()
            corr = self.corr(coords)
            plan = self.plan()
            # the hidden state, updated in place (the reference reassigns pg.net); a removal of this frame may still be pending
            # on it (EdgeStore.keep(defer_net=True)): the update operator's first kernel gathers the rows, its last one
            # writes them back in compact order
            netbuf, net_rows, n_kept, net_all = self.pg.edges.net_deferred()
            # target = coords[..., P//2, P//2] + delta.float(); pg.target / pg.weight = ...  (dpvo.py:339-343): written by
            # the heads kernel straight into the edge store
            es = self.pg.edges
            target, weight = es.view("target"), es.view("weight")
            self.network.update(
                netbuf[None], self.imap, corr, None, self.pg.ii, self.pg.jj, self.pg.kk, plan=plan,
                inp_rows=self.pg.kk, inp_mod=self.M * self.pmem, corr_is_padded=(corr.stride(1) == 896), out=netbuf,
                coords=coords.contiguous(), target_out=target, weight_out=weight,
                net_rows=None if net_rows is None else (net_rows, n_kept, net_all))
            self.pg.edges.net_written()
            lmbda = 1e-4
            target, weight = target[None], weight[None]

        with Timer("BA", enabled=self.enable_timing):
            try:
                # run global bundle adjustment if there exist long-range edges
                # (long-range edges active?  The reference asks the device, `(ii < n - REMOVAL_WINDOW - 1).any()`: a host wait for the
                #  update operator in front of the global BA's own host work.  The tracker knows: _lr_active is that count)
                long_range = self._lr_active > 0
                if _CHECK_MIRROR and self.cfg.LOOP_CLOSURE:
                    assert long_range == bool((self.pg.ii < self.n - self.cfg.REMOVAL_WINDOW - 1).any()), "long-range edge count diverged"
                if self.cfg.LOOP_CLOSURE and long_range and not self.ran_global_ba[self.n]:
                    self.__run_global_BA()
                else:
                    t0 = self.n - self.cfg.OPTIMIZATION_WINDOW if self.is_initialized else 1
                    t0 = max(t0, 1)
                    fastba.BA(self.poses, self.patches, self.intrinsics,
                              target, weight, lmbda, self.pg.ii, self.pg.jj, self.pg.kk, t0, self.n, M=self.M,
                              iterations=2, eff_impl=False, plan=plan)
            except Exception as e:      # the reference swallows everything with a bare except (dpvo.py:355-356)
                print("Warning BA failed...", repr(e))

            # points = pops.point_cloud(...); self.pg.points_[:len(points)] = points[:]  (dpvo.py:358-360), in place
            pops.point_cloud(self.poses, self.patches[:, :self.m], self.intrinsics, self.ix[:self.m], out=self.pg.points_)

    def _edges_forw(self):
        r = self.cfg.PATCH_LIFETIME
        t0 = self.M * max((self.n - r), 0)
        t1 = self.M * max((self.n - 1), 0)
        return flatmeshgrid(
            torch.arange(t0, t1, device=self.device),
            torch.arange(self.n - 1, self.n, device=self.device), indexing='ij')

    def _edges_back(self):
        r = self.cfg.PATCH_LIFETIME
        t0 = self.M * max((self.n - 1), 0)
        t1 = self.M * max((self.n - 0), 0)
        return flatmeshgrid(torch.arange(t0, t1, device=self.device),
                            torch.arange(max(self.n - r, 0), self.n, device=self.device), indexing='ij')

    def _upload_image(self, img):
        """host uint8 [3,H,W] (any strides; the reader's HWC buffer seen through .permute(2,0,1) is the common case) -> device [3,H,W]
        contiguous, enqueued on the CURRENT stream (the encoder stream when there is one).  Three pinned and three device slots: a device
        slot is read by the frame that owns it (normalisation on this stream, patch colours on the main stream) and rewritten three
        frames later, behind that frame's _fp_done; a pinned slot is rewritten once its own copy has completed."""
        H, W = img.shape[-2:]
        r = self._img_ring
        if r is None or r["hw"] != (H, W):
            r = self._img_ring = {"hw": (H, W), "i": 0,
                                  "pin": [torch.empty(3 * H * W, dtype=torch.uint8).pin_memory() for _ in range(3)],
                                  "raw": [torch.empty(3 * H * W, dtype=torch.uint8, device=self.device) for _ in range(3)],
                                  "chw": [torch.empty(3, H, W, dtype=torch.uint8, device=self.device) for _ in range(3)],
                                  "ev": [torch.cuda.Event() for _ in range(3)]}
        k = r["i"]
        r["i"] = (k + 1) % 3
        if r["ev"][k].cuda_event:
            r["ev"][k].synchronize()
        hwc = img.stride() == (1, 3 * W, 3)                 # HWC memory behind a CHW view: copied as it lies, transposed on the device
        src = img.permute(1, 2, 0) if hwc else img
        r["pin"][k].view(src.shape).copy_(src)
        r["raw"][k].copy_(r["pin"][k], non_blocking=True)
        r["ev"][k].record()
        if not hwc:
            return r["raw"][k].view(3, H, W)
        r["chw"][k].copy_(r["raw"][k].view(H, W, 3).permute(2, 0, 1))
        return r["chw"][k]

    def __call__(self, tstamp, image, intrinsics, patch_coords=None, depth_init=None, image_ready=None):
        """slam(tstamp, image, intrinsics) -- dpvo/dpvo.py:377-473 (see _call)"""
        self._busy += 1
        try:
            return self._call(tstamp, image, intrinsics, patch_coords, depth_init, image_ready)
        finally:
            self._busy -= 1

    def _call(self, tstamp, image, intrinsics, patch_coords=None, depth_init=None, image_ready=None):
        """ track new frame (dpvo.py:377-473).  `patch_coords` / `depth_init` optionally inject the two random
        draws of the reference (patch centroids net.py:132-133, depth rand_like dpvo.py:427) for reproducible tests.
        `image_ready` (only matters with overlap_encoders, where the image is read on a second HIP stream):
          None  -- the image may still be in flight on the caller's current stream (`torch.from_numpy(img).cuda()` from pageable
                   memory returns early): the encoder stream waits for the current stream's position.  Always correct, but the
                   encoders of frame t+1 then start only after frame t's update / BA (no overlap);
          a torch.cuda.Event -- recorded by the caller behind the producer of the image (e.g. on its upload stream): the
                   encoder stream waits for that event only;
          False -- the caller guarantees the image is already resident (a pre-staged sequence): no wait. """

        if (self.n + 1) >= self.N:
            raise Exception(f'The buffer size is too small. You can increase it using "--opts BUFFER_SIZE={self.N*2}"')
        loop_pre, self._loop_pre = self._loop_pre, None         # (the previous frame call's tail: valid for THIS call's evaluation or for none)

        # image = 2 * (image[None,None] / 255.0) - 0.5, plus the f16 copy the encoders eat: one kernel.  The kernel reads raw
        # uint8 [3,H,W] on this device; anything else the reference's arithmetic would accept (CPU tensor, float image) is
        # converted here instead of being reinterpreted
        if torch.cuda.current_device() != self.device.index and self.device.index is not None:
            raise L.DPVOHipError(f"DPVO was built on {self.device} but the current device is cuda:{torch.cuda.current_device()}: "
                                 "wrap the call in torch.cuda.device(...)")
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(image)
        if image.dim() != 3 or image.shape[0] != 3:
            raise ValueError(f"image must be [3,H,W] (got {tuple(image.shape)})")
        # A uint8 image still in HOST memory (what a reader process hands over) is uploaded by the tracker itself, on the encoder
        # stream, through a pinned ring (_upload_image): the copy then overlaps the previous frame instead of draining the pipeline
        # the way the reference's own loop does (demo.py:38 `.cuda()` from pageable memory synchronises the caller's stream)
        host_img = image if (image.device.type == "cpu" and image.dtype == torch.uint8) else None
        if host_img is None and (image.dtype != torch.uint8 or image.device != self.device):
            image = image.to(self.device).clamp(0, 255).to(torch.uint8)
        if isinstance(intrinsics, torch.Tensor) and (intrinsics.dtype != torch.float32 or intrinsics.device != self.device):
            intrinsics = intrinsics.to(self.device, torch.float32)
        image_u8 = image.contiguous() if host_img is None else None
        H, W = image.shape[-2:]
        if host_img is not None:
            image_ready = False                 # (the upload is ordered on the stream that reads the image)
        hip_enc = self._hip_enc is not None and H % 16 == 0 and W % 16 == 0 and self.cfg.CENTROID_SEL_STRAT == 'RANDOM'
        side = pre_rng = fs_deferred = rng_done = None
        appended = False
        if hip_enc and self.overlap_encoders:
            if self._enc_stream is None:
                self._enc_stream = torch.cuda.Stream(device=self.device)
            side = self._enc_stream
            if self._fp_done is not None:       # the previous frame's readers of _imap_full / the encoder workspace
                side.wait_event(self._fp_done)
            hold_ev = None
            if getattr(self, "_hold_event", None) is not None and self._fu_pending is not None:
                hold_ev = self._hold_event      # (only the encoder launches wait for it: the random draws and the image normalisation run at once)
            # the caller's stream may still be producing / uploading the image (torch.from_numpy(img).cuda() from pageable
            # memory returns before the copy has landed): order the side stream behind it
            if image_ready is None:
                image_ready = torch.cuda.Event()
                image_ready.record(torch.cuda.current_stream(self.device))
            if image_ready is not False:
                side.wait_event(image_ready)
        main_stream = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(side if side is not None else main_stream):
            self._stamp(0)
            if hip_enc and side is not None and patch_coords is None and depth_init is None and self.P == 3:
                # the frame's three random draws (same generator order as the serial path) depend on nothing: issued FIRST on
                # the side stream -- behind the encoders they sat on the frame's critical path (the next frame starts when the
                # side stream is done)
                hh_, ww_ = self._fmap1_cl.shape[1:3]
                pre_rng = (torch.randint(1, ww_ - 1, size=[1, self.M], device=self.device),
                           torch.randint(1, hh_ - 1, size=[1, self.M], device=self.device),
                           torch.rand(1, self.M, 1, 1, dtype=torch.float32, device=self.device))
                for t_ in pre_rng:
                    t_.record_stream(main_stream)        # allocated on the side stream, read on the main one
                # ... and PRODUCED on the side stream: the main stream's first reader (frame-state part 1: coordinate / depth
                # patches) must wait for the draws themselves, not only for the allocator (record_stream orders nothing).  The
                # one-call path joins the encoders late (ev_enc, in front of part 2), so this event is what orders part 1
                if self._rng_done_ev is None:
                    self._rng_done_ev = [torch.cuda.Event(), torch.cuda.Event()]
                rng_done = self._rng_done_ev[self.counter & 1]
                rng_done.record(side)
            self._stamp(1)
            if host_img is not None:
                image_u8 = self._upload_image(host_img)
            img32 = torch.empty(1, 1, 3, H, W, dtype=torch.float32, device=self.device) if not self._enc_half else None
            img16 = torch.empty(1, 1, 3, H, W, dtype=torch.float16, device=self.device) if self._enc_half else None
            L.check(L.lib().dpvo_normalize_image(L.ptr(image_u8), L.ptr(img32), L.ptr(img16), L.i64(image_u8.numel()),
                                                 L.stream()), "dpvo_normalize_image")
            maps = None
            if hip_enc:
                # both encoders as MFMA launches, fmap written straight into its channels-last ring slot
                # (the reference also writes fmap1_[n % mem] before the motion probe may reject the frame, dpvo.py:437)
                slot = self._fmap1_cl[self.n % self.mem]
                if self._imap_full is None:
                    self._imap_full = torch.empty(H // 4, W // 4, self.DIM, dtype=torch.float16, device=self.device)
                if side is not None and hold_ev is not None:
                    self._pace_hold(hold_ev)
                    self._hip_enc(img16[0, 0], fmap_out=slot, imap_out=self._imap_full, hold_event=hold_ev, hold_at=0)
                else:
                    self._hip_enc(img16[0, 0], fmap_out=slot, imap_out=self._imap_full)
                self._stamp(2)
                if side is not None:
                    if self._enc_done_ev is None:
                        self._enc_done_ev = [torch.cuda.Event(), torch.cuda.Event()]      # (two, alternating: one may still be awaited)
                    enc_done = self._enc_done_ev[self.counter & 1]
                    enc_done.record(side)
        join = None
        if hip_enc:
            n_spec = self.n
            self.flush()                        # the previous frame's keyframe decision, now that the GPU has the encoders to chew on
            slot_spec = slot
            if self.n != n_spec:                # that keyframe was dropped: the new frame lives one slot lower
                slot = self._fmap1_cl[self.n % self.mem]

            def join():
                """order the main stream behind the side stream's encoders (and move the feature map if its slot changed).  The
                one-call frame path does both inside dpvo_frame_update, behind the plan and the reprojection, instead"""
                if side is not None:
                    main_stream.wait_event(enc_done)
                if slot is not slot_spec:
                    slot.copy_(slot_spec)
            maps = (slot, self._imap_full)
        self.flush()

        if _HOST_TRACE is not None: _HOST_TRACE.append(("fast", __import__("time").perf_counter()))
        fast = maps is not None and self.P == 3 and (patch_coords is None or patch_coords.numel() == 2 * self.M)
        if fast:
            # Patchifier's gathers + every per-frame state store in ONE launch (dpvo_frame_patches); the three random
            # draws are the reference's own, in its order: randint x, randint y (net.py:132-133), rand depth (dpvo.py:427)
            hh, ww = maps[0].shape[:2]
            xs = ys = cdev = None
            if pre_rng is not None:
                xs, ys, depth = pre_rng
            else:
                if patch_coords is None:
                    xs = torch.randint(1, ww - 1, size=[1, self.M], device=self.device)
                    ys = torch.randint(1, hh - 1, size=[1, self.M], device=self.device)
                else:
                    cdev = patch_coords.reshape(self.M, 2).to(device=self.device, dtype=torch.float32).contiguous()
                if depth_init is None:
                    depth = torch.rand(1, self.M, 1, 1, dtype=torch.float32, device=self.device)
                else:
                    depth = depth_init.reshape(self.M).to(device=self.device, dtype=torch.float32).contiguous()
            self.tlist.append(tstamp)
            self.pg.tstamps_[self.n] = self.counter
            intr_dev = intrinsics if (torch.is_tensor(intrinsics) and intrinsics.is_cuda and
                                      intrinsics.dtype == torch.float32 and intrinsics.is_contiguous()) else None
            if intr_dev is None:
                self.pg.intrinsics_[self.n] = intrinsics / self.RES
            n = self.n
            # steady state: patch gathers + state stores, motion model, depth median, pyramid level 1 and the new frame's
            # edges as ONE C-ABI call (dpvo_frame_state); otherwise the same entries one by one
            # LOOP_CLOSURE: PatchGraph.edges_loop() is due every GLOBAL_OPT_FREQ frames -- and on EVERY frame while it finds nothing
            # (dpvo.py:449-455: last_global_ba only moves when edges were found).  It reads poses / patches of frames the new frame
            # does not touch, so it is evaluated here, ahead of the frame call (one host round trip); loop edges found => this frame
            # takes the call-by-call path (they go in front of the frame's own edges, and update() then runs the global BA)
            self._loop_try = None
            if self.cfg.LOOP_CLOSURE and self.is_initialized and (n + 1) - self.last_global_ba >= self.cfg.GLOBAL_OPT_FREQ:
                self._loop_try = self.pg.edges_loop(n=n + 1, pre=loop_pre)
            loop_found = self._loop_try is not None and self._loop_try[0].numel() > 0
            # (also while long-range edges are active: the frame then takes the call-by-call path -- _frame_call_ok() is false -- but its
            #  state stores and its own edges are still the one dpvo_frame_state call; only a frame that appends loop edges, which go in
            #  FRONT of its own, issues the entries one by one)
            composite = (self.is_initialized and not loop_found and True
                         and self.cfg.MOTION_MODEL == 'DAMPED_LINEAR' and n > 1 and 3 * self.M * self.P * self.P <= 4096)
            fac = None
            if n > 1 and self.cfg.MOTION_MODEL == 'DAMPED_LINEAR':
                a, b, c = ([1, 1, 1] + self.tlist[-3:])[-3:]          # (the last three time stamps, padded with 1 like the reference's [1]*3 + tlist)
                fac = (c - b) / (b - a)
            if _HOST_TRACE is not None: _HOST_TRACE.append(("comp", __import__("time").perf_counter()))
            if composite:
                es = self.pg.edges
                total = es.frame_edge_count(n + 1, self.M, self.cfg.PATCH_LIFETIME)
                es.reserve(total)
                fs = self._fs
                if fs is None:
                    fs = self._fs = L.FrameState()
                dp = lambda t: None if t is None else t.data_ptr()
                rp = lambda t, i: t.data_ptr() + int(i) * t.stride(0) * t.element_size()
                fast_call = self._frame_call_ok(n + 1) and not getattr(self, "_plan_exact", False)
                fs.imap, fs.img_u8, fs.coords = dp(maps[1]), dp(image_u8), dp(cdev)
                fs.xs, fs.ys, fs.depth, fs.intrinsics = dp(xs), dp(ys), dp(depth), dp(intr_dev)
                fs.mm_scale, fs.res = self.cfg.MOTION_DAMPING * fac, self.RES
                fs.H, fs.W, fs.CF, fs.CI = H, W, 128, self.DIM
                if not fast_call:       # (dpvo_frame_update derives everything that depends on the frame number itself)
                    fs.fmap = dp(maps[0])
                    fs.gmap_slot, fs.imap_slot = rp(self._gmap_cl, n % self.pmem), rp(self.imap_, n % self.pmem)
                    fs.patches_slot, fs.colors_slot = rp(self.pg.patches_, n), rp(self.pg.colors_, n)
                    fs.intrinsics_slot = rp(self.pg.intrinsics_, n) if intr_dev is not None else None
                    fs.index_row, fs.index_map = rp(self.pg.index_, n + 1), rp(self.pg.index_map_, n + 1)
                    fs.poses, fs.mm_n = dp(self.pg.poses_), n
                    fs.patches_all, fs.md_n = dp(self.pg.patches_), n
                    fs.fmap2_slot = rp(self._fmap2_cl, n % self.mem)
                    # (no in-place zeroing of the new state rows while a deferred compaction is pending: live rows may still sit there)
                    fs.ii, fs.jj, fs.kk, fs.ix = dp(es.a["ii"]), dp(es.a["jj"]), dp(es.a["kk"]), dp(self.ix)
                    fs.net = dp(es.a["net"]) if es.net_pending is None else None
                    fs.frame_next, fs.m_next, fs.E0, fs.n_new = n + 1, self.m + self.M, es.E, 0
                    fs.M, fs.h, fs.w, fs.P = self.M, hh, ww, self.P
                    fs.ap_n, fs.ap_r, fs.D = n + 1, self.cfg.PATCH_LIFETIME, es.D
                # (the one-call frame path issues it itself, in front of the plan: no Python between the two)
                fs_deferred = fs if fast_call else None
                if fs_deferred is None:
                    join()
                    L.check(L.lib().dpvo_frame_state(ctypes.byref(fs), L.stream()), "dpvo_frame_state")
                    assert fs.n_new == total
                else:
                    # the frame call waits for the encoders itself, behind its plan and reprojection (dpvo_frame_update_t.ev_enc)
                    self._fs_join = (enc_done.cuda_event if side is not None else None,
                                     slot_spec.data_ptr() if slot is not slot_spec else None)
                    if rng_done is not None:        # (normally long complete: the draws are the side stream's first work)
                        main_stream.wait_event(rng_done)
                    if side is None and slot is not slot_spec:
                        join()
                es.appended_frame(n + 1, self.M, self.cfg.PATCH_LIFETIME, total)
                self._plan = None
                appended = True
            else:
                join()
                L.check(L.lib().dpvo_frame_patches(
                    L.ptr(maps[0]), L.ptr(maps[1]), L.ptr(image_u8), L.ptr(cdev), L.ptr(xs), L.ptr(ys), L.ptr(depth),
                    L.ptr(intr_dev), L.f32(self.RES), L.row_ptr(self._gmap_cl, n % self.pmem),
                    L.row_ptr(self.imap_, n % self.pmem), L.row_ptr(self.pg.patches_, n), L.row_ptr(self.pg.colors_, n),
                    L.row_ptr(self.pg.intrinsics_, n) if intr_dev is not None else L.ptr(None),
                    L.row_ptr(self.pg.index_, n + 1), L.row_ptr(self.pg.index_map_, n + 1), L.ptr(None),
                    L.i32(self.M), L.i32(hh), L.i32(ww), L.i32(H), L.i32(W), L.i32(128), L.i32(self.DIM), L.i32(self.P),
                    L.i64(self.n + 1), L.i64(self.m + self.M), L.stream()), "dpvo_frame_patches")
                if self.n > 1:
                    if fac is not None:
                        L.check(L.lib().dpvo_motion_model(L.ptr(self.pg.poses_), L.i32(self.n),
                                                          L.f32(self.cfg.MOTION_DAMPING * fac), L.stream()), "dpvo_motion_model")
                    else:
                        self.pg.poses_[self.n] = self.poses[self.n - 1]
                if self.is_initialized:
                    if 3 * self.M * self.P * self.P <= 4096:
                        L.check(L.lib().dpvo_median_depth(L.ptr(self.pg.patches_), L.i32(self.n), L.i32(self.M), L.i32(self.P),
                                                          L.stream()), "dpvo_median_depth")
                    else:
                        self.pg.patches_[self.n, :, 2] = torch.median(self.pg.patches_[self.n - 3:self.n, :, 2])
                L.check(L.lib().dpvo_pool4_nhwc(L.ptr(maps[0]), L.row_ptr(self._fmap2_cl, self.n % self.mem), L.i32(hh), L.i32(ww),
                                                L.i32(128), L.stream()), "dpvo_pool4_nhwc")
            if self.overlap_encoders:
                if self._fp_done is None:
                    self._fp_done = torch.cuda.Event()
                if fs_deferred is None:
                    self._fp_done.record()
                elif not self._fp_done.cuda_event:
                    self._fp_done.record()              # (creates the handle; re-recorded behind dpvo_frame_state inside the call)
        else:
            if join is not None:
                join()
            fmap, gmap, imap, patches, _, coords = \
                self.network.patchify(img32 if img32 is not None else img16,
                                      patches_per_image=self.cfg.PATCHES_PER_FRAME,
                                      centroid_sel_strat=self.cfg.CENTROID_SEL_STRAT,
                                      coords=patch_coords, half=self._enc_half, images_f16=img16, return_coords=True,
                                      maps=maps)

            ### update state attributes ###
            self.tlist.append(tstamp)
            self.pg.tstamps_[self.n] = self.counter
            self.pg.intrinsics_[self.n] = intrinsics / self.RES

            # color info for visualization (clr = (clr[0,:,[2,1,0]] + 0.5) * (255.0 / 2) -> uint8): one kernel on the u8 image
            L.check(L.lib().dpvo_patch_colors(L.ptr(image_u8), L.ptr(coords[0].contiguous()), L.ptr(self.pg.colors_[self.n]),
                                              L.i32(self.M), L.i32(H), L.i32(W), L.stream()), "dpvo_patch_colors")

            self.pg.index_[self.n + 1] = self.n + 1
            self.pg.index_map_[self.n + 1] = self.m + self.M

            if self.n > 1:
                if self.cfg.MOTION_MODEL == 'DAMPED_LINEAR':
                    # To deal with varying camera hz
                    a, b, c = ([1, 1, 1] + self.tlist[-3:])[-3:]          # (the last three time stamps, padded with 1 like the reference's [1]*3 + tlist)
                    fac = (c - b) / (b - a)
                    # poses_[n] = Exp(MOTION_DAMPING * fac * Log(P1 * P2^-1)) * P1: one kernel (was ~8 lietorch launches)
                    L.check(L.lib().dpvo_motion_model(L.ptr(self.pg.poses_), L.i32(self.n),
                                                      L.f32(self.cfg.MOTION_DAMPING * fac), L.stream()), "dpvo_motion_model")
                else:
                    tvec_qvec = self.poses[self.n - 1]
                    self.pg.poses_[self.n] = tvec_qvec

            # TODO better depth initialization
            patches = patches.float()
            if depth_init is None:
                patches[:, :, 2] = torch.rand_like(patches[:, :, 2, 0, 0, None, None])
            else:
                patches[:, :, 2] = depth_init.view(1, -1, 1, 1).to(patches)
            self.pg.patches_[self.n] = patches
            if self.is_initialized:
                # s = torch.median(patches_[n-3:n,:,2]); patches[:,:,2] = s: one kernel (was sort + gather + fills)
                if 3 * self.M * self.P * self.P <= 4096:
                    L.check(L.lib().dpvo_median_depth(L.ptr(self.pg.patches_), L.i32(self.n), L.i32(self.M), L.i32(self.P),
                                                      L.stream()), "dpvo_median_depth")
                else:
                    self.pg.patches_[self.n, :, 2] = torch.median(self.pg.patches_[self.n - 3:self.n, :, 2])

            ### update network attributes ###
            self.imap_[self.n % self.pmem] = imap.squeeze()
            self.gmap_[self.n % self.pmem] = gmap.squeeze()
            # fmap1_[:, n % mem] = avg_pool2d(fmap, 1, 1); fmap2_[:, n % mem] = avg_pool2d(fmap, 4, 4): one transposing kernel
            if maps is not None:
                hh, ww = maps[0].shape[:2]
                L.check(L.lib().dpvo_pool4_nhwc(L.ptr(maps[0]), L.ptr(self._fmap2_cl[self.n % self.mem]), L.i32(hh), L.i32(ww),
                                                L.i32(128), L.stream()), "dpvo_pool4_nhwc")
            else:
                fm = fmap[0, 0].contiguous()
                L.check(L.lib().dpvo_store_features(L.ptr(fm), L.ptr(self._fmap1_cl[self.n % self.mem]),
                                                    L.ptr(self._fmap2_cl[self.n % self.mem]), L.i32(L.dtype_code(fm.dtype)),
                                                    L.i32(fm.shape[0]), L.i32(fm.shape[1]), L.i32(fm.shape[2]), L.stream()),
                        "dpvo_store_features")

        self.counter += 1
        if self.n > 0 and not self.is_initialized:
            if self.motion_probe() < 2.0:
                self.pg.delta[self.counter - 1] = (self.counter - 2, SE3.Identity(1, device=self.device)[0])
                return

        self.n += 1
        self.m += self.M

        if self.cfg.LOOP_CLOSURE:
            if self.n - self.last_global_ba >= self.cfg.GLOBAL_OPT_FREQ:
                """ Add loop closure factors """
                lii, ljj = self._loop_try if self._loop_try is not None else self.pg.edges_loop()
                self._loop_try = None
                if lii.numel() > 0:
                    self.last_global_ba = self.n
                    self.append_factors(lii, ljj, lr_count=getattr(self.pg, "loop_lr_count", None))

        # Add forward and backward factors
        if not appended:
            self.append_frame_factors()

        if self.n == 8 and not self.is_initialized:
            self.is_initialized = True

            for itr in range(12):
                self.update()

        elif self.is_initialized:
            if fs_deferred is not None or (self._frame_call_ok() and not getattr(self, "_plan_exact", False)):
                self._frame_update_call(fs_deferred)
                if not self.defer_keyframe:
                    self.flush()
            else:
                self.update()
                self.keyframe()
