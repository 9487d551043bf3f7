"""PatchGraph: state container of the reference (dpvo/patchgraph.py:11-111) with the same attribute names.

Differences that do not change results: `net` is kept in float32 (the reference's torch.cat promotes it to
float32 after the first update anyway, dpvo.py:220 + net.py:78); index tensors stay int64 on the device.
`reduce_edges` (numba in the reference, loop_closure/optim_utils.py:23-60) is restated in plain numpy/Python."""
import ctypes

import numpy as np
import torch

from . import _lib as L
from . import projective_ops as pops
from .lietorch import SE3
from .utils import flatmeshgrid

_NORMALIZE_FUSED = True     # tools/lc_ab.py sets it False: PatchGraph.normalize as torch operations (measurements)
MAX_LOOP_PAIRS = 1000       # reduce_edges(max_num_edges=1000) of the reference's edges_loop (patchgraph.py:76)
_INAC_PREALLOC_FRAC = 0.02  # the inactive-edge store is preallocated from at most this fraction of the device memory that is free


class EdgeStore:
    """Preallocated per-edge arrays (ii, jj, kk int64; net f32 [.,D]; target, weight f32 [.,2]) with an element count.

    The reference rebuilds these tensors with torch.cat / boolean-mask indexing on every frame (dpvo.py:215-238):
    ~45 small launches and ~150 MB of copies.  Here appends write into the tail of capacity buffers (one kernel) and
    removals compact into a second buffer set (one nonzero + one gather kernel), then the sets swap."""

    def __init__(self, D, device, with_state=True, cap=65536, mirror=False):
        self.D, self.dev, self.with_state = D, device, with_state
        self.E = 0
        self._alloc(cap)
        # Host mirror of (ii, jj, kk) as int32 numpy arrays: the integer bookkeeping is deterministic, so the removal masks
        # of DPVO.keyframe can be computed on the host WHILE the GPU runs the update -- no nonzero() read-back.
        # _h is None when the mirror is unknown (re-read from the device on demand); _log holds the mutations that have been
        # applied to the device arrays but not yet to the mirror (applied lazily, off the critical path).
        self.mirror = mirror
        self._h = None if not mirror else {k: np.empty(cap, dtype=np.int32) for k in ("ii", "jj", "kk")}
        self._hE = 0
        self._log = []
        self._stage = None          # pinned / device index staging buffers (ring of 4) + their copy stream
        self._stage_flip = 0
        self._staged = {}           # device staging buffer -> its copy event, between stage_indices and the gather that reads it
        self._gathered = None       # recorded on the compute stream after a gather that read staged indices
        # Deferred compaction of the hidden state (`net`, 1.5 KB per edge = 95 % of what a removal moves): keep(defer_net=True)
        # compacts everything else and only REMEMBERS the keep list; the update operator's first kernel then reads row g of the
        # state from net[keep[g]] (zeros for the edges appended since) and its last kernel writes the rows back compact
        # (dpvo_update_forward_fused_rows).  Any other reader goes through view("net"), which materialises first.
        self.net_pending = None     # (keep indices [n_kept] int64 on the device, n_kept) or None

    def _new(self, cap):
        d = {k: torch.empty(cap, dtype=torch.long, device=self.dev) for k in ("ii", "jj", "kk")}
        d["target"] = torch.zeros(cap, 2, dtype=torch.float, device=self.dev)
        d["weight"] = torch.zeros(cap, 2, dtype=torch.float, device=self.dev)
        if self.with_state:
            d["net"] = torch.empty(cap, self.D, dtype=torch.float, device=self.dev)
        return d

    def _alloc(self, cap):
        self.cap = cap
        # two sets for the store that compacts (ping-pong of keep()); the inactive store only ever appends: its second set would double
        # every reallocation for nothing and is created by the first keep(), should one ever come
        self.a, self.b = self._new(cap), (self._new(cap) if self.with_state else None)

    def reserve(self, n):
        if self.E + n <= self.cap:
            return
        self.materialize_net()
        old, E = self.a, self.E
        self._alloc(max(2 * self.cap, self.E + n))
        for k, v in old.items():
            self.a[k][:E] = v[:E]
        self._stage = None
        if self._h is not None:
            for k in self._h:
                h = np.empty(self.cap, dtype=np.int32)
                h[:self._hE] = self._h[k][:self._hE]
                self._h[k] = h

    # ---- host mirror -------------------------------------------------------------------------------------------
    def host(self):
        """(ii, jj, kk) of the active edges as int32 numpy views.  Appends and compactions are only LOGGED when they happen
        (the GPU is starved for launches in that part of the frame) and replayed here, where the caller has slack; one
        device read-back only if the mirror was lost"""
        if not self.mirror:
            raise RuntimeError("EdgeStore created without a host mirror")
        if self._h is None:
            self._h = {k: np.empty(self.cap, dtype=np.int32) for k in ("ii", "jj", "kk")}
            for k in self._h:
                self._h[k][:self.E] = self.a[k][:self.E].cpu().numpy()
            self._hE, self._log = self.E, []
        for op in self._log:
            if op[0] == "keep":
                idx = op[1]
                for k in self._h:
                    self._h[k][:idx.size] = self._h[k][:self._hE][idx]
                self._hE = idx.size
            elif op[0] == "frame":
                # append_edges_kernel's edges (ix[k] == k // M: index_ rows hold their own frame number, dpvo.py:405)
                _, n, M, r = op
                E0 = self._hE
                f0, f1 = M * max(n - r, 0), M * max(n - 1, 0)
                nf = f1 - f0
                jlo = max(n - r, 0)
                nj = n - jlo
                total = nf + M * nj
                hk, hj, hi = self._h["kk"], self._h["jj"], self._h["ii"]
                hk[E0:E0 + nf] = np.arange(f0, f1, dtype=np.int32)
                hj[E0:E0 + nf] = n - 1
                hk[E0 + nf:E0 + total] = np.repeat(np.arange(f1, f1 + M, dtype=np.int32), nj)
                hj[E0 + nf:E0 + total] = np.tile(np.arange(jlo, n, dtype=np.int32), M)
                hi[E0:E0 + total] = hk[E0:E0 + total] // M
                self._hE = E0 + total
            else:
                _, ii, jj, kk = op
                n = ii.size
                for k, v in (("ii", ii), ("jj", jj), ("kk", kk)):
                    self._h[k][self._hE:self._hE + n] = v
                self._hE += n
        self._log = []
        assert self._hE == self.E
        return {k: v[:self.E] for k, v in self._h.items()}

    def invalidate_host(self):
        if self.mirror:
            self._h, self._log = None, []

    def stage_indices(self, idx_np):
        """host int64 indices -> device tensor, through pinned memory, without blocking the host.  The copy is issued on a
        dedicated copy stream, so it runs as soon as the host has the list (while the compute stream is still busy with the
        current frame) instead of queueing behind it.  Ordering: the consumer (gather_into) waits for the copy's event if it
        has not completed yet; the copy stream waits for the last compaction (`keep`) before it rewrites a staging buffer."""
        n = int(idx_np.size)
        if self._stage is None:
            mk = lambda: {"pin": torch.empty(self.cap, dtype=torch.int64).pin_memory(),
                          "dev": torch.empty(self.cap, dtype=torch.int64, device=self.dev), "copied": torch.cuda.Event()}
            self._stage = [mk() for _ in range(4)]
            self._copy_stream = torch.cuda.Stream(device=self.dev)
            self._staged = {}
        st = self._stage[self._stage_flip]
        self._stage_flip = (self._stage_flip + 1) % len(self._stage)
        st["copied"].synchronize()                   # the previous copy out of this pinned buffer (two frames ago)
        st["pin"][:n].numpy()[:] = idx_np
        cs = self._copy_stream
        if self._gathered is not None:
            cs.wait_event(self._gathered)
        with torch.cuda.stream(cs):
            st["dev"][:n].copy_(st["pin"][:n], non_blocking=True)
            st["copied"].record(cs)
        self._staged[st["dev"].data_ptr()] = st["copied"]
        return st["dev"][:n]

    def view(self, name):
        if name == "net" and self.net_pending is not None:
            self.materialize_net()
        return self.a[name][:self.E]

    def net_deferred(self):
        """(state buffer [E, D], keep indices or None, n_kept, the whole capacity buffer the indices point into) WITHOUT
        materialising: for dpvo_update_forward_fused_rows"""
        rows, n_kept = self.net_pending if self.net_pending is not None else (None, 0)
        return self.a["net"][:self.E], rows, n_kept, self.a["net"]

    def net_written(self):
        """the update operator has rewritten every state row in compact order"""
        if self.net_pending is not None:
            self.net_pending = None
            self._mark_gathered()           # (its first kernel read the staged keep list)

    def materialize_net(self):
        """apply a deferred compaction of the state now (readers other than the fused update operator)"""
        if self.net_pending is None:
            return
        rows, n_kept = self.net_pending
        self.net_pending = None
        vp = ctypes.c_void_p
        if n_kept:
            L.check(L.lib().dpvo_gather_edges(
                vp(rows.data_ptr()), ctypes.c_int64(n_kept), vp(self.a["ii"].data_ptr()), vp(self.a["jj"].data_ptr()),
                vp(self.a["kk"].data_ptr()), vp(self.a["net"].data_ptr()), vp(0), vp(0), vp(0), vp(0), vp(0),
                vp(self.b["net"].data_ptr()), vp(0), vp(0), ctypes.c_int(self.D), L.stream()), "dpvo_gather_edges")
        self.a["net"], self.b["net"] = self.b["net"], self.a["net"]
        if self.E > n_kept:
            self.a["net"][n_kept:self.E].zero_()
        self._mark_gathered()

    def assign(self, name, value):
        value = value.reshape((-1,) + tuple(self.a[name].shape[1:]))
        n = value.shape[0]
        if name in ("ii", "jj", "kk") and n != self.E:
            raise ValueError("edge arrays must be resized through EdgeStore.append / keep")
        self.a[name][:n] = value
        if name in ("ii", "jj", "kk"):
            self.invalidate_host()

    @staticmethod
    def frame_edge_count(n, M, r):
        """number of edges append_factors(edges_forw) + append_factors(edges_back) add for frame count n (dpvo.py:362-375)"""
        return M * (max(n - 1, 0) - max(n - r, 0)) + M * (n - max(n - r, 0))

    def appended_frame(self, n, M, r, total):
        """bookkeeping after the kernel of append_frame has been issued (by append_frame or inside dpvo_frame_state)"""
        if self._h is not None:
            self._log.append(("frame", n, M, r))
        self.E += total

    def append_frame(self, ix, n, M, r):
        """append_factors(edges_forw) + append_factors(edges_back) for frame count n: one kernel"""
        total = self.frame_edge_count(n, M, r)
        self.reserve(total)
        cnt = ctypes.c_int64(0)
        # (with a deferred compaction pending the new rows must not be zeroed in place: live rows may still sit there)
        L.check(L.lib().dpvo_append_edges(L.ptr(self.a["ii"]), L.ptr(self.a["jj"]), L.ptr(self.a["kk"]),
                                          L.ptr(self.a["net"] if self.net_pending is None else None), L.ptr(ix), L.i64(self.E), L.i32(n), L.i32(M), L.i32(r),
                                          L.i32(self.D), ctypes.byref(cnt), L.stream()), "dpvo_append_edges")
        assert cnt.value == total
        self.appended_frame(n, M, r, total)

    def append(self, ii, jj, kk, net=None, target=None, weight=None):
        n = ii.numel()
        self.reserve(n)
        if self.with_state and self.net_pending is not None:
            self.materialize_net()
        E = self.E
        self.a["ii"][E:E + n] = ii; self.a["jj"][E:E + n] = jj; self.a["kk"][E:E + n] = kk
        if self.with_state:
            if net is None:
                self.a["net"][E:E + n].zero_()
            else:
                self.a["net"][E:E + n] = net.reshape(n, self.D)
        for name, val in (("target", target), ("weight", weight)):
            if val is None:
                self.a[name][E:E + n].zero_()
            else:
                self.a[name][E:E + n] = val.reshape(n, 2)
        if self._h is not None and n:
            self._log.append(("edges",) + tuple(v.cpu().numpy().astype(np.int32) for v in (ii, jj, kk)))
        self.E += n

    def gather_into(self, idx, dst, dst_off, skip_net=False):
        """dst arrays [dst_off : dst_off+len(idx)] = self arrays[idx] (one kernel)"""
        n = idx.numel()
        copied = self._wait_staged(idx)
        # (raw pointer arithmetic instead of tensor slices: this call sits in the host-paced start of a frame)
        vp = ctypes.c_void_p
        src, has_net = self.a, dst.get("net") is not None and not skip_net
        o = lambda t, row_bytes: vp(t.data_ptr() + dst_off * row_bytes)
        L.check(L.lib().dpvo_gather_edges(
            vp(idx.data_ptr()), ctypes.c_int64(n), vp(src["ii"].data_ptr()), vp(src["jj"].data_ptr()), vp(src["kk"].data_ptr()),
            vp(src["net"].data_ptr()) if has_net else vp(0), vp(src["target"].data_ptr()), vp(src["weight"].data_ptr()),
            o(dst["ii"], 8), o(dst["jj"], 8), o(dst["kk"], 8), o(dst["net"], 4 * self.D) if has_net else vp(0),
            o(dst["target"], 8), o(dst["weight"], 8), ctypes.c_int(self.D), L.stream()), "dpvo_gather_edges")
        if copied:
            self._mark_gathered()

    def _wait_staged(self, idx):
        """indices staged by stage_indices: order the consumer after their copy.  Normally the copy finished a frame ago; a
        cross-stream wait costs ~15 us of queue time on this platform, so it is only inserted when needed."""
        copied = self._staged.pop(idx.data_ptr(), None) if (idx.numel() and self._staged) else None
        if copied is not None and not copied.query():
            torch.cuda.current_stream(self.dev).wait_event(copied)
        return copied is not None

    def _mark_gathered(self):
        """the staging buffers may be rewritten once the gather just issued has run"""
        if self._gathered is None:
            self._gathered = torch.cuda.Event()
        self._gathered.record()

    def keep(self, idx, idx_host=None, also=None, defer_net=False):
        """compact to the edges listed in idx (sorted ascending), ping-pong buffers.  idx_host: the same indices as a numpy
        array, which keeps the host mirror alive (applied lazily).  also = (idx2, dst_store): additionally append the edges
        idx2 to another store (the inactive edges of remove_factors) -- both gathers in one launch."""
        defer_net = defer_net and self.with_state
        if self.b is None:
            self.b = self._new(self.cap)
        if self.net_pending is not None:
            self.materialize_net()              # (a second removal before the update operator ran: apply the first one now)
        if also is None:
            self.gather_into(idx, self.b, 0, skip_net=defer_net)
        else:
            idx2, dst = also
            n2 = idx2.numel()
            dst.reserve(n2)
            self._wait_staged(idx); self._wait_staged(idx2)
            vp = ctypes.c_void_p
            src = self.a
            o = lambda t, off, row_bytes: vp(0) if t is None else vp(t.data_ptr() + off * row_bytes)
            L.check(L.lib().dpvo_gather_edges2(
                vp(idx2.data_ptr()), ctypes.c_int64(n2), o(dst.a["ii"], dst.E, 8), o(dst.a["jj"], dst.E, 8), o(dst.a["kk"], dst.E, 8),
                o(dst.a.get("net"), dst.E, 4 * self.D), o(dst.a["target"], dst.E, 8), o(dst.a["weight"], dst.E, 8),
                vp(idx.data_ptr()), ctypes.c_int64(idx.numel()), o(self.b["ii"], 0, 8), o(self.b["jj"], 0, 8), o(self.b["kk"], 0, 8),
                o(None if defer_net else self.b.get("net"), 0, 4 * self.D), o(self.b["target"], 0, 8), o(self.b["weight"], 0, 8),
                vp(src["ii"].data_ptr()), vp(src["jj"].data_ptr()), vp(src["kk"].data_ptr()),
                vp(src["net"].data_ptr()) if "net" in src else vp(0), vp(src["target"].data_ptr()), vp(src["weight"].data_ptr()),
                ctypes.c_int(self.D), L.stream()), "dpvo_gather_edges2")
            self._mark_gathered()
            dst.E += n2
        self.a, self.b = self.b, self.a
        if defer_net:
            self.a["net"], self.b["net"] = self.b["net"], self.a["net"]       # the state stays where it is ...
            self.net_pending = (idx, idx.numel())                            # ... until the update operator gathers it
        if self._h is not None:
            if idx_host is None:
                self.invalidate_host()
            else:
                self._log.append(("keep", idx_host))
        self.E = idx.numel()


def reduce_edges(flow_mag, ii, jj, max_num_edges, nms):
    """Greedy NMS over candidate loop-closure edges (optim_utils.py:23-60); integer result is bit-exact with the
    reference for the same np.argsort tie-breaking."""
    es = []
    if ii.size == 0:
        return np.zeros((0, 2), dtype=np.int64)
    # (plain Python lists and a set: the loop runs over ~800 candidates on the host with the GPU idle behind it, and numpy scalar
    #  indexing cost three times the loop body)
    order, il, jl, big = np.argsort(flow_mag).tolist(), ii.tolist(), jj.tolist(), (flow_mag >= 1000).tolist()
    ignore = set()
    for idx in order:
        if len(es) + 1 > max_num_edges:
            break
        i, j = il[idx], jl[idx]
        if (j - i) < 30:
            continue
        if big[idx]:
            continue
        if (i, j) in ignore:
            continue
        es.append((i, j))
        for di in range(-nms, nms + 1):
            ignore.add((i + di, j))
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
        self._norm_scratch = None   # dpvo_normalize's scratch (scale, pose 0, partial sums)
        self._loop_host = self._loop_ev = None      # edges_loop: pinned result buffer of dpvo_loop_flow + its event
        self.loop_pre_hits = 0                      # evaluations served by the previous frame call's tail (dpvo_frame_update_t.loop_out)
        self.colors_ = torch.zeros(self.N, self.M, 3, dtype=torch.uint8, device=dev)

        self.index_ = torch.zeros(self.N, self.M, dtype=torch.long, device=dev)
        self.index_map_ = torch.zeros(self.N, dtype=torch.long, device=dev)

        # initialize poses to identity matrix
        self.poses_[:, 6] = 1.0

        # store relative poses for removed frames
        self.delta = {}

        ### edge information: preallocated stores, exposed under the reference's attribute names ###
        # Capacity from the configuration, for the same reason as the inactive store's below: a doubling in the middle of a tracked frame
        # reallocates both sets, the one-call path's scratch and four pinned staging buffers (measured with LOOP_CLOSURE, the first time
        # a batch of loop edges took E past 65 536: one frame of 9.9 ms, profiles/r06_h_lc_host_trace.txt).  Bound: every frame of the removal
        # window (+ the newest two) with its 2 PATCH_LIFETIME + 2 neighbours, plus -- LOOP_CLOSURE -- the MAX_LOOP_PAIRS frame pairs
        # edges_loop() can return at once, M edges each.  Growing by doubling remains the fallback (append_factors of a caller's own edges).
        r_, w_ = int(getattr(self.cfg, "PATCH_LIFETIME", 13)), int(getattr(self.cfg, "REMOVAL_WINDOW", 22))
        need = self.M * (w_ + 2) * (2 * r_ + 2) + (MAX_LOOP_PAIRS * self.M if getattr(self.cfg, "LOOP_CLOSURE", False) else 0)
        self.edges = EdgeStore(DIM, dev, with_state=True, mirror=True, cap=max(1 << 16, 1 << (need - 1).bit_length()))
        ### inactive edge information (i.e., no longer updated, but useful for BA) ###
        # sized for the whole buffer up front when that is affordable (every keyframe eventually retires its ~2 * PATCH_LIFETIME * M
        # edges here, remove_factors(store=True): 40 B per edge, 0.4 GB for the default 4096-frame buffer on a 288 GB device): growing
        # it by doubling allocates -- a device-wide sync -- in the middle of a tracked frame (a 0.3 ms hiccup once per doubling)
        # "Affordable" is checked: at most _INAC_PREALLOC_FRAC (2 %) of the device memory that is free right
        # now -- many trackers per device, a small GPU or a test suite that builds dozens of them fall back to the doubling store
        per_frame = 2 * int(getattr(self.cfg, "PATCH_LIFETIME", 13)) * self.M
        cap = max(1 << 17, min(self.N * per_frame, 1 << 24))
        if torch.device(dev).type == "cuda":
            free_b, _ = torch.cuda.mem_get_info(dev)
            budget = _INAC_PREALLOC_FRAC * free_b
            cap = max(1 << 17, min(cap, int(budget // (2 * 40))))      # (two ping-pong sets of 40 B per edge)
        self.edges_inac = EdgeStore(DIM, dev, with_state=False, cap=cap)

    # active edges (views of the store; assignment copies into it, in-place ops on the views work as in the reference)
    ii = property(lambda self: self.edges.view("ii"), lambda self, v: self.edges.assign("ii", v))
    jj = property(lambda self: self.edges.view("jj"), lambda self, v: self.edges.assign("jj", v))
    kk = property(lambda self: self.edges.view("kk"), lambda self, v: self.edges.assign("kk", v))
    net = property(lambda self: self.edges.view("net")[None], lambda self, v: self.edges.assign("net", v))
    target = property(lambda self: self.edges.view("target")[None], lambda self, v: self.edges.assign("target", v))
    weight = property(lambda self: self.edges.view("weight")[None], lambda self, v: self.edges.assign("weight", v))
    ii_inac = property(lambda self: self.edges_inac.view("ii"))
    jj_inac = property(lambda self: self.edges_inac.view("jj"))
    kk_inac = property(lambda self: self.edges_inac.view("kk"))
    target_inac = property(lambda self: self.edges_inac.view("target")[None])
    weight_inac = property(lambda self: self.edges_inac.view("weight")[None])

    def edges_loop(self, n=None, pre=None):
        """Adding edges from old patches to new frames (patchgraph.py:56-82).  n: the frame count to evaluate for (default self.n;
        the tracker asks for n + 1 just before it counts a new frame: none of the candidates involves that frame).  pre: (pinned
        buffer, event) of a candidate test that already ran on the device on the current state (dpvo_frame_update_t.loop_out: the tail of
        the previous frame's call) -- used if it was made for this n, ignored otherwise."""
        n = self.n if n is None else n
        lc_range = self.cfg.MAX_EDGE_AGE
        l = n - self.cfg.REMOVAL_WINDOW  # l is the upper bound for "old" patches
        dev = self.poses_.device
        self.loop_lr_count = 0
        if l <= 0:
            return torch.empty(2, 0, dtype=torch.long, device=dev)

        # candidate edges: every (target frame j, old source frame i) pair, j-major (the reference's flatmeshgrid of jj and the old
        # frames' patch ids kk, ii = ix[kk]); flow magnitude of the M patches' centre pixels, validity rule and masked mean of
        # patchgraph.py:64-72 in ONE launch whose result lands in pinned host memory (dpvo_loop_flow) -- the reference's ~20 small
        # operations and its three boolean-mask selections cost six host round trips with the GPU idle behind them, on EVERY frame
        # while no loop is found (tools/lc_host_trace.py: 752 us; LOOP_CLOSURE without loops ran at 740-800 frames/sec instead of 1 000)
        j0, n_j = n - self.cfg.GLOBAL_OPT_FREQ, self.cfg.GLOBAL_OPT_FREQ - self.cfg.KEYFRAME_INDEX
        i0 = max(l - lc_range, 0)
        n_i = l - i0
        if n_j <= 0 or j0 < 0:
            return torch.empty(2, 0, dtype=torch.long, device=dev)
        fm_h = None
        if pre is not None:
            host_p, ev_p = pre
            ev_p.synchronize()
            hp = host_p.numpy()
            if int(hp[0]) == n and int(hp[1]) == n_j * n_i and host_p.numel() >= 2 + n_j * n_i:
                fm_h = hp[2:2 + n_j * n_i].copy()               # same kernel, same state, same ranges: the bits of the launch below
                self.loop_pre_hits += 1
        if fm_h is None:
            if self._loop_host is None or self._loop_host.numel() < n_j * n_i:
                self._loop_host = torch.empty(max(n_j * min(lc_range, self.N), n_j * n_i), dtype=torch.float32).pin_memory()
                self._loop_ev = torch.cuda.Event()
            L.check(L.lib().dpvo_loop_flow(L.ptr(self.poses_), L.ptr(self.patches_), L.ptr(self.intrinsics_), L.ptr(self.index_),
                                           L.i64(j0), L.i64(n_j), L.i64(i0), L.i64(n_i), L.i32(self.M), L.i32(self.P), L.f32(0.5),
                                           ctypes.c_void_p(self._loop_host.data_ptr()), L.stream()), "dpvo_loop_flow")
            self._loop_ev.record()
            self._loop_ev.synchronize()
            fm_h = self._loop_host[:n_j * n_i].numpy().copy()
        # mask = flow_mag < BACKEND_THRESH; reduce_edges(flow_mag[mask], ii[::M][mask], jj[::M][mask], ...) (patchgraph.py:73-76) on the
        # host: the candidates' frame numbers are a function of n (ix[k] == k // M: index_ rows hold their own frame number, dpvo.py:405)
        j_f = np.arange(j0, j0 + n_j, dtype=np.int64)
        i_f = np.arange(i0, l, dtype=np.int64)
        mask = fm_h < np.float32(self.cfg.BACKEND_THRESH)
        es = reduce_edges(fm_h[mask], np.tile(i_f, j_f.size)[mask], np.repeat(j_f, i_f.size)[mask], max_num_edges=MAX_LOOP_PAIRS, nms=1)

        # how many of the edges this call returns are long-range ones by update()'s test `ii < n - REMOVAL_WINDOW - 1` (dpvo.py:348)
        # at the frame count they were evaluated for: known here on the host, the tracker need not ask the device
        es_np = np.asarray(es).reshape(-1, 2)
        self.loop_lr_count = int(np.count_nonzero(es_np[:, 0] < n - self.cfg.REMOVAL_WINDOW - 1)) * self.M
        edges = torch.as_tensor(es, device=dev).reshape(-1, 2)
        ii = edges[:, 0][:, None].expand(-1, self.M)
        jj = edges[:, 1][:, None].expand(-1, self.M)
        kk = ii.mul(self.M) + torch.arange(self.M, device=dev)
        return kk.flatten(), jj.flatten()

    def normalize(self):
        """normalize depth and poses (patchgraph.py:84-95)"""
        if _NORMALIZE_FUSED and self.patches_.dtype == torch.float32 and self.patches_.is_contiguous() and self.poses_.is_contiguous():
            # the mean, both rescalings and the re-anchoring on pose 0 as two launches (dpvo_normalize), the point cloud as a third
            if self._norm_scratch is None:
                self._norm_scratch = torch.zeros(L.lib().dpvo_normalize_scratch_bytes() // 4, dtype=torch.float32, device=self.poses_.device)
            L.check(L.lib().dpvo_normalize(L.ptr(self.poses_), L.ptr(self.patches_), L.i32(self.n), L.i32(self.M), L.i32(self.patches_.shape[-1]),
                                           L.ptr(self._norm_scratch), L.stream()), "dpvo_normalize")
            s = self._norm_scratch[0]
            for t, (t0, dP) in self.delta.items():
                self.delta[t] = (t0, dP.scale(s))
            pops.point_cloud(self.poses, self.patches[:, :self.m], self.intrinsics, self.ix[:self.m], out=self.points_)
            return
        s = self.patches_[:self.n, :, 2].mean()
        self.patches_[:self.n, :, 2] /= s
        self.poses_[:self.n, :3] *= s
        for t, (t0, dP) in self.delta.items():
            self.delta[t] = (t0, dP.scale(s))
        self.poses_[:self.n] = (SE3(self.poses_[:self.n]) * SE3(self.poses_[0:1]).inv()).data

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
