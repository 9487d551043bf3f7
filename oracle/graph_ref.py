"""Integer restatement of the DPVO front-end's edge bookkeeping -- TEST INFRASTRUCTURE ONLY.

Follows dpvo/dpvo.py of the reference line by line for everything that touches indices: frame acceptance
(:399-408, :441-446), append_factors (:215-221) with edges_forw / edges_back (:362-375), keyframe removal and
renumbering (:266-303), window removal (:305-310).  The two data-dependent decisions (motion probe >= 2.0,
mean flow < KEYFRAME_THRESH) are INPUTS, so the integer state can be compared bit for bit with the GPU
implementation driven through the same decisions.  numpy only.
"""
import numpy as np


class GraphRef:
    def __init__(self, M=96, PATCH_LIFETIME=13, REMOVAL_WINDOW=22, KEYFRAME_INDEX=4, BUFFER_SIZE=4096):
        self.M, self.r, self.R, self.KI, self.N = M, PATCH_LIFETIME, REMOVAL_WINDOW, KEYFRAME_INDEX, BUFFER_SIZE
        self.n = 0
        self.m = 0
        self.counter = 0
        self.is_initialized = False
        self.ii = np.zeros(0, np.int64); self.jj = np.zeros(0, np.int64); self.kk = np.zeros(0, np.int64)
        self.ii_inac = np.zeros(0, np.int64); self.jj_inac = np.zeros(0, np.int64); self.kk_inac = np.zeros(0, np.int64)
        self.index_ = np.zeros((BUFFER_SIZE, M), np.int64)
        self.tstamps_ = np.zeros(BUFFER_SIZE, np.int64)
        self.delta = {}          # t -> t0 (relative pose itself is float state, not tracked here)
        self.n_updates = 0

    @property
    def ix(self):
        return self.index_.reshape(-1)

    def _append(self, kk, jj):                                   # append_factors(ii=patch ids, jj)  (:215-221)
        self.jj = np.concatenate([self.jj, jj]); self.kk = np.concatenate([self.kk, kk])
        self.ii = np.concatenate([self.ii, self.ix[kk]])

    def _remove(self, m, store):                                 # remove_factors (:223-238)
        if store:
            self.ii_inac = np.concatenate([self.ii_inac, self.ii[m]])
            self.jj_inac = np.concatenate([self.jj_inac, self.jj[m]])
            self.kk_inac = np.concatenate([self.kk_inac, self.kk[m]])
        self.ii, self.jj, self.kk = self.ii[~m], self.jj[~m], self.kk[~m]

    def frame(self, accept=True, drop_keyframe=False):
        """one __call__ (:377-473).  Returns the list of (event, payload) the float pipeline would run."""
        M, n = self.M, self.n
        assert n + 1 < self.N
        self.tstamps_[n] = self.counter                          # :400
        self.index_[n + 1] = n + 1                               # :407
        self.counter += 1                                        # :440
        if n > 0 and not self.is_initialized and not accept:     # :441-444
            self.delta[self.counter - 1] = self.counter - 2
            return "skipped"
        self.n += 1; self.m += M                                 # :446-447
        n = self.n
        # edges_forw (:362-368)  flatmeshgrid(kk range, [n-1]) -> kk-major
        t0, t1 = M * max(n - self.r, 0), M * max(n - 1, 0)
        k1 = np.arange(t0, t1, dtype=np.int64)
        self._append(k1, np.full_like(k1, n - 1))
        # edges_back (:370-375)
        k2 = np.repeat(np.arange(M * max(n - 1, 0), M * n, dtype=np.int64), n - max(n - self.r, 0))
        j2 = np.tile(np.arange(max(n - self.r, 0), n, dtype=np.int64), M)
        self._append(k2, j2)
        if n == 8 and not self.is_initialized:                   # :461-465
            self.is_initialized = True
            self.n_updates += 12
            return "initialized"
        if self.is_initialized:                                  # :467-469
            self.n_updates += 1
            self._keyframe(drop_keyframe)
            return "tracked"
        return "buffered"

    def _keyframe(self, drop):                                   # :266-310
        M = self.M
        if drop:
            k = self.n - self.KI
            self.delta[int(self.tstamps_[k])] = int(self.tstamps_[k - 1])
            self._remove((self.ii == k) | (self.jj == k), store=False)
            self.kk[self.ii > k] -= M
            self.ii[self.ii > k] -= 1
            self.jj[self.jj > k] -= 1
            for i in range(k, self.n - 1):
                self.tstamps_[i] = self.tstamps_[i + 1]
            self.n -= 1; self.m -= M
        self._remove(self.ix[self.kk] < self.n - self.R, store=True)
