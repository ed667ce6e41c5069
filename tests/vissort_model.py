"""CPU model of GS_SORT_VISIBLE (include/gsplat_c.h gs_renderer_set_sort_mode; csrc/gs_vissort.hip) -- test infrastructure.

The reference sorts ALL splats, stably, through the previous order, every SortPoints (GaussianSplatRenderer.cs:612-639;
SplatUtilities.compute:69-82), so after sorts with matrices M_1 .. M_k its order buffer is sorted lexicographically by
(key under M_k, key under M_k-1, ..., key under M_1, index).  The visible-only mode sorts the visible splats by that chain directly:
the kept matrices most recent first, a matrix that occurs again deeper in the chain dropped there (whatever is still tied when the
chain reaches its second occurrence is tied under it), then the base order (the index while the base is CSSetIndices' identity).
`depth` truncates the chain (what ABI 7 did at 32 rows -- kept to show what that loses); `consolidate()` is what the library does instead
when its history is full: the recorded sorts are carried out on all N and the result becomes the new base.
"""
from __future__ import annotations

import numpy as np

import oracle_lib as O


class VisibleSortModel:
    def __init__(self, asset, depth: int = 32):
        self.asset, self.depth = asset, depth
        self.hist = []                      # distinct rows (4 floats), most recent first
        self.dropped = 0
        self._keyer = O.Oracle(asset)       # its order stays the identity: calc_distances gives the keys BY SPLAT INDEX
        self.base = np.arange(asset.splatCount, dtype=np.uint32)      # the order the recorded sorts start from
        self.rank = self.base.copy()        # its inverse
        self._keys = {}

    def reset(self):
        self.hist, self.dropped = [], 0
        self.base = np.arange(self.asset.splatCount, dtype=np.uint32)
        self.rank = self.base.copy()

    def consolidate(self):
        """base := the reference's buffer now, computed the way the library does: ONE stable sort of the base by the most recent matrix,
        then every run of equal keys re-ordered by the older matrices' keys (most recent first) and, last, the position."""
        if not self.hist:
            return
        k0 = self.keys_by_index(self.hist[0][1])
        order = self.base[np.argsort(k0[self.base], kind="stable")]
        keys = k0[order]
        cols = [np.arange(len(order))]                                  # position = the base order (least significant)
        for _, m in reversed(self.hist[1:]):
            cols.append(self.keys_by_index(m)[order])
        cols.append(keys)                                               # primary: runs stay where they are
        self.base = order[np.lexsort(cols)]
        self.rank = np.empty_like(self.base)
        self.rank[self.base] = np.arange(len(self.base), dtype=np.uint32)
        self.hist = self.hist[:1]

    def push(self, matrix_sort: np.ndarray):
        m = np.ascontiguousarray(matrix_sort, np.float32).reshape(16)
        row = m[8:12].tobytes()
        if self.hist and self.hist[0][0] == row:
            return
        self.hist = [h for h in self.hist if h[0] != row]
        self.hist.insert(0, (row, m.copy()))
        if len(self.hist) > self.depth:
            self.hist.pop()
            self.dropped += 1

    def keys_by_index(self, m16: np.ndarray) -> np.ndarray:
        k = np.ascontiguousarray(m16, np.float32).tobytes()
        if k not in self._keys:                                         # (a long history asks for the same rows every frame)
            if len(self._keys) > 512:
                self._keys.clear()
            self._keys[k] = self._keyer.calc_distances(m16).copy()
        return self._keys[k]

    def visible_order(self, visible: np.ndarray) -> np.ndarray:
        """visible: bool[N].  The visible splat indices in the order the mode draws them."""
        idx = np.nonzero(visible)[0].astype(np.uint32)
        if not self.hist:
            return idx[np.argsort(self.rank[idx], kind="stable")]
        cols = [self.rank[idx]]                                         # np.lexsort: LAST key is the primary one
        for _, m in reversed(self.hist):
            cols.append(self.keys_by_index(m)[idx])
        return idx[np.lexsort(cols)]

    def longest_run(self, visible: np.ndarray) -> int:
        """Longest run of equal keys (current matrix) among the visible splats."""
        if not self.hist:
            return int(visible.sum())
        k = self.keys_by_index(self.hist[0][1])[visible]
        if k.size == 0:
            return 0
        return int(np.unique(k, return_counts=True)[1].max())


def visible_bits(orc: O.Oracle, params) -> np.ndarray:
    """bool[N]: what calc_view's per-frame launch marks visible (oracle side; orc.calc_view(params) must have run)."""
    _, _, vis = orc.raster_records(params)
    return np.unpackbits(vis.view(np.uint8), bitorder="little")[:orc.n].astype(bool)
