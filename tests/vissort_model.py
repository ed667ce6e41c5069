"""CPU model of GS_SORT_VISIBLE (include/gsplat_c.h gs_renderer_set_sort_mode; csrc/gs_vissort.hip) -- test infrastructure.

The reference sorts ALL splats, stably, through the previous order, every SortPoints (GaussianSplatRenderer.cs:612-639;
SplatUtilities.compute:69-82), so after sorts with matrices M_1 .. M_k its order buffer is sorted lexicographically by
(key under M_k, key under M_k-1, ..., key under M_1, index).  The visible-only mode sorts the visible splats by that chain directly:
the kept matrices most recent first, a matrix that occurs again deeper in the chain dropped there (whatever is still tied when the
chain reaches its second occurrence is tied under it), at most `depth` of them, then the index.
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

    def reset(self):
        self.hist, self.dropped = [], 0

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
        return self._keyer.calc_distances(m16).copy()

    def visible_order(self, visible: np.ndarray) -> np.ndarray:
        """visible: bool[N].  The visible splat indices in the order the mode draws them."""
        idx = np.nonzero(visible)[0].astype(np.uint32)
        if not self.hist:
            return idx
        cols = [idx]                                                    # np.lexsort: LAST key is the primary one
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
