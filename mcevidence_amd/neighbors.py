"""``NearestNeighbors``-shaped front end of the HIP search, so that the call at
``/root/reference/MCEvidence.py:1093-1104`` can be replaced by changing one import:

    from mcevidence_amd.neighbors import NearestNeighbors
    nbrs = NearestNeighbors(n_neighbors=kmax+1, metric='euclidean').fit(samples2)
    DkNN, indices = nbrs.kneighbors(samples)

Only what that call site uses is provided (euclidean metric, ``fit``,
``kneighbors``).  ``algorithm``, ``leaf_size`` and ``n_jobs`` are accepted and
ignored: the search is always the exact brute-force MFMA kernel.
"""
from __future__ import annotations

import numpy as np

from . import _capi


class NearestNeighbors(object):
    def __init__(self, n_neighbors=5, metric="euclidean", leaf_size=20, algorithm="auto", n_jobs=None, device=0, **kw):
        if metric not in ("euclidean", "l2", "minkowski"):
            raise ValueError("only the euclidean metric is supported, got %r" % (metric,))
        if metric == "minkowski" and kw.get("p", 2) != 2:
            raise ValueError("minkowski is supported for p=2 only")
        self.n_neighbors = int(n_neighbors)
        self.device = device
        self._fit_X = None
        self._fit_method = "hip_mfma_brute"

    def fit(self, X, y=None):
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError("Expected 2D array, got %dD array instead" % X.ndim)
        self._fit_X = X
        self.n_samples_fit_ = X.shape[0]
        self.n_features_in_ = X.shape[1]
        return self

    def kneighbors(self, X=None, n_neighbors=None, return_distance=True):
        """Like sklearn: with X=None the training points are queried and each point's own
        row is excluded; with an explicit X nothing is excluded (a point queried against
        itself comes back at distance ~0 in column 0)."""
        if self._fit_X is None:
            raise RuntimeError("This NearestNeighbors instance is not fitted yet. Call 'fit' first.")
        K = self.n_neighbors if n_neighbors is None else int(n_neighbors)
        if X is None:
            dist, idx = _capi.knn(self._fit_X, self._fit_X, K, self_mode=_capi.SELF_EXCLUDE, device=self.device)
        else:
            X = np.ascontiguousarray(X, dtype=np.float64)
            same = X is self._fit_X or (X.shape == self._fit_X.shape and X.ctypes.data == self._fit_X.ctypes.data)
            mode = _capi.SELF_INCLUDE if same else _capi.SELF_NONE
            dist, idx = _capi.knn(X, self._fit_X, K, self_mode=mode, device=self.device)
        return (dist, idx) if return_distance else idx
