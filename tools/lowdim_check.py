import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from mcevidence_amd import _capi
from oracle import oracle_np as orc
rng = np.random.default_rng(0)
for n, d in ((300000, 1), (1000000, 2), (1000000, 3)):
    X = rng.standard_normal((n, d))
    t = time.perf_counter(); dist, idx = _capi.knn(X, X, 5, self_mode=2); t = time.perf_counter() - t
    rows = np.sort(rng.choice(n, 2000, replace=False))
    od, oi = orc.knn_brute(X[rows], X, 6); od, oi = od[:, 1:], oi[:, 1:]
    print(n, d, "wall %.3fs" % t, "max rel err", np.max(np.abs(dist[rows] - od) / od), "idx agree", np.mean(idx[rows] == oi), _capi.last_kernel()[:40])
