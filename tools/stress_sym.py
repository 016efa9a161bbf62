import os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from mcevidence_amd import _capi as capi
capi.set_prune_mode(capi.PRUNE_OFF)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)
t0 = time.time(); bad = 0
for case in range(ncase):
    d = int(rng.choice([1, 2, 3, 5, 6, 8, 13, 15, 16, 27, 31, 33, 47, 50, 63]))
    big = case % 10 == 9
    n = int(rng.integers(100000, 400000)) if big else int(rng.integers(1025, 60000))
    K = int(rng.integers(1, 17))
    sm = [capi.SELF_EXCLUDE, capi.SELF_INCLUDE, capi.SELF_NONE][case % 3]
    kind = case % 6
    Y = rng.standard_normal((n, d)) * rng.uniform(0.1, 30.0) + rng.standard_normal(d) * rng.uniform(0, 50.0)
    if kind == 1: Y[rng.integers(0, n, n // 3)] = Y[rng.integers(0, n, n // 3)]
    if kind == 2: Y = np.round(Y)
    if kind == 3: Y[: n // 2] = Y[: n // 2] * 1e-3 + 40.0          # a tight far cluster
    if kind == 4: Y = np.exp(Y / max(1.0, np.abs(Y).max()) * 3)    # skewed
    os.environ.update(MCE_SYM_SEED_ROWS=str(int(rng.integers(64, 40000))), MCE_SYM_SEED_SHARE="2", MCE_SYM_SEED_MODE=str(int(rng.integers(0, 3))),
                      MCE_SYM_PANEL=str(int(rng.choice([1, 2, 5, 16, 96]))), MCE_SYM_BUCKET=str(int(rng.choice([1, 4, 30, 200]))))
    capi.set_sym_mode(capi.SYM_OFF); d0, i0 = capi.knn(Y, Y, K, self_mode=sm)
    capi.set_sym_mode(capi.SYM_FORCE)
    for rep in range(2 if not big else 1):
        d1, i1 = capi.knn(Y, Y, K, self_mode=sm)
        assert "symmetric" in capi.last_kernel()
        if not (np.array_equal(d0, d1) and np.array_equal(i0, i1)):
            bad += 1
            print("MISMATCH case", case, "rep", rep, d, n, K, sm, kind, capi.last_kernel(), int(np.sum((d0 != d1) | (i0 != i1))), flush=True)
print("seed", seed, "cases", ncase, "mismatches", bad, "in %.0f s" % (time.time() - t0))
