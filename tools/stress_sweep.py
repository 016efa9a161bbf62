"""GPU box: random exhaustive-sweep searches (auto and cross sets, every self mode, duplicates, lattices, K up to 32) with the
default plan -- reference splits, seed phase -- against ONE unseeded sweep over the whole set: distances and rows bit for bit.
usage: python tools/stress_sweep.py [seed] [cases]"""
import os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from mcevidence_amd import _capi as capi
capi.set_prune_mode(capi.PRUNE_OFF); capi.set_sym_mode(capi.SYM_OFF)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
t0 = time.time(); bad = 0; seeded = 0; splits = 0
for case in range(ncase):
    d = int(rng.choice([1, 2, 3, 5, 6, 8, 13, 15, 16, 17, 27, 31, 33, 47, 50, 63]))
    big = case % 12 == 11
    nr = int(rng.integers(150000, 500000)) if big else int(np.exp(rng.uniform(np.log(600), np.log(140000))))
    K = int(rng.integers(1, 33)) if case % 5 == 0 else int(rng.integers(1, 11))
    K = min(K, nr - 2)
    kind = case % 6
    Y = rng.standard_normal((nr, d)) * rng.uniform(0.1, 30.0) + rng.standard_normal(d) * rng.uniform(0, 50.0)
    if kind == 1: Y[rng.integers(0, nr, nr // 3)] = Y[rng.integers(0, nr, nr // 3)]
    if kind == 2: Y = np.round(Y)
    if kind == 3: Y[: nr // 2] = Y[: nr // 2] * 1e-3 + 40.0
    cross = case % 4 == 3
    if cross:
        nq = int(np.exp(rng.uniform(np.log(100), np.log(140000))))
        X = rng.standard_normal((nq, d)) * rng.uniform(0.1, 30.0)
        X[: min(nq, 40)] = Y[: min(nq, 40)]
        sm = capi.SELF_NONE
    else:
        X = Y
        sm = [capi.SELF_EXCLUDE, capi.SELF_INCLUDE, capi.SELF_NONE][case % 3]
    for k_ in ("MCE_F16_SEED_ROWS", "MCE_RSPLIT"): os.environ.pop(k_, None)
    d1, i1 = capi.knn(X, Y, K, self_mode=sm)
    k1 = capi.last_kernel()
    seeded += " seed=" in k1
    splits += "rsplit=1 " not in k1 + " "
    os.environ["MCE_F16_SEED_ROWS"] = "0"; os.environ["MCE_RSPLIT"] = "1"
    d0, i0 = capi.knn(X, Y, K, self_mode=sm)
    if not (np.array_equal(d0, d1) and np.array_equal(i0, i1)):
        bad += 1
        print("MISMATCH case", case, X.shape, Y.shape, K, sm, k1, flush=True)
print("seed", seed, "cases", ncase, "mismatches", bad, "seeded", seeded, "split", splits, "in %.0f s" % (time.time() - t0))
