import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcevidence_amd import _capi as capi
def run(n, d, K, sm):
    rng = np.random.default_rng(n + d)
    Y = rng.standard_normal((n, d))
    capi.set_prune_mode(capi.PRUNE_OFF)
    capi.set_sym_mode(capi.SYM_OFF); d0, i0 = capi.knn(Y, Y, K, self_mode=sm)
    capi.set_sym_mode(capi.SYM_FORCE); d1, i1 = capi.knn(Y, Y, K, self_mode=sm)
    c = Y.mean(axis=0)
    key = ((Y - c) ** 2).sum(axis=1).astype(np.float32)
    order = np.argsort(key, kind="stable")
    pos = np.empty(n, dtype=np.int64); pos[order] = np.arange(n)
    bad_rows = np.unique(np.argwhere((d0 != d1) | (i0 != i1))[:, 0])
    lost_lower = lost_same = lost_higher = 0
    ex = []
    for r in bad_rows:
        missing = set(i0[r]) - set(i1[r])
        for m in missing:
            br, bm = pos[r] // 512, pos[m] // 512
            if bm < br: lost_lower += 1
            elif bm == br: lost_same += 1
            else: lost_higher += 1
            if len(ex) < 6: ex.append((int(r), int(pos[r]), int(m), int(pos[m]), int(pos[m] // 32), int(pos[r]) % 64))
    print("n=%d d=%d K=%d: %d bad rows; lost neighbours in a lower block %d, same block %d, higher block %d (row side)" % (n, d, K, len(bad_rows), lost_lower, lost_same, lost_higher))
    print("   (row, sorted pos, missing nb, its sorted pos, its tile, row's lane):", ex)
for n, d, K in [(7777, 27, 9), (20000, 6, 10), (2048, 6, 4), (1024, 2, 1), (513, 4, 3)]:
    run(n, d, K, capi.SELF_EXCLUDE)
