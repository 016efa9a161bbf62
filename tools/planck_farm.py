"""Planck-driver workload (reference planck_mcevidence.py:306-348): many independent chains of
6k-100k rows, D = 6-8, kmax = 2.  One library call per chain vs ONE batched call
(mce_evidence_feed_batch_f64).  Prints a JSON line.  Needs a GPU."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mcevidence_amd import _capi  # noqa: E402

nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
probs = []
rows = 0
for i in range(nprob):
    d = int(rng.integers(6, 9))
    n = int(np.exp(rng.uniform(np.log(6000), np.log(100000))))
    A = rng.standard_normal((d, d)) + 2 * np.eye(d)
    S = rng.standard_normal((n, d)) @ A
    probs.append((S, None, d, 0, 2, rng.integers(1, 5, n).astype(float), -rng.random(n)))
    rows += n

_capi.evidence_feed(*probs[0])                      # warm-up (module load, pools)
_capi.evidence_feed_batch(probs[:8])
t0 = time.perf_counter()
singles = [_capi.evidence_feed(*p) for p in probs]
t1 = time.perf_counter()
batch = _capi.evidence_feed_batch(probs)
t2 = time.perf_counter()
batch2 = _capi.evidence_feed_batch(probs)
t3 = time.perf_counter()
same = all(np.array_equal(a[0], b[0]) and a[1] == b[1] for a, b in zip(singles, batch))
print(json.dumps({"problems": nprob, "rows_total": rows, "loop_s": round(t1 - t0, 4), "batch_s": round(t2 - t1, 4),
                  "batch_again_s": round(t3 - t2, 4), "loop_ms_per_problem": round(1e3 * (t1 - t0) / nprob, 3),
                  "batch_ms_per_problem": round(1e3 * (t3 - t2) / nprob, 3), "speedup": round((t1 - t0) / (t3 - t2), 2),
                  "bit_identical": bool(same)}))
