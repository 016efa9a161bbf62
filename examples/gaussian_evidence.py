#!/usr/bin/env python
"""Known-answer demo (the recipe of the reference's gaussian example,
/root/reference/examples.py:267-342, rewritten): draw N samples from a normalised D-dimensional
Gaussian -- the evidence is exactly 1, so ln E = 0 with prior volume 1 -- and estimate ln E from
the chain with the k-th nearest-neighbour estimator on the GPU.

    python examples/gaussian_evidence.py [N] [D] [kmax]
"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mcevidence_amd import MCEvidence                    # noqa: E402
from mcevidence_amd.synth import gaussian_chain          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 5

chain = gaussian_chain(seed=1, n=n, d=d, cov="corr")     # columns: weight, -lnL, theta_1..theta_d
t0 = time.perf_counter()
lnE = MCEvidence([chain], kmax=kmax, verbose=0).evidence()
dt = time.perf_counter() - t0
print("N=%d D=%d: ln E (k=1..%d) = %s   [true value 0]   %.3f s" % (n, d, kmax - 1, np.array2string(lnE, precision=4), dt))
# cross evidence: neighbours of one half of the chain searched in the other half
np.random.seed(0)
lnX = MCEvidence([chain], kmax=kmax, split=True, verbose=0).evidence()
print("cross-evidence (k=2..%d)    = %s" % (kmax, np.array2string(lnX, precision=4)))
