#!/usr/bin/env python
"""GPU box: per-call cost of evidence() on Planck-sized chains (the reference driver's pattern:
thousands of independent ~27k x 6 problems, planck_mcevidence.py:306-348)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging; logging.disable(logging.CRITICAL)
import mcevidence_amd as pkg
from mcevidence_amd.synth import planck_like_chains
chains, _, _ = planck_like_chains(seed=1)
for reps, ch in ((200, chains), (200, chains[:1])):
    mce = pkg.MCEvidence(ch, ndim=6, kmax=2, verbose=0)
    mce.evidence()
    t0 = time.perf_counter()
    for _ in range(reps):
        mce.evidence()
    t = (time.perf_counter() - t0) / reps
    print("N=%d ndim=6 kmax=2: %.3f ms per evidence() call" % (mce.nsample[0], t * 1e3))
