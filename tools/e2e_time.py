#!/usr/bin/env python
"""GPU box: end-to-end time of MCEvidence(...).evidence() (host feeders + H2D + hot path)."""
import sys, os, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging; logging.disable(logging.CRITICAL)
import mcevidence_amd as pkg
from mcevidence_amd.synth import gaussian_chain
n, d, kmax = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 27, 10
chain = gaussian_chain(3, n, d, cov="corr")
mce = pkg.MCEvidence([chain], kmax=kmax, verbose=0)
mce.evidence()  # warm (library load, first-touch)
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
lnE = mce.evidence()
pr.disable(); t = time.perf_counter() - t0
print("evidence() wall: %.3f s  lnE[0]=%.10f" % (t, lnE[0]))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:2500])
