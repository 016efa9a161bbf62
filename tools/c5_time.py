#!/usr/bin/env python
"""GPU box: the C5 hot path (n = 10^7, d = 6, kmax = 10, whole evidence() reduction) timed on the library MCE_LIB names,
with the reduction's output printed in full precision -- two builds must print the same digits.
usage: MCE_LIB=build_ab/lib_x.so python tools/c5_time.py [--n N] [--d D] [--kmax K] [--reps R]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
n, d, kmax, reps = arg("--n", 10_000_000), arg("--d", 6), arg("--kmax", 10), arg("--reps", 3)
import torch
from mcevidence_amd import _capi
from mcevidence_amd.synth import gaussian_chain
theta = gaussian_chain(6, n, d, cov="corr")[:, 2:]
ev, U = np.linalg.eigh(np.cov(theta.T))
X = np.ascontiguousarray((theta @ U) / np.sqrt(ev)); del theta
dev = torch.device("cuda")
Xd = torch.from_numpy(X).to(dev)
w = torch.ones(n, dtype=torch.float64, device=dev); fs = torch.zeros(n, dtype=torch.float64, device=dev)
wsb = _capi.knn_workspace_bytes(n, n, d, kmax - 1) + _capi.dotp_workspace_bytes(n, kmax)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
out = torch.zeros(kmax, dtype=torch.float64, device=dev)
ms = []
for _ in range(reps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _capi.knn_dotp_dev(Xd.data_ptr(), n, Xd.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), wsb, 0)
    torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
print("call ms (first = warm-up):", " ".join("%.2f" % m for m in ms), "| min %.2f" % min(ms[1:]))
print("pruned walk:", _capi.last_prune_stats(), "| out:", " ".join(repr(float(v)) for v in out.cpu().numpy()[1:4]))
