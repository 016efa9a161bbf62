#!/usr/bin/env python
"""GPU box: per-wave cycle breakdown of the pruned walk at C5 (library built with -DMCE_PRUNE_PROF=1; MCE_LIB selects it).
usage: MCE_LIB=build_ab/lib_x_prof.so MCE_PRUNE_PROF=1 python tools/prune_prof.py [--lists short|long]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["MCE_PRUNE_PROF"] = "1"
if "--lists" in sys.argv: os.environ["MCE_PRUNE_LISTS"] = sys.argv[sys.argv.index("--lists") + 1]
import torch
from mcevidence_amd import _capi
from mcevidence_amd.synth import gaussian_chain
n, d, kmax = 10_000_000, 6, 10
theta = gaussian_chain(6, n, d, cov="corr")[:, 2:]
ev, U = np.linalg.eigh(np.cov(theta.T))
X = np.ascontiguousarray((theta @ U) / np.sqrt(ev)); del theta
dev = torch.device("cuda")
Xd = torch.from_numpy(X).to(dev)
w = torch.ones(n, dtype=torch.float64, device=dev); fs = torch.zeros(n, dtype=torch.float64, device=dev)
wsb = _capi.knn_workspace_bytes(n, n, d, kmax - 1) + _capi.dotp_workspace_bytes(n, kmax)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
out = torch.zeros(kmax, dtype=torch.float64, device=dev)
for _ in range(2):
    _capi.set_profiling(True); torch.cuda.synchronize()
    _capi.knn_dotp_dev(Xd.data_ptr(), n, Xd.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), wsb, 0)
    torch.cuda.synchronize()
    print(_capi.last_kernel_ms(), _capi.last_kernel()[-40:])
    _capi.set_profiling(False)
print(_capi.last_prune_stats())
