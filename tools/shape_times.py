#!/usr/bin/env python
"""GPU box: time the fused search + reduction on resident data for a list of shapes under given modes, one line each.
usage: python tools/shape_times.py "n,d,K[,sym=0|1|2][,prune=0|1|2][,mode=0|1][,cross=m]" ...     (auto evidence unless cross=<rows of s2>)"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from mcevidence_amd import _capi
dev = torch.device("cuda")
for spec in sys.argv[1:]:
    parts = spec.split(",")
    n, d, K = int(parts[0]), int(parts[1]), int(parts[2])
    opt = dict(p.split("=") for p in parts[3:])
    _capi.set_sym_mode(int(opt.get("sym", 0))); _capi.set_prune_mode(int(opt.get("prune", 0))); _capi.set_search_mode(int(opt.get("mode", 0)))
    rng = np.random.default_rng(n + d)
    X = torch.from_numpy(rng.standard_normal((n, d))).to(dev)
    m = int(opt.get("cross", 0))
    Y = torch.from_numpy(rng.standard_normal((m, d))).to(dev) if m else X
    k0 = 0 if m else 1
    kmax = K + k0
    nr = m if m else n
    w = torch.ones(n, dtype=torch.float64, device=dev); fs = torch.zeros(n, dtype=torch.float64, device=dev)
    wsb = _capi.knn_workspace_bytes(n, nr, d, K) + _capi.dotp_workspace_bytes(n, kmax)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev); out = torch.zeros(kmax, dtype=torch.float64, device=dev)
    ms = []
    for _ in range(int(opt.get("reps", 3)) + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _capi.knn_dotp_dev(X.data_ptr(), n, Y.data_ptr(), nr, d, kmax, k0, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), wsb, 0)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    print("%-40s min %9.3f ms  (%s)  sum[k0]=%r  | %s" % (spec, min(ms[1:]), " ".join("%.2f" % x for x in ms[1:]), float(out[k0].item()), _capi.last_kernel()), flush=True)
    del X, Y, ws, w, fs
    torch.cuda.empty_cache()
