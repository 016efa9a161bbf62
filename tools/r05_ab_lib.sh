#!/bin/bash
# GPU box: same-box A/B of two builds of the library (MCE_LIB) on the headline config: C3 step and sweep-kernel time
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for lib in "${1:-_ab/libold.so}" mcevidence_amd/libmcevidence_hip.so; do
    MCE_LIB=$PWD/$lib python - "$lib" <<'PY'
import sys, json, time
import numpy as np, torch
import bench
from mcevidence_amd import _capi
cfg = bench.prep_config("C3")
X, kmax = cfg["X"], cfg["kmax"]; n, d = X.shape
dev = torch.device("cuda")
Xd = torch.from_numpy(X).to(dev); w = torch.from_numpy(cfg["weight"]).to(dev); fs = torch.from_numpy(cfg["fs"]).to(dev)
wsb = _capi.knn_workspace_bytes(n, n, d, kmax - 1) + _capi.dotp_workspace_bytes(n, kmax)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev); out = torch.zeros(kmax, dtype=torch.float64, device=dev)
ts, ks = [], []
for _ in range(6):
    _capi.set_profiling(True); torch.cuda.synchronize(); t0 = time.perf_counter()
    _capi.knn_dotp_dev(Xd.data_ptr(), n, Xd.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), wsb, 0)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); ks.append(_capi.last_kernel_ms()); _capi.set_profiling(False)
print("%-40s step min %.2f med %.2f | kernel min %.2f med %.2f | %s" % (sys.argv[1], min(ts[1:]), float(np.median(ts[1:])), min(ks[1:]), float(np.median(ks[1:])), out.cpu().numpy()[1:3]))
PY
  done
done
