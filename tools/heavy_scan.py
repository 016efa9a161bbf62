#!/usr/bin/env python
"""GPU box: C5 (10M x 6, kmax 10; HEAVY_SCAN_N rows, HEAVY_SCAN_TAILS=<nu> for Student-t tails) through the pruned walk, every rank's share of a W-GPU run timed serially (as
tools/predict_scaling.py) for several settings of the heavy-block splitting (MCE_PRUNE_HEAVY="<blocks>,<S>").
usage: python tools/heavy_scan.py [W ...]  -> gpurun_out/heavy_scan.json"""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from mcevidence_amd import _capi
from mcevidence_amd.synth import gaussian_chain

worlds = [int(x) for x in sys.argv[1:] if "," not in x and x != "default"] or [1, 8]
sys_settings = [x for x in sys.argv[1:] if "," in x or x == "default"] or ["0", "default"]
N_ROWS = int(os.environ.get("HEAVY_SCAN_N", "10000000"))
theta = gaussian_chain(6, N_ROWS, 6, cov="corr")[:, 2:]
if os.environ.get("HEAVY_SCAN_TAILS"):      # heavy tails: Student-t with that many degrees of freedom (outlier queries reach far)
    nu = float(os.environ["HEAVY_SCAN_TAILS"])
    theta = theta / np.sqrt(np.random.default_rng(7).chisquare(nu, size=(len(theta), 1)) / nu)
cov = np.cov(theta.T); ev, U = np.linalg.eigh(cov)
X = np.ascontiguousarray((theta @ U) / np.sqrt(ev)); del theta
n, d = X.shape; kmax = 10
dev = torch.device("cuda")
Xd = torch.from_numpy(X).to(dev)
w = torch.ones(n, dtype=torch.float64, device=dev); fs = torch.zeros(n, dtype=torch.float64, device=dev)
wsb = _capi.knn_workspace_bytes(n, n, d, kmax - 1) + _capi.dotp_workspace_bytes(n, kmax)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
out = torch.zeros(kmax, dtype=torch.float64, device=dev)
res = []
ref = None
for setting in (sys_settings):
    if setting == "default": os.environ.pop("MCE_PRUNE_HEAVY", None)
    else: os.environ["MCE_PRUNE_HEAVY"] = setting
    for W in worlds:
        per, kern, tot = [], [], np.zeros(kmax)
        for r in range(W):
            best, km = 1e30, None
            for _ in range(2):
                _capi.set_profiling(True); torch.cuda.synchronize(); t0 = time.perf_counter()
                _capi.knn_dotp_part_dev(Xd.data_ptr(), n, d, kmax, r, W, w.data_ptr(), fs.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, 0)
                torch.cuda.synchronize(); t = time.perf_counter() - t0
                k = _capi.last_kernel_ms(); _capi.set_profiling(False)
                if t < best: best, km = t, k
            per.append(round(best * 1e3, 2)); kern.append(round(km, 2)); tot += out.cpu().numpy()
        if ref is None: ref = tot.copy()
        rec = dict(heavy=setting, W=W, step_ms=max(per), rank_ms=per, rank_walk_kernel_ms=kern, rel_dev=float(np.max(np.abs(tot[1:] - ref[1:]) / ref[1:])), kernel=_capi.last_kernel())
        print(json.dumps(rec), flush=True)
        res.append(rec)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(REPO, "gpurun_out", "heavy_scan.json"), "w"), indent=1)
