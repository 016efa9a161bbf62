#!/usr/bin/env python
"""GPU box: where the tail of a pruned walk is.  Runs C5 (or --n/--d) once with MCE_PRUNE_TIMES set and prints the
distribution of the per-workgroup durations by position in the dispatch order.
usage: python tools/prune_times.py [--n 10000000] [--d 6] [--kmax 10] [--heavy "0"] -> gpurun_out/prune_times.json"""
import argparse, json, os, struct, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000); ap.add_argument("--d", type=int, default=6); ap.add_argument("--kmax", type=int, default=10)
ap.add_argument("--heavy", default="0")
a = ap.parse_args()
os.environ["MCE_PRUNE_HEAVY"] = a.heavy
tf = "/tmp/prune_times.bin"
os.environ["MCE_PRUNE_TIMES"] = tf
import torch
from mcevidence_amd import _capi
from mcevidence_amd.synth import gaussian_chain
theta = gaussian_chain(6, a.n, a.d, cov="corr")[:, 2:]
ev, U = np.linalg.eigh(np.cov(theta.T))
X = np.ascontiguousarray((theta @ U) / np.sqrt(ev)); del theta
n, d = X.shape
dev = torch.device("cuda")
Xd = torch.from_numpy(X).to(dev)
w = torch.ones(n, dtype=torch.float64, device=dev); fs = torch.zeros(n, dtype=torch.float64, device=dev)
wsb = _capi.knn_workspace_bytes(n, n, d, a.kmax - 1) + _capi.dotp_workspace_bytes(n, a.kmax)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
out = torch.zeros(a.kmax, dtype=torch.float64, device=dev)
_capi.set_prune_mode(_capi.PRUNE_FORCE)
_capi.knn_dotp_dev(Xd.data_ptr(), n, Xd.data_ptr(), n, d, a.kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), wsb, 0)
torch.cuda.synchronize()
raw = open(tf, "rb").read()
nwg, hv_n, hv_S, waves = struct.unpack("4i", raw[:16])
us = np.frombuffer(raw[16:], dtype=np.float32)
hw = hv_n * waves * hv_S
slot = np.where(np.arange(nwg) < hw, np.arange(nwg) // (waves * hv_S), hv_n + (np.arange(nwg) - hw) // waves)
order = np.argsort(-us)
res = dict(kernel=_capi.last_kernel(), n_wg=int(nwg), total_wg_seconds=float(us.sum() * 1e-6), mean_us=float(us.mean()),
           quantiles_us={str(q): float(np.quantile(us, q)) for q in (0.5, 0.9, 0.99, 0.999, 0.9999, 1.0)},
           top=[dict(wg=int(i), slot=int(slot[i]), ms=round(float(us[i]) * 1e-3, 2)) for i in order[:40]],
           wg_over_10ms=int((us > 1e4).sum()), wg_over_20ms=int((us > 2e4).sum()), wg_over_40ms=int((us > 4e4).sum()),
           slots_of_wg_over_10ms=sorted(set(int(slot[i]) for i in np.flatnonzero(us > 1e4)))[:200],
           # time by dispatch-order decile
           mean_us_by_slot_decile=[float(us[(slot >= lo) & (slot < hi)].mean()) for lo, hi in zip(np.linspace(0, slot.max() + 1, 11)[:-1], np.linspace(0, slot.max() + 1, 11)[1:])])
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(REPO, "gpurun_out", "prune_times.json"), "w"), indent=1)
print(json.dumps(res)[:3000])
