#!/usr/bin/env python
"""GPU box (ONE GPU): predicted 1/2/4/8-GPU times of the auto-evidence hot path, config C3 (and C5 with --c5).

Every rank's share of an N-GPU run (mce_knn_dotp_part_f64_dev: the symmetric partition of DESIGN.md 5, or the pruned
walk's block-cyclic parts) is timed SERIALLY on this one GPU, resident data; the slowest rank is the predicted step time
(the ranks run concurrently on a real node; the all-reduce of kmax doubles adds a few tens of microseconds).  PREDICTED,
not measured: no multi-GPU box is reachable from the build container.  Also checks that the shares add up to the
single-rank sums.   usage: python tools/predict_scaling.py [--c5] [--reps 3] -> gpurun_out/predicted_scaling.json"""
import json, math, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from mcevidence_amd import _capi
from mcevidence_amd.synth import gaussian_chain


def whiten(theta):
    cov = np.cov(theta.T); ev, U = np.linalg.eigh(cov)
    return np.ascontiguousarray((theta @ U) / np.sqrt(ev))


def run(name, X, kmax, worlds, reps):
    n, d = X.shape
    dev = torch.device("cuda")
    Xd = torch.from_numpy(X).to(dev)
    w = torch.ones(n, dtype=torch.float64, device=dev); fs = torch.zeros(n, dtype=torch.float64, device=dev)
    wsb = _capi.knn_workspace_bytes(n, n, d, kmax - 1) + _capi.dotp_workspace_bytes(n, kmax)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    out = torch.zeros(kmax, dtype=torch.float64, device=dev)
    res = dict(config=name, n=n, d=d, kmax=kmax, label="PREDICTED from one GPU: every rank's share timed serially, slowest rank = step time", worlds={})
    whole = None
    for W in worlds:
        per_rank, kern, total = [], [], np.zeros(kmax)
        for r in range(W):
            best, km = 1e30, None
            for _ in range(reps):
                _capi.set_profiling(True); torch.cuda.synchronize(); t0 = time.perf_counter()
                _capi.knn_dotp_part_dev(Xd.data_ptr(), n, d, kmax, r, W, w.data_ptr(), fs.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, 0)
                torch.cuda.synchronize(); t = time.perf_counter() - t0
                k = _capi.last_kernel_ms(); _capi.set_profiling(False)
                if t < best: best, km = t, k
            per_rank.append(round(best * 1e3, 3)); kern.append(round(km, 3)); total += out.cpu().numpy()
        if whole is None: whole = total.copy()
        step = max(per_rank)
        res["worlds"][str(W)] = dict(rank_ms=per_rank, rank_search_kernel_ms=kern, predicted_step_ms=step, predicted_queries_per_s=round(n / (step * 1e-3), 1),
                                     speedup_vs_1=None, max_rel_dev_of_summed_dotp_vs_1gpu=float(np.max(np.abs(total[1:] - whole[1:]) / whole[1:])), kernel=_capi.last_kernel())
    t1 = res["worlds"][str(worlds[0])]["predicted_step_ms"]
    for W in worlds:
        res["worlds"][str(W)]["speedup_vs_1"] = round(t1 / res["worlds"][str(W)]["predicted_step_ms"], 3)
        res["worlds"][str(W)]["efficiency"] = round(t1 / res["worlds"][str(W)]["predicted_step_ms"] / W, 3)
    print(json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
    out = [run("C3", whiten(gaussian_chain(3, 1_000_000, 27, cov="corr")[:, 2:]), 10, (1, 2, 4, 8), reps)]
    if "--c5" in sys.argv:
        out.append(run("C5", whiten(gaussian_chain(6, 10_000_000, 6, cov="corr")[:, 2:]), 10, (1, 2, 4, 8), max(1, reps - 1)))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "predicted_scaling.json"), "w"), indent=1)
