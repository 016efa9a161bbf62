#!/usr/bin/env python
"""GPU box (ONE GPU): predicted 1/2/4/8-GPU times of the hot path for the BASELINE configs that are multi-GPU workloads --
C3 (auto, 1M x 27), C4 (cross, 1M + 1M x 15: query shards of s1 against the replicated s2) and C5 (auto, 10M x 6).

Every rank's share of a W-GPU run (auto: mce_knn_dotp_part_f64_dev -- the symmetric partition of DESIGN.md 5, query shards
beyond four ranks, every W-th wave of the pruned walk; cross: contiguous rows of s1) is timed SERIALLY on this one GPU,
resident data; the slowest rank is the predicted step time (the ranks run concurrently on a real node; the all-reduce of
kmax doubles adds a few tens of microseconds).  PREDICTED, not measured: no multi-GPU box is reachable from the build
container.  Also checks that the shares add up to the single-rank sums.
usage: python tools/predict_scaling.py [C3 C4 C5] [--reps 3] -> gpurun_out/predicted_scaling.json"""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from mcevidence_amd import _capi
import bench


def run(name, worlds, reps):
    cfg = bench.prep_config(name)
    X, Y, kmax, k0 = cfg["X"], cfg["Y"], cfg["kmax"], cfg["k0"]
    S, d = X.shape
    auto = Y is None
    dev = torch.device("cuda")
    Xd = torch.from_numpy(X).to(dev)
    Yd = Xd if auto else torch.from_numpy(Y).to(dev)
    nr = S if auto else Y.shape[0]
    w = torch.from_numpy(cfg["weight"]).to(dev); fs = torch.from_numpy(cfg["fs"]).to(dev)
    wsb = _capi.knn_workspace_bytes(S, nr, d, kmax - k0) + _capi.dotp_workspace_bytes(S, kmax)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    out = torch.zeros(kmax, dtype=torch.float64, device=dev)
    res = dict(config=name, n=S, nr=nr, d=d, kmax=kmax, k0=k0, label="PREDICTED from one GPU: every rank's share timed serially, slowest rank = step time", worlds={})
    whole = None
    # link rate the exchange of the distributed k-d preparation is PRICED at (bytes per second and direction, as the pairs-once emulation)
    LINK = 50e9
    for W in worlds:
        per_rank, kern, total = [], [], np.zeros(kmax)
        # round 6: a pruned search's k-d preparation is distributed over the ranks (mce_prune_part_prepare_dev + all-reduce of the
        # permutation + mce_knn_dotp_part_prepared_f64_dev).  Emulated: every rank's part of the sorts is run first and the arrays
        # summed (what the all-reduce does); then, per rank and timed: its own part again, the summed permutation put in place, the
        # search -- plus the all_gather of the int32 permutation's ranges priced at (W - 1) / W x bytes / LINK.
        dist_prep = auto and W > 1 and _capi.prune_part_applies(S, d, kmax, W) and os.environ.get("MCE_BENCH_DIST_PREP", "1") != "0"
        perm_sum, perm_off, perm_cnt, prep_ms, xchg_ms = None, 0, 0, [], 0.0
        if dist_prep:
            for r in range(W):
                perm_off, perm_cnt = _capi.prune_part_prepare_dev(Xd.data_ptr(), S, d, kmax, r, W, ws.data_ptr(), wsb, 0)
                torch.cuda.synchronize()
                pr = ws[perm_off:perm_off + 4 * perm_cnt].view(torch.int32)
                perm_sum = pr.clone() if perm_sum is None else perm_sum + pr
            xchg_ms = (W - 1) / W * 4.0 * perm_cnt / LINK * 1e3             # an all_gather of the ranges: every rank receives (W - 1) / W of the array
        for r in range(W):
            lo, hi = (0, S) if auto else ((S * r) // W, (S * (r + 1)) // W)
            best, km = 1e30, None
            for _ in range(reps):
                _capi.set_profiling(True); torch.cuda.synchronize(); t0 = time.perf_counter()
                if dist_prep:
                    _capi.prune_part_prepare_dev(Xd.data_ptr(), S, d, kmax, r, W, ws.data_ptr(), wsb, 0)
                    torch.cuda.synchronize(); tp = time.perf_counter() - t0
                    ws[perm_off:perm_off + 4 * perm_cnt].view(torch.int32).copy_(perm_sum)          # (untimed: the collective is priced instead)
                    torch.cuda.synchronize(); t0b = time.perf_counter()
                    _capi.knn_dotp_part_prepared_dev(Xd.data_ptr(), S, d, kmax, r, W, w.data_ptr(), fs.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, 0)
                    torch.cuda.synchronize(); t = tp + (time.perf_counter() - t0b) + xchg_ms * 1e-3
                    k = _capi.last_kernel_ms(); _capi.set_profiling(False)
                    if t < best: best, km, bp = t, k, tp
                    continue
                if auto:
                    _capi.knn_dotp_part_dev(Xd.data_ptr(), S, d, kmax, r, W, w.data_ptr(), fs.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, 0)
                else:
                    _capi.knn_dotp_dev(Xd.data_ptr() + lo * d * 8, hi - lo, Yd.data_ptr(), nr, d, kmax, 0, 0, w.data_ptr() + lo * 8, fs.data_ptr() + lo * 8,
                                       out.data_ptr(), 0, ws.data_ptr(), wsb, 0)
                torch.cuda.synchronize(); t = time.perf_counter() - t0
                k = _capi.last_kernel_ms(); _capi.set_profiling(False)
                if t < best: best, km = t, k
            per_rank.append(round(best * 1e3, 3)); kern.append(round(km, 3)); total += out.cpu().numpy()
            if dist_prep: prep_ms.append(round(bp * 1e3, 3))
        if whole is None: whole = total.copy()
        step = max(per_rank)
        res["worlds"][str(W)] = dict(rank_ms=per_rank, rank_search_kernel_ms=kern, predicted_step_ms=step, predicted_queries_per_s=round(S / (step * 1e-3), 1),
                                     speedup_vs_1=None, max_rel_dev_of_summed_dotp_vs_1gpu=float(np.max(np.abs(total[k0:] - whole[k0:]) / whole[k0:])), kernel=_capi.last_kernel(),
                                     distributed_kd_preparation=bool(dist_prep), own_sorts_ms=prep_ms or None,
                                     permutation_allreduce_ms_priced=(round(xchg_ms, 3) if dist_prep else None))
        if W == worlds[0]:
            lnE = bench.lnE_from_dotp(total, cfg)
            g = bench.golden_lnE(name, cfg)
            res["lnE"] = [float(x) for x in lnE]
            if g is not None: res["max_abs_dlnE_vs_reference"] = float(np.max(np.abs(lnE - np.array(g["lnE"]))))
    t1 = res["worlds"][str(worlds[0])]["predicted_step_ms"]
    for W in worlds:
        res["worlds"][str(W)]["speedup_vs_1"] = round(t1 / res["worlds"][str(W)]["predicted_step_ms"], 3)
        res["worlds"][str(W)]["efficiency"] = round(t1 / res["worlds"][str(W)]["predicted_step_ms"] / W, 3)
    print(json.dumps(res), flush=True)
    del Xd, Yd, ws
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
    names = [a for a in sys.argv[1:] if a in ("C2", "C3", "C4", "C5")] or ["C3", "C4", "C5"]
    out = [run(nm, (1, 2, 4, 8), reps if nm != "C5" else max(1, reps - 1)) for nm in names]
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "predicted_scaling.json"), "w"), indent=1)
