#!/usr/bin/env python
"""Developer script (GPU box): quick parity + timing of the HIP path against the oracle."""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mcevidence_amd import _capi
from oracle import oracle_np as orc

def check_knn(n, d, K, mode, seed=0, nq=None):
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((n, d))
    if mode == 0:
        X = rng.standard_normal((nq or n, d))
    else:
        X = Y if nq is None else Y[:nq]
    t0 = time.perf_counter()
    dist, idx = _capi.knn(X, Y, K, self_mode=mode)
    t1 = time.perf_counter()
    od, oi = orc.knn_brute(X, Y, K, self_mode=2 if mode == 2 else 0)
    if mode == 1:
        od[:, 0] = 0.0
    err = np.max(np.abs(dist - od) / np.maximum(od, 1e-300) * (od > 0))
    same = np.mean(idx == oi)
    print("knn n=%d d=%d K=%d mode=%d nq=%d: max rel err %.3e, idx agree %.6f, wall %.3fs [%s]" % (n, d, K, mode, X.shape[0], err, same, t1 - t0, _capi.last_kernel()), flush=True)
    return err, same

def check_fused(n, d, kmax, k0, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    Y = X if k0 == 1 else rng.standard_normal((n, d))
    w = rng.integers(1, 5, n).astype(float)
    fs = -0.5 * (X ** 2).sum(1); fs -= fs.max()
    dp, dist = _capi.knn_dotp(X, Y, w, fs, kmax, k0, return_dist=True)
    od, _ = orc.knn_brute(X, Y, kmax + 1)
    ref = orc.dotp_literal(od, w, fs, d, k0, kmax)
    ref2 = orc.dotp_logdomain(od, w, fs, d, k0, kmax)
    rel = np.max(np.abs(dp[k0:] - ref[k0:]) / ref[k0:])
    print("fused n=%d d=%d kmax=%d k0=%d: max rel err dotp %.3e (logdomain-vs-literal %.3e), dist err %.3e" % (
        n, d, kmax, k0, rel, np.max(np.abs(ref2[k0:] - ref[k0:]) / ref[k0:]), np.max(np.abs(dist - od[:, k0:kmax]))), flush=True)
    d2 = _capi.dotp(od, w, fs, d, k0, kmax)
    print("   unfused dotp rel err %.3e" % np.max(np.abs(d2[k0:] - ref[k0:]) / ref[k0:]), flush=True)

def timing(n, d, kmax, reps=2):
    import torch
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.standard_normal((n, d))).cuda()
    w = torch.ones(n, dtype=torch.float64, device="cuda")
    fs = torch.zeros(n, dtype=torch.float64, device="cuda")
    K = kmax - 1
    wsb = _capi.knn_workspace_bytes(n, n, d, K) + _capi.dotp_workspace_bytes(n, kmax)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    out = torch.zeros(kmax, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for r in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _capi.knn_dotp_dev(X.data_ptr(), n, X.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), wsb, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        ks = (d + 1 + 3) // 4
        print("timing n=%d d=%d kmax=%d: %.2f ms -> %.3f Mq/s, %.2f TFLOP/s (2*4KS flops/pair) [%s] ws=%.1f MB" % (
            n, d, kmax, ms, n / ms / 1e3, n * n * 2.0 * 4 * ks / ms / 1e9, _capi.last_kernel(), wsb / 1e6), flush=True)
    return out.cpu().numpy()

if __name__ == "__main__":
    print("devices:", _capi.device_count(), "mode", _capi.get_search_mode())
    check_knn(5000, 6, 5, 0, nq=3000)
    check_knn(5000, 6, 5, 1)
    check_knn(5000, 6, 5, 2)
    check_knn(3000, 27, 11, 2)
    check_knn(20000, 15, 5, 0, nq=777)
    check_knn(1000, 3, 2, 2)
    check_knn(4000, 33, 7, 1)
    check_knn(300, 63, 32, 2)
    check_fused(6000, 6, 5, 1)
    check_fused(6000, 27, 10, 1)
    check_fused(5000, 15, 4, 0)
    if len(sys.argv) > 1:
        timing(100_000, 6, 4)
        timing(200_000, 27, 10)
        timing(1_000_000, 27, 10)
