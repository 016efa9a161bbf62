#!/usr/bin/env python
"""GPU box: distribution of the matrix core's accumulation error against the bound the fp16 filter assumes
(knn_f16.hpp: eps_q = 32 KST 2^-24 (|x^| + max |y^|)^2), over thousands of tiles of every kind the kernels can meet.

Tiles go through mce_debug_mfma_tiles_f16 -- the kernels' own v_mfma_f32_32x32x16_f16 sequence -- and are compared with
the fp64 product of the same fp16 operands (fp16 x fp16 is exact in fp64; the fp64 sum of <= 64 such terms is within
2^-46 of their absolute sum: nine orders below eps_q).  Also imported by tests/test_gpu_parity.py.

usage: python tools/mfma_error_hist.py [tiles per kind and kst = 500]  -> gpurun_out/mfma_error_model.json"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

KINDS = ("near", "antipodal", "random", "subnormal", "max_magnitude", "one_signed", "mixed_scale")


def make_tiles(kind, kst, ntiles, rng):
    """(yprime, xprime) fp16 [ntiles, 32, 16 kst] as f16_pack_refs / f16_pack_queries lay them out: y' = [-2 y^, n_hi, n_mid,
    n_lo], x' = [x^, 1, 1, 1], D = 16 kst - 3, rows scaled into the radius the kernels scale to (|.| <= 200)."""
    D = 16 * kst - 3
    x = rng.standard_normal((ntiles, 32, D))
    x *= 200.0 / np.linalg.norm(x, axis=2, keepdims=True) * (0.5 + 0.5 * rng.random((ntiles, 32, 1)))
    perm = np.argsort(rng.random((ntiles, 32)), axis=1)
    xs = np.take_along_axis(x, perm[:, :, None], axis=1)
    if kind == "near":              # references within a hair of the queries: |y^|^2 ~ 2 x^.y^ ~ 40 000, A ~ -|x^|^2
        y = xs + rng.standard_normal((ntiles, 32, D)) * 10.0 ** rng.uniform(-3, 0.5, (ntiles, 1, 1))
    elif kind == "antipodal":       # one-signed products: the partial sums peak
        y = -xs * (1.0 + 0.01 * rng.standard_normal((ntiles, 32, 1)))
    elif kind == "random":
        y = rng.standard_normal((ntiles, 32, D))
        y *= 200.0 / np.linalg.norm(y, axis=2, keepdims=True) * rng.random((ntiles, 32, 1))
    elif kind == "subnormal":       # fp16 subnormals (< 6.1e-5) next to full-size components, on either side
        y = xs.copy()
        m = rng.random((ntiles, 32, D)) < 0.5
        y[m] = rng.uniform(-6e-5, 6e-5, m.sum())
        mx = rng.random((ntiles, 32, D)) < 0.3
        x = x.copy()
        x[mx] = rng.uniform(-6e-5, 6e-5, mx.sum())
    elif kind == "max_magnitude":   # every row ON the radius, energy in few components (|x_i| up to 200: -2 y^ up to 400)
        x = np.zeros((ntiles, 32, D))
        y = np.zeros((ntiles, 32, D))
        for a in (x, y):
            k = rng.integers(1, 4, (ntiles, 32))
            for t in range(ntiles):
                for r in range(32):
                    idx = rng.choice(D, k[t, r], replace=False)
                    v = rng.standard_normal(k[t, r])
                    a[t, r, idx] = 200.0 * v / np.linalg.norm(v)
    elif kind == "one_signed":      # all components of one sign and equal size: every partial sum has the full magnitude
        s = rng.choice([-1.0, 1.0], (ntiles, 1, 1))
        x = np.full((ntiles, 32, D), 200.0 / np.sqrt(D)) * (1.0 - 0.001 * rng.random((ntiles, 32, D)))
        y = s * x[:, ::-1, :] * (1.0 - 0.001 * rng.random((ntiles, 32, D)))
    elif kind == "mixed_scale":     # rows of very different norms in one tile (2^-10 .. 1 of the radius)
        y = rng.standard_normal((ntiles, 32, D))
        y *= 200.0 / np.linalg.norm(y, axis=2, keepdims=True) * 2.0 ** rng.uniform(-10, 0, (ntiles, 32, 1))
        x = x * 2.0 ** rng.uniform(-10, 0, (ntiles, 32, 1))
    else:
        raise ValueError(kind)
    xh, yh = x.astype(np.float16), y.astype(np.float16)
    n2 = (yh.astype(np.float64) ** 2).sum(axis=2)             # |y^|^2 from the CONVERTED values, in three fp16 pieces
    n_hi = n2.astype(np.float16)
    n_mid = (n2 - n_hi.astype(np.float64)).astype(np.float16)
    n_lo = (n2 - n_hi.astype(np.float64) - n_mid.astype(np.float64)).astype(np.float16)
    yp = np.concatenate([(-2.0 * yh.astype(np.float64)).astype(np.float16), n_hi[..., None], n_mid[..., None], n_lo[..., None]], axis=2)
    xp = np.concatenate([xh, np.ones((ntiles, 32, 3), dtype=np.float16)], axis=2)
    assert yp.shape[2] == 16 * kst and np.all(np.isfinite(yp.astype(np.float64)))
    return yp, xp


def error_over_bound(yp, xp, kst):
    """|MFMA - exact| / eps_q for every (tile, row, query)"""
    from mcevidence_amd import _capi
    A = _capi.debug_mfma_tiles(yp, xp).astype(np.float64)                     # [tile, row, query]
    yp64, xp64 = yp.astype(np.float64), xp.astype(np.float64)
    exact = np.einsum("tjk,tik->tji", yp64, xp64)
    D = 16 * kst - 3
    xn = np.sqrt((xp64[:, :, :D] ** 2).sum(axis=2))                           # |x^| per query
    ymax = np.sqrt((0.25 * yp64[:, :, :D] ** 2).sum(axis=2).max(axis=1))      # max |y^| of the tile (y' holds -2 y^)
    eps = 32.0 * kst * 2.0 ** -24 * (xn[:, None, :] + ymax[:, None, None]) ** 2
    return np.abs(A - exact) / eps


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    rng = np.random.default_rng(2024)
    out = dict(what="|v_mfma_f32_32x32x16_f16 chain - exact| / eps_q, eps_q = 32 KST 2^-24 (|x^| + max|y^|)^2 (knn_f16.hpp)", tiles_per_kind_and_kst=per, kst={})
    edges = [0, 1e-4, 1e-3, 1e-2, 0.03, 0.1, 0.2, 0.3, 0.4, 0.5, 0.75, 1.0, np.inf]
    total = 0
    for kst in (1, 2, 3, 4, 5, 6, 8):
        rec = {}
        for kind in KINDS:
            yp, xp = make_tiles(kind, kst, per if kind != "max_magnitude" else min(per, 200), rng)
            r = error_over_bound(yp, xp, kst)
            total += len(yp)
            h, _ = np.histogram(r, bins=edges)
            rec[kind] = dict(tiles=len(yp), max=float(r.max()), p999=float(np.quantile(r, 0.999)), median=float(np.median(r)),
                             histogram=dict(edges=[float(e) for e in edges[:-1]] + ["inf"], counts=[int(c) for c in h]))
        out["kst"][str(kst)] = rec
    out["tiles_total"] = total
    out["max_over_everything"] = max(v["max"] for k in out["kst"].values() for v in k.values())
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "mfma_error_model.json"), "w"), indent=1)
    print(json.dumps({k: {kk: round(vv["max"], 4) for kk, vv in v.items()} for k, v in out["kst"].items()}), "tiles", total, "max", out["max_over_everything"])


if __name__ == "__main__":
    main()
