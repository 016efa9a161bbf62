"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the
golden vectors recorded from the reference.  Need a real MI355X: run with -m gpu."""
import logging
import math

import numpy as np
import pytest

from helpers import DIST_RTOL, LNE_TOL, OracleBackend, build_mce, chain_of, load_golden, orc

pytestmark = pytest.mark.gpu
logging.disable(logging.CRITICAL)
G = load_golden()


@pytest.fixture(scope="module", params=["auto_f16filter", "f64sweep"])
def capi(request):
    """every test that takes `capi` runs twice: default search mode (fp16 filter + fp64 refine
    where the shape allows it) and the fp64 MFMA sweep."""
    from mcevidence_amd import _capi
    assert _capi.device_count() >= 1, "no GPU visible: the HIP path cannot be tested"
    _capi.set_search_mode(_capi.MODE_F64 if request.param == "f64sweep" else _capi.MODE_AUTO)
    yield _capi
    _capi.set_search_mode(_capi.MODE_AUTO)


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300) * (b != 0)) if a.size else 0.0


# --------------------------------------------------------------------------- kNN
@pytest.mark.parametrize("d", [1, 2, 3, 6, 7, 8, 13, 14, 15, 16, 27, 30, 31, 33, 47, 48, 62, 63, 64, 79, 80, 95, 96, 100, 111, 112, 127])
def test_knn_matches_oracle_over_dims(capi, d):
    rng = np.random.default_rng(100 + d)
    n = 3001
    Y = rng.standard_normal((n, d))
    K = 7
    dist, idx = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
    od, oi = orc.knn_brute(Y, Y, K, self_mode=2)
    assert _rel(dist, od) < DIST_RTOL
    # filter mode: exact keys, rigorous bound; fp64 sweep (round 6): the lists carry K + 2 candidates chosen on GEMM-form keys and the
    # K are picked among them on EXACT distances -- the oracle's rows, all of them, in both modes
    assert np.array_equal(idx, oi)
    assert np.all(np.diff(dist, axis=1) >= 0)


@pytest.mark.parametrize("K", [1, 2, 4, 5, 8, 9, 11, 12, 13, 16, 17, 24, 25, 32])
def test_knn_matches_oracle_over_k(capi, K):
    rng = np.random.default_rng(200 + K)
    X = rng.standard_normal((777, 6))
    Y = rng.standard_normal((2500, 6))
    dist, idx = capi.knn(X, Y, K)
    od, oi = orc.knn_brute(X, Y, K)
    assert _rel(dist, od) < DIST_RTOL and np.array_equal(idx, oi)


@pytest.mark.parametrize("nq,nr", [(1, 50), (17, 33), (255, 256), (257, 129), (1000, 13), (513, 100000)])
def test_knn_ragged_sizes(capi, nq, nr):
    rng = np.random.default_rng(nq * 7 + nr)
    X = rng.standard_normal((nq, 5))
    Y = rng.standard_normal((nr, 5))
    K = min(12, nr)
    dist, idx = capi.knn(X, Y, K)
    od, oi = orc.knn_brute(X, Y, K)
    assert _rel(dist, od) < DIST_RTOL and np.array_equal(idx, oi)


def test_knn_self_modes(capi):
    rng = np.random.default_rng(3)
    X = rng.standard_normal((4099, 9))
    K = 6
    d0, i0 = capi.knn(X, X, K, self_mode=capi.SELF_NONE)
    d1, i1 = capi.knn(X, X, K, self_mode=capi.SELF_INCLUDE)
    d2, i2 = capi.knn(X, X, K - 1, self_mode=capi.SELF_EXCLUDE)
    assert np.all(d1[:, 0] == 0.0) and np.array_equal(i1[:, 0], np.arange(len(X)))     # exactly 0, like the kd-tree path
    assert np.array_equal(d1[:, 1:], d2) and np.array_equal(i1[:, 1:], i2)
    assert np.all(d0[:, 0] == 0.0) and np.array_equal(d0[:, 1:], d1[:, 1:])
    # query shard with an offset == rows of the full answer (multi-GPU sharding rule)
    ds, is_ = capi.knn(X[1000:1777], X, K - 1, self_mode=capi.SELF_EXCLUDE, self_offset=1000)
    assert np.array_equal(ds, d2[1000:1777]) and np.array_equal(is_, i2[1000:1777])


def test_knn_duplicates_and_ties(capi):
    rng = np.random.default_rng(4)
    X = rng.standard_normal((2000, 4))
    X[100:105] = X[99]                     # 6 identical rows
    X[1500] = X[3]
    d, i = capi.knn(X, X, 4, self_mode=capi.SELF_EXCLUDE)
    od, oi = orc.knn_brute(X, X, 4, self_mode=2)
    assert np.all(d[99:105, :4] == 0.0)   # duplicates: exactly 0 (final distances are exact differences)
    mask = od > 0
    assert _rel(d[mask], od[mask]) < DIST_RTOL
    # integer grid -> many exact ties: distances must agree exactly in value
    Z = rng.integers(0, 4, size=(1500, 3)).astype(float)
    d, _ = capi.knn(Z, Z, 8, self_mode=capi.SELF_EXCLUDE)
    od, _ = orc.knn_brute(Z, Z, 8, self_mode=2)
    assert np.allclose(d, od, rtol=0, atol=1e-7) and np.allclose(d ** 2, od ** 2, atol=1e-12)


SEED_FORCED = [dict(MCE_F16_SEED_SHARE="2", MCE_F16_SEED_ROWS="4096", MCE_F16_SEED_TG="1"),
               dict(MCE_F16_SEED_SHARE="3", MCE_F16_SEED_ROWS="20000", MCE_F16_SEED_TG="4"),
               dict(MCE_F16_SEED_SHARE="2", MCE_F16_SEED_ROWS="100000", MCE_F16_SEED_TG="64")]


@pytest.mark.parametrize("shape", [(40000, 40000, 6, 4, True), (60000, 60000, 27, 11, False), (30000, 30000, 16, 16, True),
                                   (50000, 50000, 20, 20, False), (45000, 45000, 40, 7, False), (50000, 50000, 1, 3, True)])
def test_seeded_sweep_is_bit_identical(shape, monkeypatch):
    """The filter kernel's seed phase (an upper bound on every K-th distance before the sweep) only
    removes work: lists with it forced on at small sizes == lists without it == the oracle."""
    from mcevidence_amd import _capi
    nq, nr, d, K, same = shape
    _capi.set_search_mode(_capi.MODE_AUTO)
    _capi.set_prune_mode(1)
    sym_before = _capi.get_sym_mode()
    _capi.set_sym_mode(_capi.SYM_OFF)                    # (this is about the exhaustive sweep's own seed phase)
    try:
        rng = np.random.default_rng(nq + nr + d)
        Y = rng.standard_normal((nr, d))
        Y[1000:1040] = Y[7]                              # a block of duplicates inside the seed rows
        Y[nr // 2] = Y[11]
        X = Y if same else rng.standard_normal((nq, d))
        if not same:
            X[:50] = Y[:50]                              # queries that coincide with reference rows
        sm = _capi.SELF_EXCLUDE if same else 0
        monkeypatch.setenv("MCE_F16_SEED_ROWS", "0")
        d0, i0 = _capi.knn(X, Y, K, self_mode=sm)
        assert _capi.last_kernel().startswith("knn_f16") and " seed=" not in _capi.last_kernel()
        engaged = 0
        for env in SEED_FORCED:
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            d1, i1 = _capi.knn(X, Y, K, self_mode=sm)
            assert np.array_equal(d0, d1) and np.array_equal(i0, i1), env
            engaged += " seed=" in _capi.last_kernel()
        assert engaged >= 1, _capi.last_kernel()
        rows = rng.choice(nq, size=400, replace=False)
        od, oi = orc.knn_brute(X[rows], Y, K + (1 if same else 0))
        if same:                                         # drop the own row (by index) from the oracle's lists
            keep = np.array([[j for j in oi[r] if j != rows[r]][:K] for r in range(len(rows))])
            od = np.sqrt(((X[rows][:, None, :] - Y[keep]) ** 2).sum(-1))
        assert _rel(d0[rows], od) < DIST_RTOL
    finally:
        _capi.set_sym_mode(sym_before)
        _capi.set_prune_mode(0)


@pytest.mark.parametrize("n,d,K", [(2500, 6, 1), (4096, 6, 3), (6144, 15, 4), (8192, 6, 9), (12288, 27, 9), (16384, 6, 3), (20000, 45, 9),
                                   (26862, 6, 1), (33000, 27, 4), (49152, 8, 1), (70000, 7, 2), (100000, 6, 3)])
def test_small_searches_are_split_and_seeded_by_default(n, d, K, monkeypatch):
    """DESIGN.md 3.0, small searches: up to one round of workgroups the plan takes the largest number of reference splits that
    leaves every split a seed phase (splits balanced to within one chunk, seed sized for the smallest) -- same neighbours,
    distances and rows as one unseeded sweep over the whole set, and as the oracle"""
    from mcevidence_amd import _capi
    _capi.set_search_mode(_capi.MODE_AUTO)
    _capi.set_prune_mode(1)
    sym_before = _capi.get_sym_mode()
    _capi.set_sym_mode(_capi.SYM_OFF)
    try:
        rng = np.random.default_rng(n + d + K)
        Y = rng.standard_normal((n, d)) @ (np.eye(d) + 0.2 * rng.standard_normal((d, d)))
        Y[n // 3:n // 3 + 20] = Y[5]                       # duplicates
        d1, i1 = _capi.knn(Y, Y, K, self_mode=_capi.SELF_EXCLUDE)
        k1 = _capi.last_kernel()
        assert " seed=" in k1, k1
        r1 = int(k1.split("rsplit=")[1].split()[0])
        assert r1 * ((n + 511) // 512) <= 256 or r1 == 1, k1
        monkeypatch.setenv("MCE_F16_SEED_ROWS", "0")
        monkeypatch.setenv("MCE_RSPLIT", "1")
        d0, i0 = _capi.knn(Y, Y, K, self_mode=_capi.SELF_EXCLUDE)
        k0 = _capi.last_kernel()
        assert " seed=" not in k0 and "rsplit=1" in k0, k0
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1), (k0, k1)
        rows = rng.choice(n, size=300, replace=False)
        od, oi = orc.knn_brute(Y[rows], Y, K + 1)
        keep = np.array([[j for j in oi[r] if j != rows[r]][:K] for r in range(len(rows))])
        od = np.sqrt(((Y[rows][:, None, :] - Y[keep]) ** 2).sum(-1))
        assert _rel(d1[rows], od) < DIST_RTOL
    finally:
        _capi.set_sym_mode(sym_before)
        _capi.set_prune_mode(0)


@pytest.mark.parametrize("nq,nr,d,K,same", [(140000, 140000, 6, 4, True), (135000, 90000, 27, 9, False), (150000, 150000, 10, 20, True)])
def test_trailing_round_split_is_bit_identical(nq, nr, d, K, same, monkeypatch):
    """A search whose last round of workgroups would fill less than half the chip runs as two query ranges
    (DESIGN.md 3.0): neighbours, distances and the fused sums must equal the single search's to the bit."""
    from mcevidence_amd import _capi
    _capi.set_search_mode(_capi.MODE_AUTO)
    _capi.set_prune_mode(1)
    sym_before = _capi.get_sym_mode()
    _capi.set_sym_mode(_capi.SYM_OFF)                    # (the exhaustive sweep's trailing round; the symmetric sweep has none)
    try:
        rng = np.random.default_rng(nq + d)
        Y = rng.standard_normal((nr, d))
        X = Y if same else rng.standard_normal((nq, d))
        w = rng.integers(1, 5, size=nq).astype(float)
        fs = -rng.random(nq)
        sm = _capi.SELF_EXCLUDE if same else 0
        k0 = 1 if same else 0
        out = {}
        for flag in ("0", "1"):
            monkeypatch.setenv("MCE_TAIL_SPLIT", flag)
            dist, idx = _capi.knn(X, Y, K, self_mode=sm)
            kern = _capi.last_kernel()
            dotp, dd = _capi.knn_dotp(X, Y, w, fs, K + k0, k0, return_dist=True)
            assert np.array_equal(dd, dist)                      # the fused call's distance matrix = the search's
            assert np.array_equal(dotp, _capi.knn_dotp(X, Y, w, fs, K + k0, k0))
            out[flag] = (dist, idx, dotp, kern, _capi.last_kernel())
        assert "+ tail" not in out["0"][3] and "+ tail" in out["1"][3] and "+ tail" in out["1"][4], out["1"][3:]
        assert np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1])
        assert np.array_equal(out["0"][2], out["1"][2])
        if not same:
            rows = np.sort(rng.choice(nq, size=300, replace=False))
            od, oi = orc.knn_brute(X[rows], Y, K)
            assert _rel(out["1"][0][rows], od) < DIST_RTOL and np.array_equal(out["1"][1][rows], oi)
    finally:
        _capi.set_sym_mode(sym_before)
        _capi.set_prune_mode(0)


@pytest.mark.parametrize("nq,nr,d,K,same", [(525724, 30000, 15, 4, False), (600000, 600000, 6, 3, True), (530000, 20000, 1, 1, False), (524289, 45000, 10, 2, False),
                                            (246000, 60000, 15, 4, False)])        # (481 blocks: just over the threshold, padded to 482)
def test_wide_sweep_is_bit_identical(nq, nr, d, K, same, monkeypatch):
    """Round 5: one-k-step sweeps of short lists over 480 or more query blocks run FOUR query tiles per wave (a workgroup =
    two query blocks; knn_f16.hpp, QTT = 4).  Neighbours, distances and the fused sums must equal the two-tile kernel's
    (MCE_WIDE=0) to the bit -- odd block counts (padded to even), self-exclusion on one buffer, d = 1 -- and the oracle's on
    sampled rows."""
    from mcevidence_amd import _capi
    _capi.set_search_mode(_capi.MODE_AUTO)
    _capi.set_prune_mode(_capi.PRUNE_OFF)
    sym_before = _capi.get_sym_mode()
    _capi.set_sym_mode(_capi.SYM_OFF)
    try:
        rng = np.random.default_rng(nq + d)
        Y = rng.standard_normal((nr, d))
        X = Y if same else rng.standard_normal((nq, d))
        if same:
            Y[rng.integers(0, nr, 50)] = Y[rng.integers(0, nr, 50)]          # exact duplicates: ties
        w = rng.integers(1, 5, size=nq).astype(float)
        fs = -rng.random(nq)
        sm = _capi.SELF_EXCLUDE if same else _capi.SELF_NONE
        k0 = 1 if same else 0
        out = {}
        for flag in ("0", "1"):
            monkeypatch.setenv("MCE_WIDE", flag)
            dist, idx = _capi.knn(X, Y, K, self_mode=sm)
            kern = _capi.last_kernel()
            dotp = _capi.knn_dotp(X, Y, w, fs, K + k0, k0)
            out[flag] = (dist, idx, dotp, kern)
        assert " wide" not in out["0"][3] and " wide" in out["1"][3] and "qt=4" in out["1"][3], (out["0"][3], out["1"][3])
        assert np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1])
        assert np.array_equal(out["0"][2], out["1"][2])
        rows = np.sort(rng.choice(nq, size=400, replace=False))
        rows[-1] = nq - 1                                                     # the last row of the last (half-filled) workgroup
        od, oi = orc.knn_brute(X[rows], Y, K + k0)
        if k0:
            od, oi = od[:, 1:], oi[:, 1:]
        assert _rel(out["1"][0][rows], od) < DIST_RTOL
        if not same:
            assert np.array_equal(out["1"][1][rows], oi)
    finally:
        _capi.set_sym_mode(sym_before)
        _capi.set_prune_mode(_capi.PRUNE_AUTO)


def test_knn_large_offsets_are_stable(capi):
    """un-whitened data far from the origin: GEMM-form cancellation stays within tolerance."""
    rng = np.random.default_rng(5)
    Y = 50.0 + rng.standard_normal((5000, 6))
    d, i = capi.knn(Y, Y, 5, self_mode=capi.SELF_EXCLUDE)
    od, oi = orc.knn_brute(Y, Y, 5, self_mode=2)
    assert _rel(d, od) < 1e-8 and np.mean(i == oi) > 0.999


ADVERSARIAL = {
    # name: (generator, n, d) -- inputs chosen to stress the fp16 filter's bound: dynamic range,
    # clustering far below the fp16 resolution of the extent, offsets, fp16 over/underflow before scaling
    "heavy_tails": lambda r, n, d: r.standard_t(1.5, size=(n, d)),
    "tight_clusters": lambda r, n, d: r.integers(0, 3, size=(n, 1)) * 1000.0 + 1e-3 * r.standard_normal((n, d)),
    "tiny_scale": lambda r, n, d: 1e-9 * r.standard_normal((n, d)),
    "huge_scale_offset": lambda r, n, d: 1e7 + 3e4 * r.standard_normal((n, d)),
    "anisotropic": lambda r, n, d: r.standard_normal((n, d)) * np.logspace(-4, 2, d)[None, :],
    "one_outlier": lambda r, n, d: np.vstack([r.standard_normal((n - 1, d)), np.full((1, d), 1e4)]),
    "lattice_ties": lambda r, n, d: r.integers(-3, 4, size=(n, d)).astype(float),
    "subnormal_fp16_coords": lambda r, n, d: np.hstack([r.standard_normal((n, 1)), 1e-6 * r.standard_normal((n, d - 1))]),
    "all_identical": lambda r, n, d: np.full((n, d), 3.25),
    "constant_column": lambda r, n, d: np.hstack([np.full((n, 1), -7.0), r.standard_normal((n, d - 1))]),
    "few_distinct": lambda r, n, d: r.standard_normal((5, d))[r.integers(0, 5, n)],
}


@pytest.mark.parametrize("kind", sorted(ADVERSARIAL))
@pytest.mark.parametrize("d", [1, 2, 5, 14, 15, 31])
def test_knn_adversarial_inputs_stay_exact(capi, kind, d):
    """the filter must never drop a true neighbour, whatever the data look like: the result is
    compared with the exact CPU search (distances AND neighbour sets)."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(("%s-%d" % (kind, d)).encode()))
    n, K = 4000, 6
    if d == 1 and kind in ("anisotropic", "subnormal_fp16_coords", "constant_column"):
        pytest.skip("needs at least two columns")
    Y = np.ascontiguousarray(ADVERSARIAL[kind](rng, n, d), dtype=np.float64)
    dist, idx = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
    od, oi = orc.knn_brute(Y, Y, K, self_mode=2)
    # whatever was selected, the reported distance is the exact distance to the reported row
    exact = np.sqrt(((Y[:, None, :] - Y[idx]) ** 2).sum(-1))
    assert np.allclose(dist, exact, rtol=1e-13, atol=0)
    if kind in ("all_identical", "few_distinct"):
        assert np.array_equal(dist, od)              # massive exact ties: distances must still be exact
    elif capi.get_search_mode() == capi.MODE_F64 and (d == 1 or kind in ("tight_clusters", "lattice_ties", "huge_scale_offset")):
        # documented limit of the fp64 GEMM-form sweep: it SELECTS with ~1e-16*R^2
        # absolute accuracy (R = extent about the mean), so neighbours closer together than that, or
        # exactly tied, may be swapped for an equally-near row; the default fp16-filter path is exact.
        if kind != "tight_clusters":      # (clusters 1e6 sigma apart: the GEMM form cannot rank inside a cluster)
            assert np.all(dist <= od * (1 + 1e-3) + 1e-12)
    else:
        assert np.array_equal(dist, od) or _rel(dist, od) < 1e-13
        same = np.sort(idx, axis=1) == np.sort(oi, axis=1)
        assert np.mean(same) > 0.999          # identical sets except exact distance ties
    assert np.all(np.diff(dist, axis=1) >= 0)


@pytest.mark.parametrize("n,nq,d,K,self_mode", [(3001, 3001, 128, 7, 2), (3001, 3001, 129, 1, 2), (5000, 777, 160, 14, 0), (4000, 4000, 200, 15, 1), (2500, 2500, 256, 30, 2),
                                                (2100, 100, 511, 32, 0), (1500, 1500, 1024, 5, 2), (40000, 40000, 130, 9, 2), (33, 33, 128, 32, 2), (70000, 300, 300, 6, 0),
                                                (300, 70000, 128, 3, 0)])
def test_long_rows_on_the_blocked_fp64_sweep(capi, n, nq, d, K, self_mode):
    """128 <= d <= 1024 (round 6): the fp64 MFMA sweep with the k dimension in blocks of 32 (knn_long.hpp) --
    several query blocks, reference splits, ragged ends, every list length, the self row in and out.  GEMM-form keys select K + 2
    candidates, the merge picks the K on exact direct-difference distances: the oracle's rows, all of them."""
    rng = np.random.default_rng(n + d + K)
    Y = rng.standard_normal((n, d))
    X = Y if self_mode else rng.standard_normal((nq, d))
    dist, idx = capi.knn(X, Y, K, self_mode=self_mode)
    assert "knn_long_kernel<KCAP=%d>" % (8 if K <= 6 else 16 if K <= 14 else 32) in capi.last_kernel(), capi.last_kernel()
    od, oi = orc.knn_brute(X, Y, K, self_mode=2 if self_mode == 2 else 0)
    assert np.array_equal(idx, oi) and _rel(dist, od) < 1e-13          # (the merge's fma chain against the oracle's plain sum: last-ulp)
    if self_mode == 1:
        assert np.all(dist[:, 0] == 0.0) and np.array_equal(idx[:, 0], np.arange(n))
    if self_mode == 2 and n <= 5000 and d <= 200:          # (beyond: r^d overflows the oracle's literal volume)
        # the fused call and the partitioned entry point on the same path
        w = rng.integers(1, 5, n).astype(np.float64)
        fs = -rng.random(n)
        dotp = capi.knn_dotp(Y, None, w, fs, K + 1, 1)
        want = orc.dotp_literal(np.concatenate([np.zeros((n, 1)), od], axis=1), w, fs, d, 1, K + 1)
        assert np.allclose(dotp[1:], want[1:], rtol=1e-12, atol=0.0)
        parts = sum(capi.knn_dotp_part(Y, w, fs, K + 1, p, 3) for p in range(3))
        assert np.allclose(parts[1:], dotp[1:], rtol=1e-13, atol=0.0)


def test_ab_switches_bring_the_older_kernels_back(tmp_path):
    """MCE_LONG=0 / MCE_DEEP=0 (read once per process: a child process each) put 128 <= d and 64 <= d <= 127 back on the kernels that
    served them before round 6 -- the vector-FMA kernel and the fp64 sweep's wide form: the same rows, distances equal to the last
    ulp (exact distances summed in another order)."""
    import os
    import subprocess
    import sys
    from helpers import REPO
    code = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from mcevidence_amd import _capi\n"
        "out = {}\n"
        "for d, K in ((130, 5), (100, 20), (70, 3)):\n"
        "    Y = np.random.default_rng(d + K).standard_normal((2500, d))\n"
        "    dist, idx = _capi.knn(Y, Y, K, self_mode=_capi.SELF_EXCLUDE)\n"
        "    np.save(sys.argv[1] + '_%%d_d.npy' %% d, dist); np.save(sys.argv[1] + '_%%d_i.npy' %% d, idx)\n"
        "    out[str(d)] = _capi.last_kernel()\n"
        "print(json.dumps(out))\n") % REPO
    kernels = {}
    for tag, env_extra in (("new", {}), ("old", {"MCE_LONG": "0", "MCE_DEEP": "0"})):
        env = {k: v for k, v in os.environ.items() if k not in ("MCE_LONG", "MCE_DEEP")}
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path / tag)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        import json
        kernels[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert "knn_long_kernel" in kernels["new"]["130"] and "knn_generic_kernel" in kernels["old"]["130"]
    assert "knn_deep_kernel" in kernels["new"]["100"] and "two passes" in kernels["new"]["100"] and "knn_mfma_kernel<KS=28" in kernels["old"]["100"]
    assert "knn_deep_kernel<KST=5" in kernels["new"]["70"] and "knn_mfma_kernel<KS=20" in kernels["old"]["70"]
    for d in (130, 100, 70):
        dn, do = np.load(str(tmp_path / "new") + "_%d_d.npy" % d), np.load(str(tmp_path / "old") + "_%d_d.npy" % d)
        assert np.array_equal(np.load(str(tmp_path / "new") + "_%d_i.npy" % d), np.load(str(tmp_path / "old") + "_%d_i.npy" % d))
        assert _rel(dn, do) < 1e-14


@pytest.mark.parametrize("d,K,kernel", [(64, 5, "knn_deep_kernel<KST=5"), (79, 3, "knn_deep_kernel<KST=5"), (80, 20, "knn_deep_kernel<KST=6,KCAP=16"), (100, 12, "knn_deep_kernel<KST=8"),
                                        (90, 16, "knn_deep_kernel<KST=6"), (127, 32, "knn_deep_kernel<KST=8,KCAP=16"), (128, 6, "knn_long_kernel<KCAP=8"), (10, 40, "generic"), (200, 33, "generic"), (100, 33, "generic"),
                                        (129, 14, "knn_long_kernel<KCAP=16"), (200, 32, "knn_long_kernel<KCAP=32"), (300, 15, "knn_long_kernel<KCAP=32"), (1024, 3, "knn_long_kernel<KCAP=8")])
def test_beyond_the_filter_kernels_limits(capi, d, K, kernel):
    """64 <= d <= 127: the DEEP fp16 filter (round 6: 5, 6 or 8 k-steps; K <= 16 in one pass, 17..32 in two) or -- search mode 1 -- the fp64 MFMA
    sweep at KS = 20..32, one query tile per wave (round 5: the vector-FMA kernel took 66x the time of d = 63).  128 <= d <= 1024
    (round 6): the fp64 MFMA sweep with the k dimension in blocks (knn_long.hpp).  K > 32: the plain exact kernel -- no shape the
    reference accepts is refused."""
    rng = np.random.default_rng(d * 7 + K)
    Y = rng.standard_normal((1500, d))
    X = rng.standard_normal((333, d))
    dist, idx = capi.knn(X, Y, K)
    if "deep" in kernel and capi.get_search_mode() == capi.MODE_F64:
        kernel = "knn_mfma_kernel<KS=%d" % (4 * ((d + 1 + 15) // 16))
    assert kernel in capi.last_kernel(), capi.last_kernel()
    if d <= 128 and K <= 32:          # the run-time certificate covers these shapes too: an independent exact scan agrees
        assert capi.verify_knn(X, Y, dist, nsample=200) == 0
        bad = dist.copy(); bad[7, K - 1] *= 0.99
        assert capi.verify_knn(X, Y, bad, nsample=len(X)) == 1
    od, oi = orc.knn_brute(X, Y, K) if K <= 64 else (None, None)
    assert _rel(dist, od) < 1e-13 and np.array_equal(idx, oi)
    d1, i1 = capi.knn(Y, Y, K, self_mode=capi.SELF_INCLUDE)
    d2, i2 = capi.knn(Y, Y, K - 1, self_mode=capi.SELF_EXCLUDE)
    # (K = 33 is the vector-FMA kernel, K - 1 = 32 at d = 100 the deep filter: both exact, summed in different orders -- last ulp)
    assert np.all(d1[:, 0] == 0) and (np.array_equal(d1[:, 1:], d2) or (K == 33 and _rel(d1[:, 1:], d2) < 1e-14)) and np.array_equal(i1[:, 1:], i2)
    kmax = K if K <= 12 else 12
    w = rng.integers(1, 4, len(Y)).astype(float)
    fs = -rng.random(len(Y))
    dp, dd = capi.knn_dotp(Y, None, w, fs, kmax, 1, return_dist=True)
    full = np.zeros((len(Y), kmax)); full[:, 1:] = d2[:, :kmax - 1]
    # (log-domain oracle: at d=200 the reference's own pow(pi,d/2)*pow(r,d) overflows to inf)
    ref = orc.dotp_logdomain(full, w, fs, d, 1, kmax)
    assert np.allclose(dp[1:], ref[1:], rtol=1e-11) and np.allclose(dd, d2[:, :kmax - 1], rtol=1e-14, atol=0)
    assert np.allclose(capi.knn_dotp(Y, None, w, fs, kmax, 1), dp, rtol=1e-13)


def test_empty_and_minimal_inputs(capi):
    Y = np.random.default_rng(0).standard_normal((40, 3))
    d, i = capi.knn(np.zeros((0, 3)), Y, 4)                    # no queries: empty result, no error
    assert d.shape == (0, 4) and i.shape == (0, 4)
    d, i = capi.knn(Y[:1], Y, 40)                              # K == number of reference rows
    od, oi = orc.knn_brute(Y[:1], Y, 40)
    assert np.allclose(d, od, rtol=1e-13) and np.array_equal(i, oi)
    d, i = capi.knn(Y, Y[:2], 1)                               # two reference rows
    od, oi = orc.knn_brute(Y, Y[:2], 1)
    assert np.allclose(d, od, rtol=1e-13) and np.array_equal(i, oi)
    with pytest.raises(ValueError):
        capi.knn_dotp(np.zeros((0, 3)), Y, np.zeros(0), np.zeros(0), 3, 0)     # an empty chain is an error (as in the reference)


def test_error_codes_on_gpu(capi):
    X = np.zeros((5, 3))
    with pytest.raises(ValueError):
        capi.knn(X, X, 6)
    with pytest.raises(ValueError):
        capi.knn(X, X, 2, device=99)


# --------------------------------------------------------------------------- reduction
@pytest.mark.parametrize("n,d,kmax,k0", [(6000, 6, 5, 1), (5000, 27, 10, 1), (4000, 15, 4, 0), (3000, 2, 2, 1), (2000, 33, 6, 0)])
def test_fused_and_unfused_dotp(capi, n, d, kmax, k0):
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d))
    Y = X if k0 == 1 else rng.standard_normal((n + 37, d))
    w = rng.integers(1, 6, n).astype(float)
    fs = -0.5 * (X ** 2).sum(1)
    fs -= fs.max()
    dp, dist = capi.knn_dotp(X, None if k0 == 1 else Y, w, fs, kmax, k0, return_dist=True)
    od, _ = orc.knn_brute(X, Y, kmax - k0, self_mode=2 if k0 == 1 else 0)
    full = np.zeros((n, kmax))
    full[:, k0:] = od
    ref = orc.dotp_literal(full, w, fs, d, k0, kmax)
    assert np.allclose(dp[k0:], ref[k0:], rtol=1e-11) and np.all(dp[:k0] == 0)
    assert _rel(dist, od) < DIST_RTOL
    un = capi.dotp(full, w, fs, d, k0, kmax)
    assert np.allclose(un[k0:], ref[k0:], rtol=1e-12)
    # bitwise reproducible run to run (fixed-order reduction, no atomics)
    dp2 = capi.knn_dotp(X, None if k0 == 1 else Y, w, fs, kmax, k0)
    assert np.array_equal(dp, dp2)


def test_dotp_zero_distance_terms(capi):
    rng = np.random.default_rng(8)
    dist = np.abs(rng.standard_normal((500, 4)))
    dist[7, 1] = 0.0
    out = capi.dotp(dist, np.ones(500), np.zeros(500), 6, 1, 4)
    ref = orc.dotp_literal(dist, np.ones(500), np.zeros(500), 6, 1, 4)
    assert np.all(np.isfinite(out)) and np.allclose(out[1:], ref[1:], rtol=1e-12)


def test_dotp_signed_weights_and_zero_likelihood_rows(capi):
    """negative weights keep the reference's signed term; fs = -inf rows contribute zero (fused and unfused)"""
    rng = np.random.default_rng(12)
    n, d, k0, kmax = 3000, 5, 1, 4
    X = rng.standard_normal((n, d))
    w = rng.integers(1, 5, n).astype(float)
    w[[3, 77, 2000]] = [-2.0, -1.0, -0.5]
    fs = -rng.random(n)
    fs[[9, 77, 1500]] = -np.inf
    od, _ = orc.knn_brute(X, X, kmax - k0, self_mode=2)
    full = np.zeros((n, kmax))
    full[:, k0:] = od
    ref = orc.dotp_literal(full, w, fs, d, k0, kmax)
    dp = capi.knn_dotp(X, None, w, fs, kmax, k0)
    un = capi.dotp(full, w, fs, d, k0, kmax)
    assert np.all(np.isfinite(dp)) and np.allclose(dp[k0:], ref[k0:], rtol=1e-11) and np.allclose(un[k0:], ref[k0:], rtol=1e-12)
    with pytest.raises(ValueError):
        capi.knn_dotp(X, None, w, np.where(np.arange(n) == 4, np.inf, fs), kmax, k0)
    with pytest.raises(ValueError):
        capi.knn_dotp(X, None, w, np.where(np.arange(n) == 4, np.nan, fs), kmax, k0)


def test_query_sharding_sums_to_full(capi):
    """shard the queries, add the partial sums."""
    rng = np.random.default_rng(9)
    n, d, kmax = 20011, 6, 5
    X = rng.standard_normal((n, d))
    w = np.ones(n)
    fs = -rng.random(n)
    full = capi.knn_dotp(X, None, w, fs, kmax, 1)
    parts = np.zeros(kmax)
    for lo, hi in ((0, 5000), (5000, 13337), (13337, n)):
        parts += capi.knn_dotp(X[lo:hi], X, w[lo:hi], fs[lo:hi], kmax, 1, self_offset=lo)
    assert np.allclose(parts, full, rtol=1e-13)


def test_single_process_multi_device_split(capi):
    """mce_knn_dotp_f64(devices, ndev): one host thread per listed device, rows split evenly, partials
    added on the host in device order.  With one GPU on the box the same ordinal is listed 3 times:
    the sharding, the self-exclusion offsets and the sum are exercised all the same."""
    rng = np.random.default_rng(21)
    n, d, kmax = 30011, 7, 5
    X = rng.standard_normal((n, d))
    w = rng.integers(1, 4, n).astype(float)
    fs = -rng.random(n)
    one, dist1 = capi.knn_dotp(X, None, w, fs, kmax, 1, return_dist=True)
    three, dist3 = capi.knn_dotp(X, None, w, fs, kmax, 1, return_dist=True, devices=[0, 0, 0])
    assert np.allclose(three, one, rtol=1e-13) and np.array_equal(dist1, dist3)
    Y = rng.standard_normal((20000, d))
    assert np.allclose(capi.knn_dotp(X, Y, w, fs, kmax, 0, devices=[0, 0]), capi.knn_dotp(X, Y, w, fs, kmax, 0), rtol=1e-13)
    with pytest.raises((ValueError, RuntimeError)):
        capi.knn_dotp(X, None, w, fs, kmax, 1, devices=[0, 7])


# --------------------------------------------------------------------------- the class, against the reference's outputs
@pytest.mark.parametrize("name", sorted(G))
def test_class_on_gpu_reproduces_reference(name, capi):
    import mcevidence_amd as pkg
    case = G[name]
    if case["tag"] in ("big", "c5") and capi.get_search_mode() == capi.MODE_F64 and case["ndim"] < 10:
        pytest.skip("1M x 6 / 10M x 6 through the fp64 sweep is covered by the auto mode")
    mce = build_mce(case)
    assert mce.backend.name == "hip"
    lnE = mce.evidence(**case["ev"])
    assert np.max(np.abs(lnE - np.array(case["lnE"]))) < LNE_TOL, (lnE, case["lnE"])


@pytest.mark.parametrize("name", [n for n in sorted(G) if G[n]["tag"] not in ("big", "c4", "c5")])
def test_device_feeder_route_matches_reference_and_host_route(name):
    """covariance + whitening on the device (mce_evidence_feed_f64) vs the reference's outputs, and vs
    the host-feeder route of the same class."""
    import mcevidence_amd as pkg
    case = G[name]
    calls = {}

    class Spy(pkg.HipBackend):
        def evidence_feed(self, *a, **k):
            calls["feed"] = calls.get("feed", 0) + 1
            return super().evidence_feed(*a, **k)

    def run(backend):
        return build_mce(case, backend=backend).evidence(**case["ev"])

    dev = run(Spy())
    host = run(type("H", (), {"name": "hip", "knn_dotp": pkg.HipBackend().knn_dotp})())
    cov = case["ev"].get("covtype", "all")
    if cov is None:
        cov = case["mce"].get("covtype", "single")
    two_systems = cov == "single" and case["mce"].get("split", False)   # keeps np.linalg.eig's conventions: host route
    assert calls.get("feed", 0) == (1 if cov in ("all", "single") and not two_systems else 0)
    assert np.max(np.abs(dev - np.array(case["lnE"]))) < LNE_TOL
    assert np.max(np.abs(dev - host)) < 1e-10


@pytest.mark.parametrize("d,split", [(64, False), (90, False), (127, False), (80, True)])
def test_device_feeders_beyond_63_dimensions(d, split):
    """64 <= d <= 127 (round 5): covariance (pairs in batches of 2048), the Jacobi eigen-system and the wide whitening kernel on the
    device, the search on the deep fp16 filter (round 6; round 5: the fp64 sweep's wide form) -- the class no longer leaves the
    device-feeder route at d = 64.  Same ln E as
    the host-feeder route (np.cov + np.linalg.eig + whitening on the host) of the same class; d = 128 is refused by the feeders and
    served by the host route."""
    import mcevidence_amd as pkg
    from mcevidence_amd import _capi
    from mcevidence_amd.synth import gaussian_chain
    chain = gaussian_chain(seed=d, n=24000, d=d, weights="int", cov="corr")
    calls = {}

    class Spy(pkg.HipBackend):
        def evidence_feed(self, *a, **k):
            out = super().evidence_feed(*a, **k)
            calls["feed"] = calls.get("feed", 0) + (out is not None)
            return out

    def run(backend):
        m = pkg.MCEvidence([chain], kmax=4, verbose=0, backend=backend)
        if split:
            m.set_split(np.arange(0, 12000), np.arange(12000, 24000))
        return m.evidence(covtype="all")

    with _capi.options(search_mode=_capi.MODE_AUTO):          # (whatever the module's `capi` fixture has set process-wide)
        dev = run(Spy())
        assert calls.get("feed", 0) == 1 and "knn_deep_kernel<KST=" in _capi.last_kernel(), _capi.last_kernel()
    with _capi.options(search_mode=_capi.MODE_F64):          # ... and on the fp64 sweep's wide form when asked for
        dev64 = run(Spy())
        assert "knn_mfma_kernel<KS=" in _capi.last_kernel(), _capi.last_kernel()
    assert np.max(np.abs(dev - dev64)) < 1e-10
    host = run(type("H", (), {"name": "hip", "knn_dotp": pkg.HipBackend().knn_dotp})())
    assert np.all(np.isfinite(dev)) and np.max(np.abs(dev - host)) < 1e-9, (dev, host)
    if d == 127:
        wide = gaussian_chain(seed=1, n=3000, d=128, cov="corr")
        with pytest.raises(ValueError):
            _capi.evidence_feed(wide[:, 2:], None, 128, 0, 3, np.ones(3000), np.zeros(3000))
        calls.clear()
        lnE = pkg.MCEvidence([wide], kmax=3, verbose=0, backend=Spy()).evidence()
        assert calls.get("feed", 0) == 0 and np.all(np.isfinite(lnE)) and "knn_long_kernel" in _capi.last_kernel()


def _feed_problems(rng, count):
    """mixed bag of small evidence problems: auto/cross, both covariance modes, ragged sizes, padded rows"""
    probs = []
    for i in range(count):
        d = int(rng.integers(1, 9)) if i % 7 else int(rng.integers(9, 30))
        n1 = int(rng.integers(40, 3000))
        kmax = int(rng.integers(2, 7))
        A = rng.standard_normal((d, d)) + 2.0 * np.eye(d)
        ld = d + int(rng.integers(0, 4))                      # nuisance columns behind the first d
        S1 = np.zeros((n1, ld))
        S1[:, :d] = rng.standard_normal((n1, d)) @ A + rng.standard_normal(d) * 5
        S1[:, d:] = 1e6
        S2 = None
        if i % 3 == 1:
            n2 = int(rng.integers(40, 3000))
            S2 = rng.standard_normal((n2, d)) @ A
        w = rng.integers(1, 5, n1).astype(float)
        fs = -rng.random(n1) * 3
        probs.append((S1, S2, d, int(i % 2), kmax, w, fs))
    return probs


def test_batched_feed_of_planck_sized_chains_under_load(capi):
    """The Planck driver pattern (reference planck_mcevidence.py:306-348): hundreds of chains of 6 k - 100 k rows, d = 6-8,
    kmax = 2, searched concurrently on many streams.  Every problem must come out exactly as from a call of its own, every
    time: with several searches sharing the chip a wave once read a staging buffer that another wave's LDS-DMA was still
    filling (knn_f16.hpp: dma_barrier) -- a seed bound from half-landed rows, ~2 wrong sums per batch of 300."""
    rng = np.random.default_rng(0)
    probs = []
    for i in range(300):
        d = int(rng.integers(6, 9))
        n = int(np.exp(rng.uniform(np.log(6000), np.log(100000))))
        A = rng.standard_normal((d, d)) + 2 * np.eye(d)
        probs.append((rng.standard_normal((n, d)) @ A, None, d, 0, 2, rng.integers(1, 5, n).astype(float), -rng.random(n)))
    singles = [capi.evidence_feed(*p) for p in probs]
    assert all(np.all(np.isfinite(s[0])) for s in singles)
    for rep in range(4):
        batch = capi.evidence_feed_batch(probs)
        bad = [i for i, (a, b) in enumerate(zip(singles, batch)) if not (np.array_equal(a[0], b[0]) and a[1] == b[1])]
        assert not bad, (rep, bad[:5], [(singles[i][0], batch[i][0]) for i in bad[:2]])


def test_batched_feed_is_bit_identical_to_single_calls(capi, monkeypatch):
    """mce_evidence_feed_batch_f64: same dotp / J / eigenvalues as one
    mce_evidence_feed_f64 call per problem -- bit for bit -- in one wave and in several."""
    rng = np.random.default_rng(77)
    probs = _feed_problems(rng, 60)
    singles = [capi.evidence_feed(*p) for p in probs]
    for wave_bytes in (None, "3000000"):
        if wave_bytes:
            monkeypatch.setenv("MCE_FEED_WAVE_BYTES", wave_bytes)
        batch = capi.evidence_feed_batch(probs)
        assert len(batch) == len(probs)
        for (d1, j1, e1), (d2, j2, e2) in zip(singles, batch):
            assert np.array_equal(d1, d2) and j1 == j2 and np.array_equal(e1, e2)
    monkeypatch.delenv("MCE_FEED_WAVE_BYTES")
    # and against the oracle (NumPy feeders + exact CPU search)
    for p, (dotp, jac, ev) in list(zip(probs, batch))[:12]:
        S1, S2, d, cov_mode, kmax, w, fs = p
        from helpers import OracleFeedBackend
        odot, ojac = OracleFeedBackend().evidence_feed(S1, S2, d, cov_mode, kmax, w, fs)
        k0 = 0 if S2 is not None else 1
        assert np.allclose(dotp[k0:], odot[k0:], rtol=1e-9) and abs(jac - ojac) <= 1e-11 * ojac
    assert capi.evidence_feed_batch([]) == []


def test_batched_feed_isolates_failing_problems(capi):
    rng = np.random.default_rng(78)
    probs = _feed_problems(rng, 9)
    good = capi.evidence_feed_batch(probs)
    bad = list(probs)
    S1, S2, d, cm, kmax, w, fs = bad[4]
    bad[4] = (S1[:3], None, d, cm, 9, w[:3], fs[:3])            # kmax-1 = 8 neighbours from 2 usable rows
    dup = np.concatenate([bad[6][0][:, :2], bad[6][0][:, :1]], axis=1)   # third column == first: singular covariance
    bad[6] = (np.ascontiguousarray(dup), None, 3, 0, 3, bad[6][5], bad[6][6])
    with pytest.raises(ValueError, match="problem 4: Expected n_neighbors <= n_samples_fit"):
        capi.evidence_feed_batch(bad)
    out = capi.evidence_feed_batch(bad, return_exceptions=True)
    assert isinstance(out[4], ValueError)
    for i in (0, 1, 2, 3, 5, 7, 8):
        assert np.array_equal(out[i][0], good[i][0]) and out[i][1] == good[i][1]
    # a (numerically) singular covariance either raises "math domain error" or yields a tiny positive
    # eigenvalue, as NumPy's eig would; it must not disturb its neighbours either way
    assert isinstance(out[6], ValueError) or out[6][2].min() < 1e-9 * out[6][2].max()


@pytest.mark.parametrize("nobj", [25])
def test_evidence_many_matches_evidence_one_by_one(nobj):
    """the class-level batch (reference pattern: planck_mcevidence.py:306-348) against per-object
    evidence() and against the reference's golden numbers."""
    import mcevidence_amd as pkg
    names = [n for n in sorted(G) if G[n]["tag"] == "small" and G[n]["seed_split"] is None
             and G[n]["ev"].get("covtype", "all") in ("all", "single")][:nobj]
    assert len(names) >= 3

    def objs():
        return [pkg.MCEvidence([chain_of(G[n])], verbose=0, **G[n]["mce"]) for n in names]
    # the goldens use different evidence() kwargs per case: group by kwargs
    by_kw = {}
    for n in names:
        by_kw.setdefault(tuple(sorted(G[n]["ev"].items())), []).append(n)
    for kw, group in by_kw.items():
        ms = [pkg.MCEvidence([chain_of(G[n])], verbose=0, **G[n]["mce"]) for n in group]
        many = pkg.evidence_many(ms, **dict(kw))
        for n, m, got in zip(group, ms, many):
            one = m.evidence(**dict(kw))
            assert np.array_equal(got, one)
            assert np.max(np.abs(got - np.array(G[n]["lnE"]))) < LNE_TOL
    # Planck-shaped farm: many chains, 6 parameters, kmax = 2, info=True
    from mcevidence_amd.synth import gaussian_chain
    ms = [pkg.MCEvidence([gaussian_chain(seed=900 + i, n=3000 + 137 * i, d=6)], kmax=2, verbose=0, priorvolume=1.0 + i) for i in range(40)]
    many = pkg.evidence_many(ms, info=True)
    for m, (lnE, info) in zip(ms, many):
        one, info1 = m.evidence(info=True)
        assert np.array_equal(lnE, one) and info is m.info and info1 is m.info


# --------------------------------------------------------------------------- spatial pruning
@pytest.fixture()
def prune_modes():
    from mcevidence_amd import _capi
    _capi.set_search_mode(_capi.MODE_AUTO)
    yield _capi
    _capi.set_prune_mode(_capi.PRUNE_AUTO)


def _both(capi, fn):
    capi.set_prune_mode(capi.PRUNE_OFF)
    a = fn()
    ka = capi.last_kernel()
    capi.set_prune_mode(capi.PRUNE_FORCE)
    b = fn()
    kb = capi.last_kernel()
    assert "pruned" in kb and "pruned" not in ka, (ka, kb)
    return a, b


@pytest.mark.parametrize("n,d,K", [(5000, 1, 3), (20000, 2, 5), (40000, 3, 1), (60000, 6, 10), (33333, 6, 16), (25000, 9, 4), (20000, 13, 7),
                                   (45000, 5, 8), (30000, 7, 4), (50000, 8, 12), (64000, 4, 9)])   # (every per-query reach variant: d = 1 .. 8, odd and even)
def test_pruned_search_is_bit_identical_auto(prune_modes, n, d, K):
    """k-d ordered, box-pruned walk vs the exhaustive sweep: same distances,
    same neighbour rows, same order -- bit for bit -- with the own row excluded, included, ignored."""
    capi = prune_modes
    rng = np.random.default_rng(n + d)
    Y = rng.standard_normal((n, d)) @ (np.eye(d) + 0.3 * rng.standard_normal((d, d)))
    for sm in (capi.SELF_EXCLUDE, capi.SELF_INCLUDE, capi.SELF_NONE):
        (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, K, self_mode=sm))
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    od, oi = orc.knn_brute(Y[:3000], Y, K, self_mode=2)
    assert _rel(d1[:3000] if sm == capi.SELF_EXCLUDE else capi.knn(Y[:3000], Y, K, self_mode=capi.SELF_EXCLUDE)[0], od) < DIST_RTOL


@pytest.mark.parametrize("n,d,K", [(70001, 3, 9), (150000, 6, 9), (90000, 6, 5), (60000, 2, 9), (80000, 5, 10)])
def test_pruned_heavy_waves_and_short_lists_are_bit_identical(prune_modes, n, d, K, monkeypatch):
    """Round 4's two variations of the pruned walk against the exhaustive sweep, bit for bit: (a) HEAVY waves -- the first
    waves of the dispatch order served by S workgroups each, lists folded afterwards -- for every S, for a handful of waves,
    for the library's own count and for as many as the side arrays hold; (b) for K = 9 and 10 both instantiations -- K list
    entries in registers (three waves per SIMD, the library's choice) and twelve (two waves, MCE_PRUNE_LISTS=long) -- with and
    without heavy waves.  Same buffer as X and Y (auto evidence: the heavy split applies to one set only)."""
    capi = prune_modes
    rng = np.random.default_rng(n + 7 * d + K)
    Y = rng.standard_normal((n, d)) @ (np.eye(d) + 0.3 * rng.standard_normal((d, d)))
    Y[rng.integers(0, n, 40)] *= 6.0                      # a few far outliers: the waves that hold them are the heavy ones
    capi.set_prune_mode(capi.PRUNE_OFF)
    want_d, want_i = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
    capi.set_prune_mode(capi.PRUNE_FORCE)
    seen = set()
    for lists in ((None, "long") if K in (9, 10) else (None,)):
        for heavy in ("0", None, "auto", "5,2", "64,3", "300,8", "100000,4", "7,5"):      # (None: the library's default, off since round 4; "auto": its old rule)
            for name, val in (("MCE_PRUNE_HEAVY", heavy), ("MCE_PRUNE_LISTS", lists)):
                if val is None:
                    monkeypatch.delenv(name, raising=False)
                else:
                    monkeypatch.setenv(name, val)
            got_d, got_i = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
            k = capi.last_kernel()
            assert "pruned" in k, k
            seen.add((k.split("heavy=")[1]))
            assert np.array_equal(got_d, want_d) and np.array_equal(got_i, want_i), (heavy, lists, k)
    assert any(x.startswith("0x1") for x in seen) and any("x8" in x for x in seen) and any("x5" in x for x in seen)
    if K in (9, 10):        # (K = 10: the ten-entry instantiation)
        assert any(x.endswith("lists=%d" % K) for x in seen) and any(x.endswith("lists=12") for x in seen)


@pytest.mark.parametrize("n,d,K", [(5000, 1, 3), (70001, 3, 9), (150000, 6, 10), (40000, 6, 2), (33000, 10, 16)])
def test_pruned_search_same_buffer_is_bit_identical(prune_modes, n, d, K):
    """queries and references in ONE device buffer (what evidence() passes): the k-d order is shared, the
    walk starts from the tiles next to the wave's own (bootstrap) -- results still bit-identical, at the
    array ends too."""
    import torch
    capi = prune_modes
    rng = np.random.default_rng(n * 3 + d)
    Yh = rng.standard_normal((n, d)) @ (np.eye(d) + 0.2 * rng.standard_normal((d, d)))
    Y = torch.from_numpy(Yh).cuda()
    dist = torch.empty((n, K), dtype=torch.float64, device="cuda")
    idx = torch.empty((n, K), dtype=torch.int64, device="cuda")

    def run(sm):
        wsb = capi.knn_workspace_bytes(n, n, d, K)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        capi.knn_dev(Y.data_ptr(), n, Y.data_ptr(), n, d, K, sm, 0, dist.data_ptr(), idx.data_ptr(), ws.data_ptr(), wsb, 0)
        torch.cuda.synchronize()
        return dist.cpu().numpy().copy(), idx.cpu().numpy().copy()
    for sm in (capi.SELF_EXCLUDE, capi.SELF_INCLUDE):
        (d0, i0), (d1, i1) = _both(capi, lambda: run(sm))
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    od, oi = orc.knn_brute(Yh[-2500:], Yh, K, self_mode=1)
    assert _rel(d1[-2500:], od) < DIST_RTOL and np.mean(i1[-2500:] == oi) > 0.999


@pytest.mark.parametrize("nq,nr,d,K", [(7000, 50000, 4, 6), (513, 200000, 6, 3), (30000, 9000, 3, 12), (100, 5000, 8, 2)])
def test_pruned_search_is_bit_identical_cross_and_shards(prune_modes, nq, nr, d, K):
    capi = prune_modes
    rng = np.random.default_rng(nq + nr)
    Y = rng.standard_normal((nr, d))
    X = rng.standard_normal((nq, d)) * 1.3 + 0.2
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(X, Y, K))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    od, oi = orc.knn_brute(X[:2000], Y, K)
    assert _rel(d1[:2000], od) < DIST_RTOL and np.array_equal(i1[:2000], oi)
    # a query shard of an auto search: rows [lo, hi) of Y with their own rows excluded by offset
    lo, hi = nr // 3, nr // 3 + min(nq, nr // 2)
    (s0, j0), (s1, j1) = _both(capi, lambda: capi.knn(Y[lo:hi], Y, K, self_mode=capi.SELF_EXCLUDE, self_offset=lo))
    assert np.array_equal(s0, s1) and np.array_equal(j0, j1)
    assert not np.any(j1 == np.arange(lo, hi)[:, None])


def test_pruned_search_duplicates_clusters_and_ties(prune_modes):
    """exact ties must break on the caller's row numbers, also after reordering; tight far-apart
    clusters make boxes degenerate; a constant column makes a zero-width dimension."""
    capi = prune_modes
    rng = np.random.default_rng(9)
    base = rng.standard_normal((3000, 4))
    Y = np.concatenate([base, base, base[:1500], rng.standard_normal((2000, 4)) * 1e-3 + 50.0, base[::-1]])
    Y[:, 3] = 1.25
    Y = np.ascontiguousarray(Y[rng.permutation(len(Y))])
    for K in (1, 4, 9):
        (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE))
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    od, oi = orc.knn_brute(Y[:1500], Y, 4, self_mode=2)
    g, gi = capi.knn(Y[:1500], Y, 4, self_mode=capi.SELF_EXCLUDE)
    assert np.array_equal(g, od) and np.array_equal(gi, oi)
    grid = np.stack(np.meshgrid(*[np.arange(12.0)] * 4), -1).reshape(-1, 4)     # 20736 lattice points: massive ties
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(grid, grid, 9, self_mode=capi.SELF_EXCLUDE))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)


@pytest.mark.parametrize("offset,scale,d,K", [(1.0e5, 1.0, 6, 9), (-3.0e4, 1.0e-2, 3, 4), (1.0e3, 1.0e3, 5, 8), (7.0, 1.0e-6, 2, 9)])
def test_pruned_search_far_from_the_origin_and_at_odd_scales(prune_modes, offset, scale, d, K):
    """The walk's box tests and its per-query reach test work on FLOAT coordinates with margins for what the conversion can be off
    (round 4, knn_f16.hpp: box_gap / query_reach): far from the origin a float's spacing is no longer small against the
    neighbour distances, so the margins are what keeps the pruning rigorous -- the lists must still be those of the exhaustive
    sweep, bit for bit (separate query set too: its own k-d order, the cross bootstrap)."""
    capi = prune_modes
    rng = np.random.default_rng(int(abs(offset)) + d)
    Y = (rng.standard_normal((70000, d)) @ (np.eye(d) + 0.3 * rng.standard_normal((d, d)))) * scale + offset
    Y[:, -1] += np.linspace(0.0, 50.0 * scale, len(Y))        # (one dimension spread out: elongated boxes)
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    X = Y[rng.choice(len(Y), 40000, replace=False)] + rng.standard_normal((40000, d)) * (0.1 * scale)
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(X, Y, K, self_mode=capi.SELF_NONE))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    od, oi = orc.knn_brute(X[:1000], Y, K, self_mode=0)
    assert _rel(d1[:1000], od) < DIST_RTOL and np.array_equal(i1[:1000], oi)


@pytest.mark.parametrize("k0", [1, 0])
def test_pruned_fused_reduction_and_class(prune_modes, k0):
    capi = prune_modes
    rng = np.random.default_rng(31 + k0)
    n, d, kmax = 120000, 6, 5
    X = rng.standard_normal((n, d))
    Y = None if k0 == 1 else rng.standard_normal((90000, d))
    w = rng.integers(1, 5, n).astype(float)
    fs = -rng.random(n) * 4
    (p0, q0), (p1, q1) = _both(capi, lambda: capi.knn_dotp(X, Y, w, fs, kmax, k0, return_dist=True))
    assert np.array_equal(q0, q1)                          # the distances that entered the sums
    assert np.allclose(p0, p1, rtol=1e-13, atol=0)         # sums: same terms, different association
    import mcevidence_amd as pkg
    case = G["auto_n100000_d6_k4_C2"]
    (a, b) = _both(capi, lambda: pkg.MCEvidence([chain_of(case)], verbose=0, **case["mce"]).evidence(**case["ev"]))
    assert np.max(np.abs(b - np.array(case["lnE"]))) < LNE_TOL and np.max(np.abs(a - b)) < 1e-12


def test_pruned_search_at_1M_x_6_matches_reference_golden(prune_modes):
    """N = 1M, D = 6: automatic mode picks the pruned walk; ln E against the reference's own output."""
    capi = prune_modes
    import mcevidence_amd as pkg
    case = G["auto_n1000000_d6_k4_unit"]
    lnE = pkg.MCEvidence([chain_of(case)], verbose=0, **case["mce"]).evidence(**case["ev"])
    assert "pruned" in capi.last_kernel()
    assert np.max(np.abs(lnE - np.array(case["lnE"]))) < LNE_TOL


def test_sampled_rows_at_full_size_C3(capi):
    """N = 1M, D = 27 (BASELINE configs[2]): 1500 sampled query rows against the exact CPU search,
    plus size-independent properties over all rows."""
    rng = np.random.default_rng(0)
    X = rng.standard_normal((1_000_000, 27))
    K = 9
    rows = np.sort(rng.choice(len(X), 1500, replace=False))
    dist, idx = capi.knn(X, X, K, self_mode=capi.SELF_EXCLUDE)
    assert np.all(np.diff(dist, axis=1) >= 0) and np.all(dist > 0)
    assert ("knn_f16" in capi.last_kernel()) == (capi.get_search_mode() != capi.MODE_F64)
    assert np.all(idx != np.arange(len(X))[:, None]) and idx.min() >= 0 and idx.max() < len(X)
    d_, i_ = orc.knn_brute(X[rows], X, K + 1)              # exact CPU search incl. self; drop column 0
    ods, ois = d_[:, 1:], i_[:, 1:]
    assert _rel(dist[rows], ods) < DIST_RTOL and np.array_equal(idx[rows], ois)
    # symmetry property: if j is i's nearest neighbour at distance r, then i is within r of j
    nn = idx[:, 0]
    assert np.all(dist[nn, 0] <= dist[:, 0] * (1 + 1e-12))


def test_filter_and_sweep_agree_on_every_row_at_full_size():
    """N = 1M, D = 27: the fp16-filter search and the fp64 sweep must return the same neighbours
    and distances for ALL 9M (query, rank) entries -- a whole-problem check of the filter's bound."""
    from mcevidence_amd import _capi
    rng = np.random.default_rng(123)
    X = rng.standard_normal((1_000_000, 27))
    _capi.set_search_mode(_capi.MODE_AUTO)
    d_f, i_f = _capi.knn(X, X, 9, self_mode=_capi.SELF_EXCLUDE)
    assert "knn_f16" in _capi.last_kernel()
    _capi.set_search_mode(_capi.MODE_F64)
    try:
        d_s, i_s = _capi.knn(X, X, 9, self_mode=_capi.SELF_EXCLUDE)
        assert "knn_mfma" in _capi.last_kernel()
    finally:
        _capi.set_search_mode(_capi.MODE_AUTO)
    assert np.array_equal(i_f, i_s)
    assert np.allclose(d_f, d_s, rtol=1e-14, atol=0)


def test_cross_at_full_size_C4():
    """N1 = N2 = 1M, D = 15, K = 5, separate query set (BASELINE configs[3]): sampled query rows against the exact
    CPU search; the fp16-filter search (seeded, one norm piece at d = 15) against the fp64 sweep on ALL rows."""
    from mcevidence_amd import _capi
    rng = np.random.default_rng(44)
    X = rng.standard_normal((1_000_000, 15))
    Y = rng.standard_normal((1_000_000, 15))
    Y[:3] = X[:3]                                                    # a few queries coincide with reference rows
    _capi.set_search_mode(_capi.MODE_AUTO)
    d_f, i_f = _capi.knn(X, Y, 5)
    assert "knn_f16" in _capi.last_kernel() and " seed=" in _capi.last_kernel()
    assert np.all(np.diff(d_f, axis=1) >= 0) and np.all(d_f[:3, 0] == 0.0) and i_f.min() >= 0 and i_f.max() < len(Y)
    rows = np.sort(rng.choice(len(X), 1000, replace=False))
    od, oi = orc.knn_brute(X[rows], Y, 5)
    assert _rel(d_f[rows], od) < DIST_RTOL and np.array_equal(i_f[rows], oi)
    _capi.set_search_mode(_capi.MODE_F64)
    try:
        d_s, i_s = _capi.knn(X, Y, 5)
        assert "knn_mfma" in _capi.last_kernel()
    finally:
        _capi.set_search_mode(_capi.MODE_AUTO)
    assert np.array_equal(i_f, i_s) and np.allclose(d_f, d_s, rtol=1e-14, atol=0)


def test_auto_at_full_size_C5(prune_modes):
    """N = 10M, D = 6, K = 9 (BASELINE configs[4]): the k-d pruned walk (automatic at this shape) against the
    exhaustive seeded sweep on ALL 90M (query, rank) entries, sampled rows against the exact CPU search, and the
    nearest-neighbour symmetry property."""
    capi = prune_modes
    rng = np.random.default_rng(55)
    X = rng.standard_normal((10_000_000, 6))
    K = 9
    capi.set_prune_mode(0)
    d_p, i_p = capi.knn(X, X, K, self_mode=capi.SELF_EXCLUDE)
    assert "pruned" in capi.last_kernel()
    assert np.all(np.diff(d_p, axis=1) >= 0) and np.all(i_p != np.arange(len(X))[:, None])
    nn = i_p[:, 0]
    assert np.all(d_p[nn, 0] <= d_p[:, 0] * (1 + 1e-12))
    rows = np.sort(rng.choice(len(X), 300, replace=False))
    od, oi = orc.knn_brute(X[rows], X, K + 1)
    assert _rel(d_p[rows], od[:, 1:]) < DIST_RTOL and np.array_equal(i_p[rows], oi[:, 1:])
    capi.set_prune_mode(1)
    d_e, i_e = capi.knn(X, X, K, self_mode=capi.SELF_EXCLUDE)
    assert "pruned" not in capi.last_kernel() and " seed=" in capi.last_kernel()
    assert np.array_equal(i_p, i_e) and np.array_equal(d_p, d_e)


def test_neighbors_shim_matches_sklearn_semantics(capi):
    from mcevidence_amd.neighbors import NearestNeighbors
    rng = np.random.default_rng(11)
    Y = rng.standard_normal((3000, 6))
    X = rng.standard_normal((500, 6))
    nb = NearestNeighbors(n_neighbors=5, metric="euclidean", leaf_size=20, algorithm="auto", n_jobs=-1).fit(Y)
    d, i = nb.kneighbors(X)
    sd, si = orc.knn_sklearn(X, Y, 5)
    assert _rel(d, sd) < DIST_RTOL and np.array_equal(i, si)
    d, i = nb.kneighbors(Y)                       # the reference's auto-evidence call shape (:1100-1104)
    sd, si = orc.knn_sklearn(Y, Y, 5)
    assert np.all(d[:, 0] == 0) and _rel(d[:, 1:], sd[:, 1:]) < DIST_RTOL and np.array_equal(i, si)
    d, i = nb.kneighbors()                        # sklearn: X=None excludes each point itself
    assert _rel(d, orc.knn_brute(Y, Y, 5, self_mode=2)[0]) < DIST_RTOL


def test_device_pointer_entry_points(capi):
    """resident data path used by bench.py: torch tensors, caller's stream, caller's workspace."""
    import torch
    rng = np.random.default_rng(12)
    n, d, kmax = 30000, 6, 4
    Xh = rng.standard_normal((n, d))
    X = torch.from_numpy(Xh).cuda()
    w = torch.ones(n, dtype=torch.float64, device="cuda")
    fs = torch.zeros(n, dtype=torch.float64, device="cuda")
    K = kmax - 1
    wsb = capi.knn_workspace_bytes(n, n, d, K) + capi.dotp_workspace_bytes(n, kmax)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    out = torch.zeros(kmax, dtype=torch.float64, device="cuda")
    dd = torch.zeros((n, K), dtype=torch.float64, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        capi.knn_dotp_dev(X.data_ptr(), n, X.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(),
                          dd.data_ptr(), ws.data_ptr(), wsb, st.cuda_stream)
    st.synchronize()
    host = capi.knn_dotp(Xh, None, np.ones(n), np.zeros(n), kmax, 1)
    assert np.array_equal(out.cpu().numpy(), host)
    od, _ = orc.knn_brute(Xh[:2000], Xh, K + 1)
    assert _rel(dd.cpu().numpy()[:2000], od[:, 1:]) < DIST_RTOL
    with pytest.raises(ValueError):
        capi.knn_dotp_dev(X.data_ptr(), n, X.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(),
                          0, ws.data_ptr(), 1024, 0)


@pytest.mark.parametrize("pruned", [False, True])
def test_dev_entry_point_under_graph_capture(capi, pruned):
    """the *_dev entry points promise: enqueue only -- no synchronisation, no allocation -- so a call can be
    captured into a HIP graph and replayed on new data in the same buffers (include/mcevidence_hip.h)."""
    import torch
    if pruned and capi.get_search_mode() == capi.MODE_F64:
        pytest.skip("the pruned walk belongs to the fp16-filter path")
    capi.set_prune_mode(capi.PRUNE_FORCE if pruned else capi.PRUNE_OFF)
    try:
        rng = np.random.default_rng(21)
        n, d, kmax = 20000, 5, 4
        K = kmax - 1
        X = torch.empty((n, d), dtype=torch.float64, device="cuda")
        w = torch.ones(n, dtype=torch.float64, device="cuda")
        fs = torch.zeros(n, dtype=torch.float64, device="cuda")
        wsb = capi.knn_workspace_bytes(n, n, d, K) + capi.dotp_workspace_bytes(n, kmax)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        out = torch.zeros(kmax, dtype=torch.float64, device="cuda")
        dd = torch.zeros((n, K), dtype=torch.float64, device="cuda")

        def call(stream):
            capi.knn_dotp_dev(X.data_ptr(), n, X.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(),
                              dd.data_ptr(), ws.data_ptr(), wsb, stream)
        data = [rng.standard_normal((n, d)) for _ in range(3)]
        X.copy_(torch.from_numpy(data[0]))
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):                       # warm-up outside capture (one-time kernel attributes)
            call(side.cuda_stream)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            call(torch.cuda.current_stream().cuda_stream)
        assert "pruned" in capi.last_kernel() if pruned else "pruned" not in capi.last_kernel()
        for h in data[1:]:
            X.copy_(torch.from_numpy(h))
            graph.replay()
            torch.cuda.synchronize()
            want_dotp, want_dist = capi.knn_dotp(h, None, np.ones(n), np.zeros(n), kmax, 1, return_dist=True)
            assert np.array_equal(dd.cpu().numpy(), want_dist)
            assert np.allclose(out.cpu().numpy(), want_dotp, rtol=1e-13, atol=0)
    finally:
        capi.set_prune_mode(capi.PRUNE_AUTO)


def test_random_shapes_against_oracle(capi, monkeypatch):
    """seeded sweep over ragged shapes, all self modes, pruning forced or off, the filter's seed phase forced on
    with random settings in a third of the cases: distances and rows against the exact CPU search."""
    rng = np.random.default_rng(2024)
    try:
        for case in range(48):
            if case % 3 == 1:
                monkeypatch.setenv("MCE_F16_SEED_SHARE", "2")
                monkeypatch.setenv("MCE_F16_SEED_ROWS", str(int(rng.integers(64, 8192))))
                monkeypatch.setenv("MCE_F16_SEED_TG", str(int(rng.integers(1, 17))))
            else:
                for k in ("MCE_F16_SEED_SHARE", "MCE_F16_SEED_ROWS", "MCE_F16_SEED_TG"):
                    monkeypatch.delenv(k, raising=False)
            d = int(rng.choice([1, 2, 3, 5, 6, 8, 13, 14, 27, 40]))
            nr = int(rng.integers(40, 20000))
            K = int(rng.integers(1, min(32, nr - 1) + 1))
            kind = int(rng.integers(0, 3))            # 0 cross, 1 shard with own rows excluded, 2 shard with own rows included
            Y = rng.standard_normal((nr, d)) * rng.uniform(0.1, 30.0) + rng.standard_normal(d) * rng.uniform(0, 50.0)
            if case % 5 == 0:
                Y[rng.integers(0, nr, nr // 3)] = Y[rng.integers(0, nr, nr // 3)]          # exact duplicates
            if kind == 0:
                nq = int(rng.integers(1, 3000))
                X = rng.standard_normal((nq, d)) * 20.0
                sm, off = capi.SELF_NONE, 0
            else:
                nq = int(rng.integers(1, min(nr, 3000) + 1))
                off = int(rng.integers(0, nr - nq + 1))
                X = np.ascontiguousarray(Y[off:off + nq])
                sm = capi.SELF_EXCLUDE if kind == 1 else capi.SELF_INCLUDE
            capi.set_prune_mode(capi.PRUNE_FORCE if case % 2 else capi.PRUNE_OFF)
            dist, idx = capi.knn(X, Y, K, self_mode=sm, self_offset=off)
            od, oi = orc.knn_brute(X, Y, K, self_mode={capi.SELF_NONE: 0, capi.SELF_INCLUDE: 1, capi.SELF_EXCLUDE: 2}[sm], self_offset=off)
            tag = (case, d, nr, nq, K, kind, capi.last_kernel())
            assert np.allclose(dist, od, rtol=DIST_RTOL, atol=0), tag
            assert np.all(np.diff(dist, axis=1) >= 0), tag
            agree = np.mean(idx == oi)
            assert agree > 0.999 or case % 5 == 0, tag              # duplicates: equal distances may swap rows
            # whatever rows were chosen, they are at the reported distances
            pick = rng.integers(0, nq, min(nq, 50))
            true = np.sqrt(((X[pick, None, :] - Y[idx[pick]]) ** 2).sum(-1))
            if sm == capi.SELF_INCLUDE:
                true[:, 0] = np.where(idx[pick, 0] == off + pick, 0.0, true[:, 0])
            assert np.allclose(true, dist[pick], rtol=1e-9, atol=1e-300), tag
    finally:
        capi.set_prune_mode(capi.PRUNE_AUTO)


def test_c_abi_from_plain_c(tmp_path):
    """the boundary is a C ABI, not a Python extension: examples/knn_from_c.c built with gcc against
    include/mcevidence_hip.h and run as its own process."""
    import os
    import subprocess
    from helpers import REPO
    exe = str(tmp_path / "knn_from_c")
    libdir = os.path.join(REPO, "mcevidence_amd")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(REPO, "include"), os.path.join(REPO, "examples", "knn_from_c.c"), "-o", exe,
                           "-L" + libdir, "-lmcevidence_hip", "-Wl,-rpath," + libdir, "-lm"])
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout and "rc -2" in out.stdout


def test_reference_set_beyond_the_26_bit_row_offsets(capi):
    """queue entries hold 26-bit row offsets relative to the workgroup's reference split: a reference set
    with more than 2^26 rows must be split (and is too large for the pruned walk), not silently wrapped."""
    rng = np.random.default_rng(5)
    nr, d, nq, K = (1 << 26) + 5000, 2, 600, 3
    Y = rng.standard_normal((nr, d))
    X = np.concatenate([Y[-300:] + 1e-4, rng.standard_normal((nq - 300, d))])     # half of them next to the LAST rows
    dist, idx = capi.knn(X, Y, K)
    if capi.get_search_mode() != capi.MODE_F64:
        assert "rsplit=1 " not in capi.last_kernel() + " " and "pruned" not in capi.last_kernel(), capi.last_kernel()
    od, oi = orc.knn_brute(X, Y, K)
    assert _rel(dist, od) < DIST_RTOL and np.array_equal(idx, oi)
    assert (idx[:300] >= nr - 400).any()                 # neighbours were found beyond row 2^26


def test_prune_stats_lifetime(prune_modes):
    """mce_last_prune_stats reads counters that live in the search's workspace: available after a *_dev
    call (caller-owned workspace), refused after a host-pointer call (the library's scratch is gone)."""
    import torch
    capi = prune_modes
    capi.set_prune_mode(capi.PRUNE_FORCE)
    rng = np.random.default_rng(3)
    n, d, K = 40000, 4, 5
    Yh = rng.standard_normal((n, d))
    capi.knn(Yh, Yh, K, self_mode=capi.SELF_EXCLUDE)
    assert "pruned" in capi.last_kernel()
    with pytest.raises(ValueError, match="no pruned search"):
        capi.last_prune_stats()
    Y = torch.from_numpy(Yh).cuda()
    wsb = capi.knn_workspace_bytes(n, n, d, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    dist = torch.empty((n, K), dtype=torch.float64, device="cuda")
    capi.knn_dev(Y.data_ptr(), n, Y.data_ptr(), n, d, K, capi.SELF_EXCLUDE, 0, dist.data_ptr(), 0, ws.data_ptr(), wsb, 0)
    windows, tiles = capi.last_prune_stats()
    assert 0.0 < tiles < 0.5 and windows > 0.0           # most tile products were never multiplied


@pytest.mark.parametrize("n,d,kmax,pruned", [(90000, 6, 5, True), (90000, 6, 5, False), (40000, 27, 10, False), (130001, 3, 3, True),
                                             (100000, 6, 17, False)])
def test_partial_sums_add_up(prune_modes, n, d, kmax, pruned):
    """mce_knn_dotp_part_f64: the parts (library-chosen: row ranges for the sweep, k-d cells for the pruned
    walk) are disjoint and complete -- their sums add up to the one-call result -- for any number of parts.
    (100 k x 6, kmax 17: a row shard of that set plans more reference splits -- each with its own 16-entry lists -- than the
    whole set does and would not fit the whole set's workspace: the library then takes fewer splits, capi_common.hpp: t_plan_cap;
    before round 4 parts 4 and 8 of this shape failed with MCE_ERR_WORKSPACE.)"""
    capi = prune_modes
    capi.set_prune_mode(capi.PRUNE_FORCE if pruned else capi.PRUNE_OFF)
    rng = np.random.default_rng(n + kmax)
    Y = rng.standard_normal((n, d))
    w = rng.integers(1, 5, n).astype(float)
    fs = -rng.random(n) * 3
    full = capi.knn_dotp(Y, None, w, fs, kmax, 1)
    for nparts in (1, 2, 3, 8):
        parts = [capi.knn_dotp_part(Y, w, fs, kmax, r, nparts) for r in range(nparts)]
        assert ("pruned" in capi.last_kernel()) == pruned
        assert np.allclose(np.sum(parts, axis=0), full, rtol=1e-12, atol=0)
        assert all((p[1:] > 0).all() and p[0] == 0 for p in parts)
    with pytest.raises(ValueError):
        capi.knn_dotp_part(Y, w, fs, kmax, 3, 3)


def test_feed_route_reports_non_finite_input():
    """evidence_feed skips the host isfinite pass (20-40 ms per 10^6 x 27): the device covariance goes
    non-finite instead and the call raises ValueError, alone or inside a batch."""
    from mcevidence_amd import _capi
    rng = np.random.default_rng(1)
    S = rng.standard_normal((5000, 4))
    w, fs = np.ones(5000), np.zeros(5000)
    for bad in (np.nan, np.inf, -np.inf):
        T = S.copy()
        T[1234, 2] = bad
        with pytest.raises(ValueError, match="NaN or infinity"):
            _capi.evidence_feed(T, None, 4, 0, 3, w, fs)
        out = _capi.evidence_feed_batch([(S, None, 4, 0, 3, w, fs), (T, None, 4, 0, 3, w, fs)], return_exceptions=True)
        assert isinstance(out[1], ValueError) and not isinstance(out[0], Exception)


# --------------------------------------------------------------------------- per-call options
def test_two_threads_with_different_modes_on_one_device():
    """The search / prune / symmetric modes travel with the call (mce_options, ABI 2): two threads share the device, one
    asks for the fp64 sweep, the other for the symmetric filter sweep, many calls each, concurrently.  Every call must
    run the kernel ITS options name (mce_last_kernel() is thread-local) and return the same neighbours; the process-wide
    defaults are untouched."""
    import threading
    from mcevidence_amd import _capi
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((6000, 9))
    want_d, want_i = orc.knn_brute(Y, Y, 5, self_mode=2)
    _capi.set_search_mode(_capi.MODE_AUTO); _capi.set_sym_mode(_capi.SYM_AUTO); _capi.set_prune_mode(_capi.PRUNE_AUTO)
    errors = []

    def worker(opt, must_contain, must_not):
        try:
            for _ in range(12):
                d, i = _capi.knn(Y, Y, 5, self_mode=_capi.SELF_EXCLUDE, options=opt)
                k = _capi.last_kernel()
                assert must_contain in k and (must_not is None or must_not not in k), k
                assert np.array_equal(i, want_i) and _rel(d, want_d) < DIST_RTOL
        except Exception as e:          # noqa: BLE001
            errors.append(repr(e))

    a = threading.Thread(target=worker, args=(_capi.Options(search_mode=_capi.MODE_F64), "knn_mfma_kernel", None))
    b = threading.Thread(target=worker, args=(_capi.Options(sym_mode=_capi.SYM_FORCE, prune_mode=_capi.PRUNE_OFF), "symmetric", "knn_mfma"))
    a.start(); b.start(); a.join(); b.join()
    assert not errors, errors
    assert _capi.get_search_mode() == _capi.MODE_AUTO and _capi.get_sym_mode() == _capi.SYM_AUTO
    # the scoped form, nested
    with _capi.options(search_mode=_capi.MODE_F64):
        _capi.knn(Y, Y, 5, self_mode=_capi.SELF_EXCLUDE)
        assert "knn_mfma_kernel" in _capi.last_kernel()
        with _capi.options(search_mode=_capi.MODE_AUTO, sym_mode=_capi.SYM_FORCE, prune_mode=_capi.PRUNE_OFF):
            _capi.knn(Y, Y, 5, self_mode=_capi.SELF_EXCLUDE)
            assert "symmetric" in _capi.last_kernel()
        _capi.knn(Y, Y, 5, self_mode=_capi.SELF_EXCLUDE)
        assert "knn_mfma_kernel" in _capi.last_kernel()
    _capi.knn(Y, Y, 5, self_mode=_capi.SELF_EXCLUDE)
    assert "knn_f16" in _capi.last_kernel() and "symmetric" not in _capi.last_kernel()
    with pytest.raises(ValueError):
        _capi.knn(Y, Y, 5, options=_capi.Options(search_mode=7))


# --------------------------------------------------------------------------- the matrix-core error model, measured
@pytest.mark.parametrize("kst", [1, 2, 3, 4, 5, 6, 8])       # (5, 6, 8: the deep filter, 64 <= d <= 127 -- knn_deep.hpp)
def test_mfma_error_model(kst):
    """The filter's bound (knn_f16.hpp) assumes: products of two fp16 values are exact, and the fp32 accumulation of the
    16 KST terms of A = |y^|^2 - 2 x^.y^ errs by at most eps_q = 32 KST 2^-24 (|x^| + max |y^|)^2.  Measured here on the
    instruction itself (mce_debug_mfma_tile_f16: the kernels' own MFMA sequence) against the exactly rounded sum, at the
    radius the kernels scale to (200), on the worst case for cancellation -- reference rows next to the query, so that
    |y^|^2 ~ 2 x^.y^ ~ 40 000 while A itself is ~ -|x^|^2 + tiny -- with every sign pattern of the partial sums, and on
    plain random rows.  The observed error must stay within eps_q (it is ~50x smaller)."""
    import math as _m
    from mcevidence_amd import _capi
    rng = np.random.default_rng(1000 + kst)
    D = 16 * kst - 3                                      # three norm pieces fill the last k-step
    worst = 0.0
    for trial in range(24):
        x = rng.standard_normal((32, D))
        x *= 200.0 / np.linalg.norm(x, axis=1, keepdims=True) * (0.5 + 0.5 * rng.random((32, 1)))
        if trial % 3 == 0:      # references within a hair of the queries (row j next to query j, and to its neighbours)
            y = x[rng.permutation(32)] + rng.standard_normal((32, D)) * 10.0 ** rng.uniform(-3, 0.5)
        elif trial % 3 == 1:    # antipodal / sign-flipped copies: the products all have one sign, partial sums peak
            y = -x[rng.permutation(32)] * (1.0 + 0.01 * rng.standard_normal((32, 1)))
        else:
            y = rng.standard_normal((32, D))
            y *= 200.0 / np.linalg.norm(y, axis=1, keepdims=True) * rng.random((32, 1))
        xh = x.astype(np.float16)
        yh = y.astype(np.float16)
        n2 = (yh.astype(np.float64) ** 2).sum(axis=1)     # |y^|^2 from the CONVERTED values, split into three fp16 pieces
        n_hi = n2.astype(np.float16)
        n_mid = (n2 - n_hi.astype(np.float64)).astype(np.float16)
        n_lo = (n2 - n_hi.astype(np.float64) - n_mid.astype(np.float64)).astype(np.float16)
        yp = np.concatenate([(-2.0 * yh.astype(np.float64)).astype(np.float16), n_hi[:, None], n_mid[:, None], n_lo[:, None]], axis=1)
        xp = np.concatenate([xh, np.ones((32, 3), dtype=np.float16)], axis=1)
        assert yp.shape[1] == 16 * kst and np.all(np.isfinite(yp.astype(np.float64)))
        A = _capi.debug_mfma_tile(yp, xp).astype(np.float64)            # [row, query]
        yp64, xp64 = yp.astype(np.float64), xp.astype(np.float64)
        exact = np.array([[_m.fsum(yp64[j] * xp64[i]) for i in range(32)] for j in range(32)])     # products exact in fp64, sum exactly rounded
        xn = np.sqrt((xh.astype(np.float64) ** 2).sum(axis=1))
        ymax = np.sqrt(n2.max())
        eps = 32.0 * kst * 2.0 ** -24 * (xn[None, :] + ymax) ** 2
        err = np.abs(A - exact)
        assert np.all(err <= eps), (kst, trial, float((err / eps).max()))
        worst = max(worst, float((err / eps).max()))
    assert worst < 0.5, worst             # headroom: the model is not tight


@pytest.mark.parametrize("kst", [1, 2, 3, 4, 5, 6, 8])
def test_mfma_error_model_over_thousands_of_tiles(kst):
    """The same model as a DISTRIBUTION: 3 000+ tiles per k-step count (12 000+ in all) through the kernels' MFMA sequence
    in one launch (mce_debug_mfma_tiles_f16) -- near-coincident rows, antipodal rows, random rows, fp16 subnormals next to
    full-size components, rows on the scaling radius with their energy in one to three components, one-signed equal
    components (every partial sum at full magnitude), rows of very different norms in one tile.  Every entry within
    eps_q; the observed maximum stays below half of it (tools/mfma_error_hist.py writes the histogram)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("mfma_error_hist", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "mfma_error_hist.py"))
    eh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(eh)
    rng = np.random.default_rng(77 + kst)
    worst, tiles = {}, 0
    for kind in eh.KINDS:
        yp, xp = eh.make_tiles(kind, kst, 150 if kind == "max_magnitude" else 500, rng)
        r = eh.error_over_bound(yp, xp, kst)
        tiles += len(yp)
        assert np.all(r <= 1.0), (kind, kst, float(r.max()))
        worst[kind] = float(r.max())
    assert tiles >= 3000
    assert max(worst.values()) < 0.5, worst


# --------------------------------------------------------------------------- config C1 on the hardware
def test_C1_file_root_through_the_hip_backend(tmp_path):
    """BASELINE configs[0]: the CosmoMC-format file root (4 chains, 26 862 rows, ndim = 6, kmax = 2) read by libmcechains
    and fed to mce_evidence_feed_f64 -- the whole file -> ln E route on the GPU box -- against the ln E the REFERENCE
    returned for the same files (tests/golden/host_pins.json: C1_all, C1_chain1..4)."""
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import planck_like_chains, write_cosmomc_chains
    from helpers import host_pins
    pins = host_pins()
    chains, names, ranges = planck_like_chains(seed=1)
    root = str(tmp_path / "base_plikHM_TT_lowTEB")
    write_cosmomc_chains(root, chains, ranges)
    pi = pkg.params_info(root, cosmo=True)
    m = pkg.MCEvidence(root, ndim=pi["ndim"], priorvolume=pi["volume"], kmax=2, verbose=0)
    assert m.backend.name == "hip" and m.nsample[0] == pins["C1_all"]["N"] == 26862
    lnE = m.evidence()
    assert np.allclose(lnE, pins["C1_all"]["lnE"], rtol=0, atol=1e-8)          # (text round trip of the chains: a few 1e-10)
    for ic in (1, 2, 3, 4):
        mi = pkg.MCEvidence(root, ndim=6, priorvolume=pi["volume"], kmax=2, verbose=0, idchain=ic)
        assert np.allclose(mi.evidence(), pins["C1_chain%d" % ic]["lnE"], rtol=0, atol=1e-8)


# --------------------------------------------------------------------------- full-size reductions
def test_cross_evidence_reduction_at_full_size_C4():
    """BASELINE configs[3]: the FUSED search + volume/weight sum of cross evidence (k0 = 0: all kmax columns count,
    MCEvidence.py:1093-1096, 1120-1122) at 1M x 1M x 15, integer weights and a non-trivial likelihood column.  The sum is
    linear in the rows, so it is checked in two steps that together cover it: the fused sums over ALL rows equal the
    literal sum formed on the host from the distances the same call returns, and on 30 000 sampled rows those distances
    equal the exact CPU search's (the search over all 10^6 rows takes the oracle 14 minutes on 256 threads: done once,
    round 3, same result)."""
    from mcevidence_amd import _capi
    from mcevidence_amd.synth import config_chain
    chain, (r1, r2) = config_chain("C4")
    theta = chain[:, 2:]
    ev, U = np.linalg.eigh(np.cov(theta.T))
    W = (theta @ U) / np.sqrt(ev)
    X, Y = np.ascontiguousarray(W[r1]), np.ascontiguousarray(W[r2])
    rng = np.random.default_rng(4)
    w = rng.integers(1, 6, len(X)).astype(np.float64)
    fs = -chain[r1, 1] - np.max(-chain[r1, 1])
    kmax = 4
    dotp, dist = _capi.knn_dotp(X, Y, w, fs, kmax, 0, return_dist=True)
    assert "knn_f16" in _capi.last_kernel()
    assert np.allclose(dotp, orc.dotp_literal(dist, w, fs, X.shape[1], 0, kmax), rtol=1e-11, atol=0)
    rows = np.sort(rng.choice(len(X), 30_000, replace=False))
    od, _ = orc.knn_brute(X[rows], Y, kmax)
    assert np.allclose(dist[rows], od, rtol=DIST_RTOL, atol=0)
    assert np.allclose(orc.dotp_literal(dist[rows], w[rows], fs[rows], X.shape[1], 0, kmax), orc.dotp_literal(od, w[rows], fs[rows], X.shape[1], 0, kmax), rtol=1e-11, atol=0)


def test_auto_evidence_reduction_at_full_size_C5():
    """BASELINE configs[4]: 10M x 6, K = 9 through the automatic path (k-d pruned walk).  The fused sums over ALL rows
    equal the sums formed on the host from the distances the same call returns; on 6000 sampled rows those distances
    -- hence those rows' partial sums -- are checked against the exact CPU search (200 000 rows, 18 minutes of the
    oracle: done once, round 3, same result)."""
    from mcevidence_amd import _capi
    _capi.set_prune_mode(_capi.PRUNE_AUTO); _capi.set_sym_mode(_capi.SYM_AUTO); _capi.set_search_mode(_capi.MODE_AUTO)
    rng = np.random.default_rng(56)
    n, d, kmax = 10_000_000, 6, 10
    X = rng.standard_normal((n, d)) @ (np.eye(d) + 0.3 * rng.standard_normal((d, d)))
    ev, U = np.linalg.eigh(np.cov(X.T))
    X = np.ascontiguousarray((X @ U) / np.sqrt(ev))
    w = rng.integers(1, 6, n).astype(np.float64)
    fs = -0.5 * (X ** 2).sum(axis=1)
    fs -= fs.max()
    dotp, dist = _capi.knn_dotp(X, None, w, fs, kmax, 1, return_dist=True)
    assert "pruned" in _capi.last_kernel()
    full = np.zeros((n, kmax)); full[:, 1:] = dist
    assert np.allclose(dotp[1:], orc.dotp_literal(full, w, fs, d, 1, kmax)[1:], rtol=1e-11, atol=0)
    rows = np.sort(rng.choice(n, 6000, replace=False))
    od, _ = orc.knn_brute(X[rows], X, kmax)                # column 0: the row itself
    assert np.allclose(dist[rows], od[:, 1:], rtol=DIST_RTOL, atol=0)
    fo = np.zeros((len(rows), kmax)); fo[:, 1:] = od[:, 1:]
    assert np.allclose(orc.dotp_literal(full[rows], w[rows], fs[rows], d, 1, kmax)[1:], orc.dotp_literal(fo, w[rows], fs[rows], d, 1, kmax)[1:], rtol=1e-11, atol=0)


# --------------------------------------------------------------------------- multi-rank evidence() on the device-feeder route
@pytest.mark.parametrize("n,d,kmax,cross", [(60000, 6, 5, False), (150000, 27, 10, False), (310000, 6, 4, False), (50000, 15, 4, True)])
def test_feed_parts_add_up_to_the_single_rank_feed(n, d, kmax, cross):
    """mce_evidence_feed_part_f64: one upload, device covariance / whitening, this rank's share of the search.  The shares
    of 1, 2, 3 and 8 ranks (all on this GPU, one after the other) add up to mce_evidence_feed_f64's sums; Jacobian and
    eigenvalues are the single-rank ones on every rank; the device-side checksum is the same on every rank and moves with
    one bit of one sample, weight or likelihood."""
    from mcevidence_amd import _capi
    from mcevidence_amd.synth import gaussian_chain
    ch = gaussian_chain(seed=n + d, n=n, d=d, weights="int", cov="corr")
    S1, w = ch[:, 2:], ch[:, 0]
    fs = -ch[:, 1] - np.max(-ch[:, 1])
    S2 = gaussian_chain(seed=n + d + 1, n=n - 1234, d=d, cov="corr")[:, 2:] if cross else None
    full, jac, ev = _capi.evidence_feed(S1, S2, d, 0, kmax, w, fs)
    k0 = 0 if cross else 1
    sums = set()
    for nparts in (1, 2, 3, 8):
        tot = np.zeros(kmax)
        for r in range(nparts):
            part, j, e, cs = _capi.evidence_feed_part(S1, S2, d, 0, kmax, w, fs, r, nparts)
            assert j == jac and np.array_equal(e, ev)
            tot += part
            sums.add(cs)
        assert np.allclose(tot[k0:], full[k0:], rtol=1e-12, atol=0), nparts
    assert len(sums) == 1
    base = sums.pop()
    # round 6: the same fingerprint from the HOST copy of the inputs (libmcechains: mce_chain_fingerprint_f64) -- what the ranks
    # compare when the node uploads the chain once -- and the same sums from inputs that are on the device already
    # (mce_evidence_feed_part_dev_f64: strided rows, s2 right behind s1)
    import torch
    from mcevidence_amd import chain_io
    assert chain_io.feed_fingerprint(S1, S2, d, w, fs) == base
    both = np.concatenate((S1[:, :d], S2[:, :d])) if cross else np.ascontiguousarray(S1[:, :d])
    wide = torch.zeros((both.shape[0], d + 3), dtype=torch.float64, device="cuda")
    wide[:, :d] = torch.from_numpy(both).cuda()
    wd, fd = torch.from_numpy(np.ascontiguousarray(w)).cuda(), torch.from_numpy(np.ascontiguousarray(fs)).cuda()
    torch.cuda.synchronize()
    tot = np.zeros(kmax)
    for r in range(3):
        part, j, e, cs = _capi.evidence_feed_part_dev(wide.data_ptr(), len(S1), d + 3, wide[len(S1):].data_ptr() if cross else 0, len(S2) if cross else 0, d + 3,
                                                      d, 0, kmax, wd.data_ptr(), fd.data_ptr(), r, 3)
        assert j == jac and np.array_equal(e, ev) and cs == base
        tot += part
    assert np.allclose(tot[k0:], full[k0:], rtol=1e-12, atol=0)
    del wide, wd, fd
    for which in range(3):
        A, ww, ff = S1.copy(), w.copy(), fs.copy()
        tgt = (A, ww, ff)[which]
        tgt[(4321,) + ((d - 1,) if which == 0 else ())] = np.nextafter(tgt[(4321,) + ((d - 1,) if which == 0 else ())], 1e9)
        assert _capi.evidence_feed_part(A, S2, d, 0, kmax, ww, ff, 0, 2)[3] != base
    assert _capi.evidence_feed_part(S1, S2, d, 0, kmax, w, fs, 0, 2, want_checksum=False)[3] is None
    with pytest.raises(ValueError):
        _capi.evidence_feed_part(S1, S2, d, 0, kmax, w, fs, 2, 2)


def _class_rank(rank, world, port, q, cross, poison, pairs_once=False, node_upload=False):
    import os
    import sys
    import time
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if pairs_once:
        os.environ["MCE_PAIRS_ONCE"] = "1"
    os.environ["MCE_NODE_UPLOAD"] = "1" if node_upload else "0"      # (gloo: the gather goes through the host -- functional check)
    torch.cuda.set_device(0)                                 # one GPU on the test box: both ranks share it
    dist.init_process_group("gloo", rank=rank, world_size=world)
    logging.disable(logging.CRITICAL)
    import mcevidence_amd as pkg
    from mcevidence_amd import _capi
    from mcevidence_amd.synth import config_chain, gaussian_chain
    if cross:
        chain, (r1, r2) = config_chain("C4", n=120000)
    else:
        chain, r1 = gaussian_chain(seed=3, n=300000, d=27, cov="corr"), None
    if poison == "fail" and rank == 1:
        # this rank's preparation fails (here: the whitening call is handed nonsense): it must still join the first collective
        boom = lambda *a, **k: (_ for _ in ()).throw(MemoryError("boom on rank 1"))
        _capi.evidence_feed_whiten = boom
        _capi.evidence_feed_whiten_dev = boom
    elif poison and rank == 1:
        chain = chain.copy()
        chain[1000, 5] = np.nextafter(chain[1000, 5], 1e9)
    m = pkg.MCEvidence([chain], kmax=4 if cross else 10, verbose=0)
    if cross:
        m.set_split(r1, r2)
    try:
        m.evidence()
        t0 = time.perf_counter()
        lnE = m.evidence()
        out = ("ok", lnE, time.perf_counter() - t0, _capi.last_kernel())
    except (RuntimeError, MemoryError) as e:
        out = ("raised", str(e), 0.0, "")
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("poison,node_upload", [(False, False), (True, False), ("fail", False), (False, True), (True, True), ("fail", True)])
def test_class_under_two_ranks_through_the_pairs_once_partition(poison, node_upload):
    """MCE_PAIRS_ONCE=1: MCEvidence(...).evidence() under a 2-rank gloo group (both ranks on the box's one GPU) whitens on the
    device (mce_evidence_feed_whiten_f64: the rows stay there) and searches through the all-pairs-once partition -- bounds
    all-reduced, candidates exchanged, sums and input fingerprints in the last all-reduce.  ln E equals the single process's to
    1e-12; a rank with one different bit makes both raise; a rank whose preparation fails raises its own error and the other
    one a RuntimeError, in the FIRST collective."""
    import socket
    import torch.multiprocessing as mp
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    if not poison:
        one = pkg.MCEvidence([gaussian_chain(seed=3, n=300000, d=27, cov="corr")], kmax=10, verbose=0).evidence()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_class_rank, args=(r, 2, port, q, False, poison, True, node_upload)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if poison == "fail":
        assert got[0][0] == got[1][0] == "raised" and "boom on rank 1" in got[1][1] and "another rank" in got[0][1], got
        return
    if poison:
        assert got[0][0] == got[1][0] == "raised" and "different samples" in got[0][1]
        return
    assert got[0][0] == got[1][0] == "ok"
    assert np.array_equal(got[0][1], got[1][1])
    assert np.max(np.abs(got[0][1] - one)) < 1e-12, (got[0][1], one)
    assert "pairs-once" in got[0][3] and "pairs-once" in got[1][3]


@pytest.mark.parametrize("cross,poison,node_upload", [(False, False, False), (True, False, False), (False, True, False),
                                                      (False, False, True), (True, False, True), (False, True, True), (True, True, True)])
def test_class_under_two_ranks_equals_the_single_rank_lnE(cross, poison, node_upload):
    """MCEvidence(...).evidence() under a 2-rank gloo group on the GPU box (both ranks on its one GPU): each rank uploads the
    chain ONCE, whitens on the device and searches its share (mce_evidence_feed_part_f64), one all-reduce -- no host
    covariance, no host hashing.  ln E equals the single-process result to 1e-12; a rank with one different bit makes
    BOTH ranks raise.  node_upload (round 6): ONE upload per node -- every rank uploads half of the rows, an all_gather hands
    both the whole set (mce_evidence_feed_part_dev_f64), the inputs are compared through host fingerprints: same ln E, and
    the flipped bit is still caught (in s1 and, for cross evidence, wherever the row lands)."""
    import socket
    import time
    import torch.multiprocessing as mp
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import config_chain, gaussian_chain
    if not poison:
        if cross:
            chain, (r1, r2) = config_chain("C4", n=120000)
            m = pkg.MCEvidence([chain], kmax=4, verbose=0).set_split(r1, r2)
        else:
            m = pkg.MCEvidence([gaussian_chain(seed=3, n=300000, d=27, cov="corr")], kmax=10, verbose=0)
        m.evidence()
        t0 = time.perf_counter()
        one = m.evidence()
        t_one = time.perf_counter() - t0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_class_rank, args=(r, 2, port, q, cross, poison, False, node_upload)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if poison:
        assert got[0][0] == got[1][0] == "raised" and "different samples" in got[0][1]
        return
    assert got[0][0] == got[1][0] == "ok"
    assert np.array_equal(got[0][1], got[1][1])
    assert np.max(np.abs(got[0][1] - one)) < 1e-12, (got[0][1], one)
    # both ranks share ONE GPU here, so a call cannot be faster than the single-process one; it must not be the old
    # host detour either (np.cov + eig + whitening + BLAKE2b of the whole set on every rank: ~10x the device call)
    # (node_upload over gloo: the gather goes through the host and TCP -- a functional route there, only bounded loosely)
    assert max(got[0][2], got[1][2]) < (1.0 if node_upload else 4.0 * t_one + 0.05), (got[0][2], got[1][2], t_one)
    if not cross:
        assert "symmetric" in got[0][3]          # 300 k x 27 over two ranks: the symmetric partition


# --------------------------------------------------------------------------- the deep fp16 filter (64 <= d <= 127; knn_deep.hpp)
@pytest.fixture()
def deep_capi():
    """the library in the DEFAULT search mode for the duration of a test, whatever the module-scoped `capi` fixture's current
    parameter has set process-wide (per-call options: thread-scoped, they win over the process default)"""
    from mcevidence_amd import _capi
    _capi.require_device()
    with _capi.options(search_mode=_capi.MODE_AUTO):
        yield _capi


def _deep_data(n, d, seed, corr=True):
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((n, d))
    if corr:
        Y = Y @ (np.eye(d) + 0.2 * rng.standard_normal((d, d)) / np.sqrt(d)) + 3.0 * rng.standard_normal((1, d))
    return Y


@pytest.mark.parametrize("d,K", [(64, 9), (72, 1), (80, 4), (100, 12), (127, 16), (64, 17), (90, 24), (127, 32)])      # (K > 16: two passes of 16-entry lists)
@pytest.mark.parametrize("variant", ["default", "redo", "noseed", "split3"])
def test_deep_filter_is_exact(variant, d, K, monkeypatch, deep_capi):
    """The fp16 filter at five to eight k-steps: distances AND rows of the exact CPU search, for every self mode, separate query
    sets and ragged sizes -- with the seed phase on and off, every candidate forced through the redo list (queue-overflow path),
    and three reference splits (several lists per query, merged)."""
    _capi = deep_capi
    if variant == "redo":
        monkeypatch.setenv("MCE_PANEL_DEBUG", "8")
    elif variant == "noseed":
        monkeypatch.setenv("MCE_F16_SEED_ROWS", "0")
    elif variant == "split3":
        monkeypatch.setenv("MCE_RSPLIT", "3")
    n = 5000 if variant == "default" else 2600
    Y = _deep_data(n + 37, d, 11 * d + K)
    X = _deep_data(700 + 13, d, 5 * d + K)
    dist, idx = _capi.knn(X, Y, K)
    assert "knn_deep_kernel" in _capi.last_kernel() and ("two passes" in _capi.last_kernel()) == (K > 16), _capi.last_kernel()
    od, oi = orc.knn_brute(X, Y, K)
    assert _rel(dist, od) < 1e-13 and np.array_equal(idx, oi)
    d2, i2 = _capi.knn(Y, Y, K, self_mode=_capi.SELF_EXCLUDE)
    od, oi = orc.knn_brute(Y, Y, K, self_mode=2)
    assert _rel(d2, od) < 1e-13 and np.array_equal(i2, oi)
    if K > 1:
        d1, i1 = _capi.knn(Y, Y, K, self_mode=_capi.SELF_INCLUDE)
        assert np.all(d1[:, 0] == 0) and np.array_equal(d1[:, 1:], d2[:, :K - 1]) and np.array_equal(i1[:, 1:], i2[:, :K - 1])
    lo = 1000                                              # a query shard of the set: self_offset = its first row
    d3, i3 = _capi.knn(Y[lo:lo + 600], Y, K, self_mode=_capi.SELF_EXCLUDE, self_offset=lo)
    assert np.array_equal(d3, d2[lo:lo + 600]) and np.array_equal(i3, i2[lo:lo + 600])
    assert _capi.verify_knn(Y, Y, d2, self_mode=_capi.SELF_EXCLUDE, nsample=len(Y)) == 0


@pytest.mark.parametrize("kind", sorted(ADVERSARIAL))
def test_deep_filter_adversarial_inputs_stay_exact(kind, deep_capi):
    """the adversarial families of test_knn_adversarial_inputs_stay_exact at d = 100 through the deep filter: whatever the data
    look like, it never drops a true neighbour"""
    import zlib
    _capi = deep_capi
    d, n, K = 100, 3000, 6
    rng = np.random.default_rng(zlib.crc32(("deep-%s" % kind).encode()))
    Y = np.ascontiguousarray(ADVERSARIAL[kind](rng, n, d), dtype=np.float64)
    dist, idx = _capi.knn(Y, Y, K, self_mode=_capi.SELF_EXCLUDE)
    assert "knn_deep_kernel" in _capi.last_kernel(), _capi.last_kernel()
    od, oi = orc.knn_brute(Y, Y, K, self_mode=2)
    exact = np.sqrt(((Y[:, None, :] - Y[idx]) ** 2).sum(-1))
    assert np.allclose(dist, exact, rtol=1e-13, atol=0)
    assert np.array_equal(dist, od) or _rel(dist, od) < 1e-13
    if kind not in ("all_identical", "few_distinct"):
        assert np.mean(np.sort(idx, axis=1) == np.sort(oi, axis=1)) > 0.999          # identical sets except exact distance ties
    assert _capi.verify_knn(Y, Y, dist, self_mode=_capi.SELF_EXCLUDE, nsample=n) == 0


def test_deep_filter_behind_the_class_and_the_fused_sums(deep_capi):
    """MCEvidence(...).evidence() at ndim = 80 (a chain taken with its derived columns): device feeders + the deep filter, ln E
    equal to the oracle's; the fused sums equal the literal sum of the returned distances; search mode 1 (fp64 sweep) agrees."""
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    _capi = deep_capi
    chain = gaussian_chain(seed=8, n=9000, d=80, weights="int", cov="corr")
    lnE = pkg.MCEvidence([chain], kmax=5, verbose=0).evidence()
    assert "knn_deep_kernel<KST=6" in _capi.last_kernel(), _capi.last_kernel()
    ref = orc.evidence_from_chain(chain, kmax=5, knn="brute")
    assert np.max(np.abs(lnE - ref["lnE"])) < LNE_TOL
    with _capi.options(search_mode=_capi.MODE_F64):
        lnE64 = pkg.MCEvidence([chain], kmax=5, verbose=0).evidence()
        assert "knn_mfma_kernel" in _capi.last_kernel()
    assert np.max(np.abs(lnE - lnE64)) < 1e-11
    X = ref["X"]
    w, fs = chain[:, 0], -chain[:, 1] - np.max(-chain[:, 1])
    dp, dd = _capi.knn_dotp(X, None, w, fs, 5, 1, return_dist=True)
    full = np.zeros((len(X), 5)); full[:, 1:] = dd
    assert np.allclose(dp[1:], orc.dotp_logdomain(full, w, fs, 80, 1, 5)[1:], rtol=1e-11)


# --------------------------------------------------------------------------- distributed k-d preparation of the pruned walk (round 6)
@pytest.mark.parametrize("n,d,kmax", [(150000, 6, 10), (70001, 3, 5), (260000, 4, 4)])
def test_distributed_kd_preparation_is_the_single_gpu_order(prune_modes, n, d, kmax):
    """every rank of a pruned multi-GPU search repeated the whole k-d preparation.  Now rank r of W = 2, 4, 8
    sorts only ITS subtree below the top log2 W levels (mce_prune_part_prepare_dev); the ranks' permutation arrays -- final inside the
    own range, zeros outside -- ADD UP to the single-GPU permutation bit for bit, and every rank's share of the search on that shared
    order (mce_knn_dotp_part_prepared_f64_dev) equals its share with the replicated preparation bit for bit; the shares add up to the
    one-GPU sums.  The W ranks run one after the other on this box's GPU."""
    import torch
    capi = prune_modes
    capi.set_prune_mode(capi.PRUNE_FORCE)
    with capi.options(search_mode=capi.MODE_AUTO):
        rng = np.random.default_rng(n + d)
        Y = rng.standard_normal((n, d)) @ (np.eye(d) + 0.3 * rng.standard_normal((d, d)))
        Yd = torch.from_numpy(Y).cuda()
        w = torch.from_numpy(rng.integers(1, 4, n).astype(np.float64)).cuda()
        fs = torch.from_numpy(-rng.random(n)).cuda()
        wsb = capi.knn_workspace_bytes(n, n, d, kmax - 1) + capi.dotp_workspace_bytes(n, kmax)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        one = torch.zeros(kmax, dtype=torch.float64, device="cuda")
        capi.knn_dotp_dev(Yd.data_ptr(), n, Yd.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), one.data_ptr(), 0, ws.data_ptr(), wsb, 0)
        torch.cuda.synchronize()
        assert "pruned" in capi.last_kernel()
        assert not capi.prune_part_applies(n, d, kmax, 3) and not capi.prune_part_applies(n, d, kmax, 1)        # powers of two only
        for W in (2, 4, 8):
            assert capi.prune_part_applies(n, d, kmax, W)
            # the single-GPU permutation: what the plain call above left in the workspace (same plan, same offset)
            perms, segs = [], []
            for r in range(W):
                off, cnt, lo, hi = capi.prune_part_prepare_dev(Yd.data_ptr(), n, d, kmax, r, W, ws.data_ptr(), wsb, 0, want_range=True)
                torch.cuda.synchronize()
                assert cnt > 0 and cnt % 2048 == 0 and lo % 2048 == 0 and 0 <= lo < hi <= cnt
                pr = ws[off:off + 4 * cnt].view(torch.int32).clone()
                assert int(torch.count_nonzero(pr[:lo])) == 0 and int(torch.count_nonzero(pr[hi:])) == 0        # zeros outside the own range
                segs.append((lo, hi))
                perms.append(pr)
            total = torch.stack(perms).sum(dim=0)
            assert segs[0][0] == 0 and segs[-1][1] == cnt and all(a[1] == b[0] for a, b in zip(segs[:-1], segs[1:]))      # rank order, tiling the array
            # reference order: the replicated preparation
            ref = torch.zeros(kmax, dtype=torch.float64, device="cuda")
            capi.knn_dotp_part_dev(Yd.data_ptr(), n, d, kmax, 0, W, w.data_ptr(), fs.data_ptr(), ref.data_ptr(), ws.data_ptr(), wsb, 0)
            torch.cuda.synchronize()
            single = ws[off:off + 4 * cnt].view(torch.int32).clone()
            assert torch.equal(total, single)                        # bit for bit the single-GPU order
            assert int((single >= 0).sum()) == n and set(single[single >= 0].cpu().numpy().tolist()) == set(range(n))
            acc = np.zeros(kmax)
            for r in range(W):
                rep = torch.zeros(kmax, dtype=torch.float64, device="cuda")
                capi.knn_dotp_part_dev(Yd.data_ptr(), n, d, kmax, r, W, w.data_ptr(), fs.data_ptr(), rep.data_ptr(), ws.data_ptr(), wsb, 0)
                torch.cuda.synchronize()
                ws[off:off + 4 * cnt].view(torch.int32).copy_(total)            # what the all-reduce hands every rank
                got = torch.zeros(kmax, dtype=torch.float64, device="cuda")
                capi.knn_dotp_part_prepared_dev(Yd.data_ptr(), n, d, kmax, r, W, w.data_ptr(), fs.data_ptr(), got.data_ptr(), ws.data_ptr(), wsb, 0)
                torch.cuda.synchronize()
                assert torch.equal(got, rep), (W, r)
                acc += got.cpu().numpy()
            assert np.allclose(acc[1:], one.cpu().numpy()[1:], rtol=1e-12, atol=0), W


def _pruned_class_rank(rank, world, port, q, poison):
    import os
    import time
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MCE_NODE_UPLOAD="1")
    torch.cuda.set_device(0)                                 # one GPU on the test box: both ranks share it
    dist.init_process_group("gloo", rank=rank, world_size=world)
    logging.disable(logging.CRITICAL)
    import mcevidence_amd as pkg
    from mcevidence_amd import _capi
    from mcevidence_amd.synth import gaussian_chain
    chain = gaussian_chain(seed=6, n=400000, d=6, cov="corr")
    if poison == "fail" and rank == 1:
        boom = lambda *a, **k: (_ for _ in ()).throw(MemoryError("boom on rank 1"))
        _capi.evidence_feed_whiten = boom
        _capi.evidence_feed_whiten_dev = boom
    elif poison and rank == 1:
        chain = chain.copy()
        chain[1000, 5] = np.nextafter(chain[1000, 5], 1e9)
    m = pkg.MCEvidence([chain], kmax=10, verbose=0)
    try:
        out = ("ok", m.evidence(), _capi.last_kernel())
    except (RuntimeError, MemoryError) as e:
        out = ("raised", str(e), "")
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("poison", [False, True, "fail"])
def test_class_under_two_ranks_with_the_distributed_kd_preparation(poison):
    """MCEvidence(...).evidence() of a chain that takes the pruned walk (400 k x 6) under a 2-rank gloo group sharing this box's GPU:
    one upload per node, whitening on the device, each rank's half of the k-d sorts, the all-reduce of the permutation, the search,
    the all-reduce of the sums (parallel.pruned_part_feed).  ln E equals the single process's to 1e-12; a rank with one different bit
    makes both raise; a rank whose whitening fails raises its own error and the other one a RuntimeError -- nobody is left in the
    permutation's all-reduce."""
    import socket
    import torch.multiprocessing as mp
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    if not poison:
        one = pkg.MCEvidence([gaussian_chain(seed=6, n=400000, d=6, cov="corr")], kmax=10, verbose=0).evidence()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pruned_class_rank, args=(r, 2, port, q, poison)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if poison == "fail":
        assert got[0][0] == got[1][0] == "raised" and "boom on rank 1" in got[1][1] and "1 of the 2 ranks" in got[0][1], got
        return
    if poison:
        assert got[0][0] == got[1][0] == "raised" and "different samples" in got[0][1]
        return
    assert got[0][0] == got[1][0] == "ok" and np.array_equal(got[0][1], got[1][1])
    assert np.max(np.abs(got[0][1] - one)) < 1e-12, (got[0][1], one)
    assert "pruned" in got[0][2] and "pruned" in got[1][2]
