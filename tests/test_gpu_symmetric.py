"""Symmetric sweep of auto-evidence searches (knn_f16.hpp, "Symmetric sweep": every pair of rows multiplied
once, gated for both of its sides) against the exhaustive sweep and the CPU oracle: same distances, same
neighbour rows, same order -- bit for bit.  Needs a real MI355X: run with -m gpu."""
import logging

import numpy as np
import pytest

from helpers import DIST_RTOL, LNE_TOL, chain_of, load_golden, orc

pytestmark = pytest.mark.gpu
logging.disable(logging.CRITICAL)
G = load_golden()


@pytest.fixture()
def sym():
    from mcevidence_amd import _capi
    assert _capi.device_count() >= 1, "no GPU visible: the HIP path cannot be tested"
    _capi.set_search_mode(_capi.MODE_AUTO)
    _capi.set_prune_mode(_capi.PRUNE_OFF)          # the pruned walk has priority where it applies
    yield _capi
    _capi.set_sym_mode(_capi.SYM_AUTO)
    _capi.set_prune_mode(_capi.PRUNE_AUTO)


def _both(capi, fn):
    capi.set_sym_mode(capi.SYM_OFF)
    a = fn()
    ka = capi.last_kernel()
    capi.set_sym_mode(capi.SYM_FORCE)
    b = fn()
    kb = capi.last_kernel()
    assert "symmetric" in kb and "symmetric" not in ka, (ka, kb)
    return a, b


def _data(n, d, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, d)) @ (np.eye(d) + 0.3 * rng.standard_normal((d, d))) + 3.0


@pytest.mark.parametrize("n,d,K", [(513, 4, 3), (700, 27, 9), (1024, 2, 1), (1025, 3, 2), (2048, 6, 4), (5000, 1, 3), (7777, 27, 9), (20000, 6, 10), (33333, 15, 16), (25000, 16, 4),
                                   (12345, 31, 12), (9000, 40, 5), (6000, 63, 8), (70001, 27, 10), (150000, 6, 4)])
def test_symmetric_sweep_is_bit_identical(sym, n, d, K):
    """all three self modes; sizes that are not whole blocks, whole chunks or even tile counts; every k-step count"""
    capi = sym
    Y = _data(n, d, n + d)
    for sm in (capi.SELF_EXCLUDE, capi.SELF_INCLUDE, capi.SELF_NONE):
        (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, K, self_mode=sm))
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1), (sm, np.argwhere(d0 != d1)[:5], np.argwhere(i0 != i1)[:5])
    m = min(n, 2000)
    od, oi = orc.knn_brute(Y[-m:], Y, K, self_mode=2, self_offset=n - m)
    g, gi = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
    # continuous data, no ties: the exact-distance lists select EXACTLY the oracle's rows (filter mode is rigorous)
    assert _rel(g[-m:], od) < DIST_RTOL and np.array_equal(gi[-m:], oi)


@pytest.mark.parametrize("n,d,K", [(700, 27, 17), (5000, 6, 24), (20000, 27, 24), (33333, 15, 32), (12345, 31, 20), (9000, 45, 19), (70001, 27, 21), (150000, 6, 17)])
def test_symmetric_two_pass_search_is_bit_identical(sym, n, d, K):
    """Round 5: 16 < K <= 32 on the symmetric sweep -- two symmetric passes over 16-entry lists (the second, knn_panel.hpp LOWER,
    keeps what lies beyond every row's 16th neighbour; its prepass bounds the K-th distance with 33 group minima) -- against
    the exhaustive two-pass search: same distances, same rows, same order, all self modes; and against the oracle."""
    capi = sym
    Y = _data(n, d, n + d + K)
    for sm in (capi.SELF_EXCLUDE, capi.SELF_INCLUDE, capi.SELF_NONE):
        (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, K, self_mode=sm))
        assert "two passes" in capi.last_kernel(), capi.last_kernel()
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1), (sm, np.argwhere(d0 != d1)[:5], np.argwhere(i0 != i1)[:5])
    m = min(n, 1500)
    od, oi = orc.knn_brute(Y[-m:], Y, K, self_mode=2, self_offset=n - m)
    g, gi = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
    assert _rel(g[-m:], od) < DIST_RTOL and np.array_equal(gi[-m:], oi)
    # the fused reduction over two lists per row
    rng = np.random.default_rng(K)
    w = rng.integers(1, 4, n).astype(float)
    fs = -rng.random(n)
    a, b = _both(capi, lambda: capi.knn_dotp(Y, None, w, fs, K + 1, 1))
    assert np.allclose(a, b, rtol=1e-12, atol=0.0)          # (the same terms, summed in sorted-row order instead of the caller's)


def test_symmetric_two_pass_ties_overflow_and_give_up(sym, monkeypatch):
    """the second pass under the conditions that break things: exact ties across the cut (a lattice: many rows at the 16th
    distance -- the cut is lexicographic in (distance, caller row)), duplicates, buckets far too small (the LOWER repair
    launch) and units that give up waiting for their block's previous unit"""
    capi = sym
    grid = np.stack(np.meshgrid(*[np.arange(10.0)] * 4), -1).reshape(-1, 4)     # 10 000 lattice points
    for K in (17, 24, 32):
        (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(grid, grid, K, self_mode=capi.SELF_EXCLUDE))
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1), K
    rng = np.random.default_rng(4)
    base = rng.standard_normal((2500, 5))
    Y = np.ascontiguousarray(np.concatenate([base, base, base[:900], rng.standard_normal((3000, 5))])[rng.permutation(8900)])
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, 20, self_mode=capi.SELF_EXCLUDE))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    Z = _data(30000, 12, 5)
    want = capi.knn(Z, Z, 22, self_mode=capi.SELF_EXCLUDE)
    capi.set_sym_mode(capi.SYM_FORCE)
    for name, val in (("MCE_SYM_BUCKET", "2"), ("MCE_SYM_SPIN_LIMIT", "0"), ("MCE_PANEL_DEBUG", "16"), ("MCE_PANEL_DEBUG", "8")):
        monkeypatch.setenv(name, val)
        if name == "MCE_SYM_SPIN_LIMIT":
            monkeypatch.setenv("MCE_SYM_PANEL", "4")            # many units per block
        got = capi.knn(Z, Z, 22, self_mode=capi.SELF_EXCLUDE)
        assert "symmetric" in capi.last_kernel() and "two passes" in capi.last_kernel()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), name
        monkeypatch.delenv(name)


def test_symmetric_two_pass_reproduces_the_reference_at_kmax_21(sym):
    """the reference's own ln E for k = 1 .. 20 (kmax = 21: tests/golden/evidence_sym2.json, 140 k x 20) through the class; the
    automatic mode takes the symmetric sweep in two passes at this size"""
    from helpers import build_mce
    from mcevidence_amd import _capi
    case = G["auto_n140000_d20_k21_corr"]
    _capi.set_sym_mode(_capi.SYM_AUTO)
    lnE = build_mce(case).evidence(**case["ev"])
    k = _capi.last_kernel()
    assert "symmetric" in k and "two passes" in k, k
    assert lnE.shape == (20,) and np.max(np.abs(lnE - np.array(case["lnE"]))) < LNE_TOL


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300) * (b != 0)) if a.size else 0.0


def test_symmetric_sweep_duplicates_clusters_and_ties(sym):
    """exact ties break on the caller's row numbers although the rows are searched in sorted order and candidates
    reach a row from two sides; tight far-apart clusters; a constant column; a lattice (massive ties at the K-th distance)"""
    capi = sym
    rng = np.random.default_rng(9)
    base = rng.standard_normal((3000, 4))
    Y = np.concatenate([base, base, base[:1500], rng.standard_normal((2000, 4)) * 1e-3 + 50.0, base[::-1]])
    Y[:, 3] = 1.25
    Y = np.ascontiguousarray(Y[rng.permutation(len(Y))])
    for K in (1, 4, 9):
        (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE))
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    od, oi = orc.knn_brute(Y[:1500], Y, 4, self_mode=2)
    g, gi = capi.knn(Y, Y, 4, self_mode=capi.SELF_EXCLUDE)
    assert np.array_equal(g[:1500], od) and np.array_equal(gi[:1500], oi)
    grid = np.stack(np.meshgrid(*[np.arange(12.0)] * 4), -1).reshape(-1, 4)     # 20736 lattice points
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(grid, grid, 9, self_mode=capi.SELF_EXCLUDE))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    same = np.tile(rng.standard_normal((1, 5)), (3000, 1))                      # every row identical: all distances 0
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(same, same, 6, self_mode=capi.SELF_EXCLUDE))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1) and np.all(d1 == 0)


@pytest.mark.parametrize("per_row", [1, 3])
def test_bucket_overflow_is_repaired(sym, per_row, monkeypatch):
    """buckets far too small for the row-side candidates: the overflowing blocks are searched again exhaustively
    inside the same call, and nothing changes in the result"""
    capi = sym
    Y = _data(30000, 8, 5)
    capi.set_sym_mode(capi.SYM_OFF)
    d0, i0 = capi.knn(Y, Y, 7, self_mode=capi.SELF_EXCLUDE)
    monkeypatch.setenv("MCE_SYM_BUCKET", str(per_row))
    capi.set_sym_mode(capi.SYM_FORCE)
    d1, i1 = capi.knn(Y, Y, 7, self_mode=capi.SELF_EXCLUDE)
    assert "symmetric" in capi.last_kernel() and "bucket=%d" % (per_row * 512) in capi.last_kernel()
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)


@pytest.mark.parametrize("n,d,kmax", [(40000, 27, 10), (30001, 6, 5), (5000, 15, 3)])
def test_fused_reduction_and_class_with_symmetric_sweep(sym, n, d, kmax):
    capi = sym
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    chain = gaussian_chain(seed=n, n=n, d=d, weights="int", cov="corr")
    X = _data(n, d, 77)
    w = np.random.default_rng(1).integers(1, 5, n).astype(float)
    fs = -np.random.default_rng(2).random(n)
    (p0, q0), (p1, q1) = _both(capi, lambda: capi.knn_dotp(X, None, w, fs, kmax, 1, return_dist=True))
    assert np.array_equal(q0, q1)                          # the distances that entered the sums
    assert np.allclose(p0, p1, rtol=1e-13, atol=0)         # sums: same terms, different association (list columns are in sorted-row order)
    assert np.array_equal(p1, capi.knn_dotp(X, None, w, fs, kmax, 1))       # and reproducible run to run, whatever order the candidates arrive in
    a, b = _both(capi, lambda: pkg.MCEvidence([chain], kmax=kmax, verbose=0).evidence())
    assert np.max(np.abs(a - b)) < 1e-12
    ref = orc.evidence_from_chain(chain, kmax=kmax, knn="brute")
    assert np.allclose(b, ref["lnE"], rtol=0, atol=LNE_TOL)


def test_symmetric_sweep_needs_one_buffer(sym):
    """separate query and reference sets of equal size, shards and cross evidence take the exhaustive sweep"""
    capi = sym
    capi.set_sym_mode(capi.SYM_FORCE)
    X, Y = _data(6000, 5, 1), _data(6000, 5, 2)
    d1, i1 = capi.knn(X, Y, 4)
    assert "symmetric" not in capi.last_kernel()
    od, oi = orc.knn_brute(X, Y, 4)
    assert _rel(d1, od) < DIST_RTOL and np.array_equal(i1, oi)
    d2, i2 = capi.knn(Y[1000:3000], Y, 4, self_mode=capi.SELF_EXCLUDE, self_offset=1000)
    assert "symmetric" not in capi.last_kernel()
    full, fi = capi.knn(Y, Y, 4, self_mode=capi.SELF_EXCLUDE)
    assert "symmetric" in capi.last_kernel()
    assert np.array_equal(full[1000:3000], d2) and np.array_equal(fi[1000:3000], i2)


def test_symmetric_sweep_golden_lnE(sym):
    """the reference's own ln E (golden vectors recorded from /root/reference) with the symmetric sweep forced on"""
    capi = sym
    import mcevidence_amd as pkg
    capi.set_sym_mode(capi.SYM_FORCE)
    done = 0
    for name in sorted(G):
        case = G[name]
        if case["tag"] == "big" or case["mce"].get("split") or case["S"] < 2000:
            continue
        mce = pkg.MCEvidence([chain_of(case)], verbose=0, **case["mce"])
        lnE = mce.evidence(**case["ev"])
        assert np.allclose(lnE, case["lnE"], rtol=0, atol=LNE_TOL), name
        done += "symmetric" in capi.last_kernel()
    assert done >= 3


def test_symmetric_sweep_under_graph_capture(sym):
    """the *_dev entry points only enqueue -- sort, prepass, units, repair and merge included -- so the call can be
    captured into a HIP graph and replayed on new data in the same buffers"""
    import torch
    capi = sym
    capi.set_sym_mode(capi.SYM_FORCE)
    rng = np.random.default_rng(21)
    n, d, kmax = 30000, 9, 4
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
    assert "symmetric" in capi.last_kernel()
    capi.set_sym_mode(capi.SYM_OFF)
    for h in data[1:]:
        X.copy_(torch.from_numpy(h))
        graph.replay()
        torch.cuda.synchronize()
        want_dotp, want_dist = capi.knn_dotp(h, None, np.ones(n), np.zeros(n), kmax, 1, return_dist=True)
        assert "symmetric" not in capi.last_kernel()
        assert np.array_equal(dd.cpu().numpy(), want_dist)
        assert np.allclose(out.cpu().numpy(), want_dotp, rtol=1e-13, atol=0)


@pytest.mark.parametrize("per_row", [None, 3000])
def test_replays_interleaved_with_eager_symmetric_calls(sym, per_row, monkeypatch):
    """a replayed graph must clear its own counters: with eager symmetric searches of this library between the replays
    (their own buffers, their own clears) a memset NODE was seen to leave the bucket counts of the captured call as
    they were -- garbage counts, stores far outside the workspace (zero_fill.hpp).  No seed phase at this size: every
    pair goes through the row side, the regime with the most bucket traffic."""
    import torch
    capi = sym
    if per_row:
        monkeypatch.setenv("MCE_SYM_BUCKET", str(per_row))
    capi.set_sym_mode(capi.SYM_FORCE)
    rng = np.random.default_rng(12)
    kmax = 4
    K = kmax - 1

    def buffers(n, d):
        wsb = capi.knn_workspace_bytes(n, n, d, K) + capi.dotp_workspace_bytes(n, kmax)
        return (torch.empty((n, d), dtype=torch.float64, device="cuda"), torch.ones(n, dtype=torch.float64, device="cuda"),
                torch.zeros(n, dtype=torch.float64, device="cuda"), torch.empty(wsb, dtype=torch.uint8, device="cuda"),
                torch.zeros(kmax, dtype=torch.float64, device="cuda"), torch.zeros((n, K), dtype=torch.float64, device="cuda"), wsb)
    side = torch.cuda.Stream()
    n, d = 30000, 6                                     # an earlier, larger call whose buffers go back to the allocator
    X, w, fs, ws, out, dd, wsb = buffers(n, d)
    X.copy_(torch.from_numpy(rng.standard_normal((n, d))))
    with torch.cuda.stream(side):
        capi.knn_dotp_dev(X.data_ptr(), n, X.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(),
                          dd.data_ptr(), ws.data_ptr(), wsb, side.cuda_stream)
    side.synchronize()
    del X, w, fs, ws, out, dd
    n, d = 20000, 5
    X, w, fs, ws, out, dd, wsb = buffers(n, d)

    def call(stream):
        capi.knn_dotp_dev(X.data_ptr(), n, X.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(),
                          dd.data_ptr(), ws.data_ptr(), wsb, stream)
    X.copy_(torch.from_numpy(rng.standard_normal((n, d))))
    with torch.cuda.stream(side):
        call(side.cuda_stream)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(torch.cuda.current_stream().cuda_stream)
    for it in range(5):
        h = rng.standard_normal((n, d))
        X.copy_(torch.from_numpy(h))
        graph.replay()
        torch.cuda.synchronize()
        hs = h if it % 2 == 0 else h[:3000]
        want_dotp, want_dist = capi.knn_dotp(hs, None, np.ones(len(hs)), np.zeros(len(hs)), kmax, 1, return_dist=True)
        assert "symmetric" in capi.last_kernel()
        if it % 2 == 0:
            assert np.array_equal(dd.cpu().numpy(), want_dist)
            assert np.allclose(out.cpu().numpy(), want_dotp, rtol=1e-13, atol=0)


@pytest.mark.parametrize("mode", ["symmetric", "pruned"])
def test_graph_capture_above_one_million_rows(sym, mode):
    """past ~1 M keys rocPRIM's radix sort clears its scratch with memset calls, which do not survive graph replay on this
    stack: a captured call sorts through the merge-sort path instead (prune.hip, sort_pairs) -- same permutation, same
    results on every replay"""
    import torch
    capi = sym
    capi.set_sym_mode(capi.SYM_FORCE if mode == "symmetric" else capi.SYM_OFF)
    capi.set_prune_mode(capi.PRUNE_OFF if mode == "symmetric" else capi.PRUNE_FORCE)
    rng = np.random.default_rng(77)
    n, d, kmax = 1_200_000, 3, 3
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
    X.copy_(torch.from_numpy(rng.standard_normal((n, d))))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        call(side.cuda_stream)
    side.synchronize()
    assert mode in capi.last_kernel()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(torch.cuda.current_stream().cuda_stream)
    for it in range(3):
        h = rng.standard_normal((n, d)) * (1.0 + it)
        X.copy_(torch.from_numpy(h))
        graph.replay()
        torch.cuda.synchronize()
        want_dotp, want_dist = capi.knn_dotp(h, None, np.ones(n), np.zeros(n), kmax, 1, return_dist=True)
        assert mode in capi.last_kernel()
        assert np.array_equal(dd.cpu().numpy(), want_dist)
        assert np.allclose(out.cpu().numpy(), want_dotp, rtol=1e-12, atol=0)


def test_random_shapes_with_random_settings(sym, monkeypatch):
    """seeded sweep over ragged shapes and every self mode with random prepass sizes and layouts, panel sizes (units per
    block) and bucket sizes (overflow + repair in some cases): distances and rows identical to the exhaustive sweep's,
    and equal to the exact CPU search on a sample"""
    capi = sym
    rng = np.random.default_rng(4242)
    for case in range(40):
        d = int(rng.choice([1, 2, 3, 5, 6, 8, 13, 14, 16, 27, 31, 33, 40, 50]))
        n = int(rng.integers(1025, 40000))
        K = int(rng.integers(1, 17))
        sm = [capi.SELF_EXCLUDE, capi.SELF_INCLUDE, capi.SELF_NONE][case % 3]
        Y = rng.standard_normal((n, d)) * rng.uniform(0.1, 30.0) + rng.standard_normal(d) * rng.uniform(0, 50.0)
        if case % 5 == 0:
            Y[rng.integers(0, n, n // 3)] = Y[rng.integers(0, n, n // 3)]          # exact duplicates
        if case % 7 == 3:
            Y = np.round(Y)                                                         # a lattice: ties at every distance
        monkeypatch.setenv("MCE_SYM_SEED_ROWS", str(int(rng.integers(64, 16384))))
        monkeypatch.setenv("MCE_SYM_SEED_SHARE", "2")
        monkeypatch.setenv("MCE_SYM_SEED_MODE", str(int(rng.integers(0, 3))))
        monkeypatch.setenv("MCE_SYM_PANEL", str(int(rng.choice([1, 2, 3, 7, 96]))))
        monkeypatch.setenv("MCE_SYM_BUCKET", str(int(rng.choice([1, 2, 8, 200]))))
        (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, K, self_mode=sm))
        tag = (case, d, n, K, sm, capi.last_kernel())
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1), tag
        pick = np.sort(rng.choice(n, 300, replace=False))
        od, oi = orc.knn_brute(Y[pick], Y, K, self_mode=0)        # (self modes differ only in the own row's handling: check values)
        if sm == capi.SELF_NONE:
            assert np.allclose(d1[pick], od, rtol=DIST_RTOL, atol=0), tag


SYM_GOLD = [n for n in sorted(G) if G[n]["tag"] == "sym"]


@pytest.mark.parametrize("name", SYM_GOLD)
def test_reference_goldens_in_the_automatic_range(name):
    """golden vectors recorded from the reference itself (oracle/gen_golden.py --sym) in the size range where the
    automatic mode switches to the symmetric sweep (more than one round of query blocks: 135 k x 27 and 140 k x 20
    take it, 70 k x 45 stays with the seeded exhaustive sweep -- and is run again forced): the class reproduces the
    reference's ln E, and the search the reference's sampled DkNN rows (what sklearn returned inside evidence())"""
    from mcevidence_amd import _capi as capi
    import mcevidence_amd as pkg
    capi.set_search_mode(capi.MODE_AUTO)
    capi.set_prune_mode(capi.PRUNE_AUTO)
    capi.set_sym_mode(capi.SYM_AUTO)
    case = G[name]
    chain = chain_of(case)
    mce = pkg.MCEvidence([chain], verbose=0, **case["mce"])
    lnE = mce.evidence(**case["ev"])
    blocks = (len(chain) + 511) // 512
    assert ("symmetric" in capi.last_kernel()) == (blocks > 256), capi.last_kernel()
    assert np.allclose(lnE, case["lnE"], rtol=0, atol=LNE_TOL), (lnE, case["lnE"])
    if blocks <= 256:
        capi.set_sym_mode(capi.SYM_FORCE)
        try:
            lnE = mce.evidence(**case["ev"])
            assert "symmetric" in capi.last_kernel(), capi.last_kernel()
            assert np.allclose(lnE, case["lnE"], rtol=0, atol=LNE_TOL), (lnE, case["lnE"])
        finally:
            capi.set_sym_mode(capi.SYM_AUTO)
    # the reference's whitened rows and neighbour distances at the sampled rows
    a = case["arrays"]
    theta = chain[:, 2:2 + case["ndim"]]
    cs = orc.covariance_eig(theta)
    X = np.ascontiguousarray(orc.whiten(theta, cs["eVec"], cs["eVal"]))
    assert np.allclose(X[a["rows"]], a["X_rows"], rtol=1e-11, atol=1e-12)
    kmax = case["kmax"]
    capi.set_sym_mode(capi.SYM_FORCE)
    try:
        d, _ = capi.knn(X, X, kmax, self_mode=capi.SELF_EXCLUDE)
    finally:
        capi.set_sym_mode(capi.SYM_AUTO)
    assert "symmetric" in capi.last_kernel()
    assert np.allclose(d[a["rows"]][:, :kmax - 1], a["DkNN_rows"][:, 1:kmax], rtol=DIST_RTOL, atol=0)



@pytest.mark.parametrize("d", [5, 20])
def test_neighbours_that_only_the_row_side_can_deliver(sym, d):
    """isolated points near the mean come first in the sorted order; their K nearest all lie in tight clusters far
    out, which later blocks handle -- and those blocks' own bounds are tiny (their neighbours are inside the cluster),
    so the pairs survive only because the streamed rows' (large) bounds are honoured: checked against the exact CPU search"""
    capi = sym
    rng = np.random.default_rng(77 + d)
    nclu, per = 60, 400
    centres = rng.standard_normal((nclu, d))
    centres *= (8.0 + 4.0 * rng.random((nclu, 1))) / np.linalg.norm(centres, axis=1, keepdims=True)     # shell of radius 8..12
    clusters = (centres[:, None, :] + 1e-3 * rng.standard_normal((nclu, per, d))).reshape(-1, d)
    lonely = 0.5 * rng.standard_normal((600, d))                                                       # near the mean, far from every cluster
    Y = np.concatenate([clusters, lonely])
    Y = np.ascontiguousarray(Y[rng.permutation(len(Y))])
    K = 6
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    od, oi = orc.knn_brute(Y, Y, K, self_mode=2)
    assert np.allclose(d1, od, rtol=DIST_RTOL, atol=0) and np.mean(i1 == oi) > 0.999
    # and the other way round: one tight cluster at the mean, isolated points far out
    core = 1e-3 * rng.standard_normal((20000, d))
    far = rng.standard_normal((300, d))
    far *= (20.0 + 10.0 * rng.random((300, 1))) / np.linalg.norm(far, axis=1, keepdims=True)
    Z = np.concatenate([core, far])
    Z = np.ascontiguousarray(Z[rng.permutation(len(Z))])
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Z, Z, K, self_mode=capi.SELF_EXCLUDE))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    od, oi = orc.knn_brute(Z[-2000:], Z, K, self_mode=2, self_offset=len(Z) - 2000)
    assert np.allclose(d1[-2000:], od, rtol=DIST_RTOL, atol=0)


def test_sampled_rows_at_2M_rows():
    """beyond BASELINE's largest high-dimensional size: 2M x 27 through the automatic symmetric sweep (7631 units of
    per panel at most, 28 panels); 300 sampled rows against the exact CPU search, and size-independent properties of the
    whole result: ascending rows, no own row, mutual consistency of nearest neighbours"""
    from mcevidence_amd import _capi as capi
    capi.set_search_mode(capi.MODE_AUTO)
    capi.set_prune_mode(capi.PRUNE_AUTO)
    capi.set_sym_mode(capi.SYM_AUTO)
    rng = np.random.default_rng(2027)
    n, d, K = 2_000_000, 27, 9
    X = rng.standard_normal((n, d))
    dist, idx = capi.knn(X, X, K, self_mode=capi.SELF_EXCLUDE)
    assert "symmetric" in capi.last_kernel(), capi.last_kernel()
    assert np.all(np.diff(dist, axis=1) >= 0) and np.all(idx != np.arange(n)[:, None]) and np.all(dist > 0)
    # mutual consistency (what a symmetric search must get right): if j is i's nearest neighbour, then i is in j's
    # list, or all K of j's neighbours are at most as far from j as i is
    j = idx[:, 0]
    assert np.all((idx[j] == np.arange(n)[:, None]).any(axis=1) | (dist[j, K - 1] <= dist[:, 0]))
    rows = np.sort(rng.choice(n, 300, replace=False))
    od, oi = orc.knn_brute(X[rows], X, K + 1)
    keep = np.array([[c for c in oi[r] if c != rows[r]][:K] for r in range(len(rows))])
    want = np.sqrt(((X[rows][:, None, :] - X[keep]) ** 2).sum(-1))
    assert np.allclose(dist[rows], want, rtol=DIST_RTOL, atol=0) and np.array_equal(idx[rows], keep)


# --------------------------------------------------------------------------- multi-GPU partition of the symmetric sweep
@pytest.mark.parametrize("n,d,kmax,W", [(40000, 27, 10, 2), (30001, 6, 5, 3), (150000, 20, 6, 4), (70001, 27, 10, 8), (5000, 15, 3, 4), (2000, 6, 4, 8)])
def test_symmetric_partition_parts_add_up_to_the_whole(sym, n, d, kmax, W):
    """mce_knn_dotp_part_f64 for auto evidence: every rank takes a contiguous range of the SORTED blocks, symmetric within
    its range and column side only against the other ranks' rows.  All W shares computed one after the
    other on this GPU add up to the single-rank sum (which itself equals the oracle's) -- whatever W, also when ranks
    end up with no blocks at all (5000 rows = 10 blocks over 4 ranks is fine; 2000 rows over 8 ranks take query shards, as
    every job of more than four ranks does)."""
    capi = sym
    rng = np.random.default_rng(n + W)
    Y = _data(n, d, n + 3 * W)
    w = rng.integers(1, 6, n).astype(np.float64)
    fs = -0.5 * rng.random(n) * 5.0
    capi.set_sym_mode(capi.SYM_OFF)
    whole = capi.knn_dotp(Y, None, w, fs, kmax, 1)
    assert "symmetric" not in capi.last_kernel()
    capi.set_sym_mode(capi.SYM_FORCE)
    parts = []
    for r in range(W):
        parts.append(capi.knn_dotp_part(Y, w, fs, kmax, r, W))
        # up to four ranks the symmetric partition, beyond that query shards (capi_common.hpp: kSymPartitionMaxParts)
        assert ("panel-kernel" in capi.last_kernel()) == (W <= 4), capi.last_kernel()
    total = np.sum(parts, axis=0)
    assert np.allclose(total[1:], whole[1:], rtol=1e-12, atol=0), (total, whole)
    if n <= 40000:
        od, _ = orc.knn_brute(Y, Y, kmax - 1, self_mode=2)
        full = np.zeros((n, kmax)); full[:, 1:] = od
        assert np.allclose(total[1:], orc.dotp_literal(full, w, fs, d, 1, kmax)[1:], rtol=1e-10, atol=0)


def _tools():
    import os, sys
    t = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if t not in sys.path:
        sys.path.insert(0, t)
    import pairs_once_emulate
    return pairs_once_emulate


@pytest.mark.parametrize("n,d,kmax,W", [(40000, 27, 10, 2), (30001, 6, 5, 3), (150000, 20, 6, 4), (70001, 27, 10, 8), (33000, 15, 17, 5), (5000, 15, 3, 4),
                                        (20000, 27, 10, 6), (9000, 40, 5, 7)])
def test_pairs_once_partition_parts_add_up_to_the_whole(sym, n, d, kmax, W):
    """mce_pairs_once_*: every pair of rows multiplied once per NODE -- rank r runs the single-GPU units of the sorted blocks
    r, r + W, ... (block a against the tiles of the blocks 0..a, both gates on) and ships the candidates it found for rows it
    does not own to their owners.  All W shares computed one after the other on this GPU, each in its own workspace, the
    exchange done with device copies (tools/pairs_once_emulate.py): the results add up to the single-rank sum, which equals
    the oracle's -- odd and even W, sizes that are not whole blocks, ranks of one or two blocks."""
    capi = sym
    rng = np.random.default_rng(n + W)
    Y = _data(n, d, n + 3 * W)
    w = rng.integers(1, 6, n).astype(np.float64)
    fs = -0.5 * rng.random(n) * 5.0
    capi.set_sym_mode(capi.SYM_OFF)
    whole = capi.knn_dotp(Y, None, w, fs, kmax, 1)
    assert "symmetric" not in capi.last_kernel()
    assert capi.pairs_once_blocks(n, d, kmax) == 0            # (the exhaustive sweep was asked for: no partition of the symmetric one)
    capi.set_sym_mode(capi.SYM_FORCE)
    assert capi.pairs_once_blocks(n, d, kmax) == (n + 511) // 512
    r = _tools().emulate(Y, w, fs, kmax, W)
    assert "pairs-once" in r["kernel"] and "panel-kernel" in r["kernel"], r["kernel"]
    assert sum(r["candidates_sent"]) == sum(r["candidates_received"]) > 0 and r["flagged_blocks"] == 0
    assert np.allclose(r["dotp"][1:], whole[1:], rtol=1e-12, atol=0), (r["dotp"], whole)
    if n <= 40000:
        od, _ = orc.knn_brute(Y, Y, kmax - 1, self_mode=2)
        full = np.zeros((n, kmax)); full[:, 1:] = od
        assert np.allclose(r["dotp"][1:], orc.dotp_literal(full, w, fs, d, 1, kmax)[1:], rtol=1e-10, atol=0)


def test_pairs_once_partition_overflow_and_give_up(sym, monkeypatch):
    """Buckets that overflow on ANOTHER rank (MCE_SYM_BUCKET=1: a block's bucket holds 512 candidates) reach the owner as
    flags -- MAX over the ranks -- and the owner searches those blocks again exhaustively; units that give up waiting
    (MCE_SYM_SPIN_LIMIT=0) flag their own block.  The sums do not change."""
    capi = sym
    n, d, kmax, W = 60000, 20, 7, 3
    rng = np.random.default_rng(5)
    Y = _data(n, d, 11)
    w = rng.integers(1, 4, n).astype(np.float64)
    fs = -rng.random(n)
    capi.set_sym_mode(capi.SYM_OFF)
    whole = capi.knn_dotp(Y, None, w, fs, kmax, 1)
    capi.set_sym_mode(capi.SYM_FORCE)
    monkeypatch.setenv("MCE_SYM_BUCKET", "1")
    r = _tools().emulate(Y, w, fs, kmax, W)
    assert r["flagged_blocks"] > 0
    assert np.allclose(r["dotp"][1:], whole[1:], rtol=1e-12, atol=0)
    monkeypatch.delenv("MCE_SYM_BUCKET")
    monkeypatch.setenv("MCE_SYM_PANEL", "2")
    monkeypatch.setenv("MCE_SYM_SPIN_LIMIT", "0")
    r = _tools().emulate(Y, w, fs, kmax, W)
    assert np.allclose(r["dotp"][1:], whole[1:], rtol=1e-12, atol=0)


def test_pairs_once_partition_refuses_what_it_cannot_do(sym):
    capi = sym
    import torch
    capi.set_sym_mode(capi.SYM_FORCE)
    assert capi.pairs_once_blocks(30000, 20, 19) == 0          # K = 18 > 16: two passes -- not partitioned this way
    n, d, kmax = 30000, 20, 6
    Y = torch.from_numpy(_data(n, d, 3)).cuda()
    wsb = capi.pairs_once_workspace_bytes(n, d, kmax, 2)
    assert wsb >= capi.knn_workspace_bytes(n, n, d, kmax - 1) + capi.dotp_workspace_bytes(n, kmax) and capi.pairs_once_workspace_bytes(n, d, 19, 2) == 0
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda"); fl = torch.zeros(capi.pairs_once_blocks(n, d, kmax), dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        capi.pairs_once_prepare_dev(Y.data_ptr(), n, d, kmax, 0, 1, ws.data_ptr(), wsb, 0)         # one rank: nothing to exchange
    with pytest.raises(ValueError):
        capi.pairs_once_sweep_dev(Y.data_ptr(), n, d, kmax, 2, 2, cnt.data_ptr(), fl.data_ptr(), ws.data_ptr(), wsb, 0)
    with pytest.raises(ValueError):
        capi.pairs_once_prepare_dev(Y.data_ptr(), n, d, kmax, 0, 2, ws.data_ptr(), wsb - 4096, 0)
    # an entry for a row the rank does not own poisons the result instead of being dropped silently
    off, nb = capi.pairs_once_prepare_dev(Y.data_ptr(), n, d, kmax, 0, 2, ws.data_ptr(), wsb, 0)
    bounds = ws[off:off + 8 * nb].view(torch.float64)
    assert nb == 512 * capi.pairs_once_blocks(n, d, kmax) and bool(torch.isfinite(bounds[:512]).all()) and bool(torch.isinf(bounds[512:1024]).all())
    capi.pairs_once_sweep_dev(Y.data_ptr(), n, d, kmax, 0, 2, cnt.data_ptr(), fl.data_ptr(), ws.data_ptr(), wsb, 0)
    w = torch.ones(n, dtype=torch.float64, device="cuda"); fs = torch.zeros(n, dtype=torch.float64, device="cuda")
    out = torch.zeros(kmax, dtype=torch.float64, device="cuda")
    bad = torch.zeros((1, 2), dtype=torch.float64, device="cuda")
    bad.view(torch.int32)[0, 2] = 5; bad.view(torch.int32)[0, 3] = 512 + 7       # {d2 = 0, src = 5, row = 519}: block 1 is rank 1's
    capi.pairs_once_finish_dev(Y.data_ptr(), n, d, kmax, 0, 2, w.data_ptr(), fs.data_ptr(), bad.data_ptr(), 1, fl.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, 0)
    torch.cuda.synchronize()
    assert torch.isnan(out).all()
    # the four calls come in order, on one workspace, with the arguments it was prepared with: anything else is refused
    with pytest.raises(ValueError, match="out of order|not prepared"):       # finish has closed the workspace
        capi.pairs_once_sweep_dev(Y.data_ptr(), n, d, kmax, 0, 2, cnt.data_ptr(), fl.data_ptr(), ws.data_ptr(), wsb, 0)
    capi.pairs_once_prepare_dev(Y.data_ptr(), n, d, kmax, 0, 2, ws.data_ptr(), wsb, 0)
    with pytest.raises(ValueError, match="out of order"):                    # no sweep yet
        capi.pairs_once_finish_dev(Y.data_ptr(), n, d, kmax, 0, 2, w.data_ptr(), fs.data_ptr(), 0, 0, fl.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, 0)
    with pytest.raises(ValueError, match="other arguments"):                 # prepared as part 0, swept as part 1
        capi.pairs_once_sweep_dev(Y.data_ptr(), n, d, kmax, 1, 2, cnt.data_ptr(), fl.data_ptr(), ws.data_ptr(), wsb, 0)
    capi.pairs_once_sweep_dev(Y.data_ptr(), n, d, kmax, 0, 2, cnt.data_ptr(), fl.data_ptr(), ws.data_ptr(), wsb, 0)
    with pytest.raises(ValueError, match="out of order"):                    # swept twice
        capi.pairs_once_sweep_dev(Y.data_ptr(), n, d, kmax, 0, 2, cnt.data_ptr(), fl.data_ptr(), ws.data_ptr(), wsb, 0)
    torch.cuda.synchronize()


def _pairs_once_rank(rank, world, port, q, n, d, kmax):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MCE_PAIRS_ONCE="1")
    torch.cuda.set_device(0)                                 # one GPU on the test box: the ranks share it
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcevidence_amd import _capi, parallel
    _capi.set_prune_mode(_capi.PRUNE_OFF)
    _capi.set_sym_mode(_capi.SYM_FORCE)
    rng = np.random.default_rng(n)
    Y = _data(n, d, n + 1)
    w = rng.integers(1, 6, n).astype(np.float64)
    fs = -0.5 * rng.random(n) * 5.0
    stats = {}
    direct = parallel.pairs_once_knn_dotp(Y, w, fs, kmax, stats=stats)
    dotp, _ = parallel.sharded_knn_dotp(Y, None, w, fs, kmax, 1)             # MCE_PAIRS_ONCE=1: the same route
    q.put((rank, direct, dotp, stats, _capi.last_kernel()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pairs_once_partition_over_gloo_ranks(sym, world):
    """parallel.pairs_once_knn_dotp under a gloo group whose ranks share the box's one GPU: sweep -> all_gather of the counts,
    all_reduce(MAX) of the flags -> export -> all_to_all_single of the candidates -> finish -> all_reduce(sum): every rank
    returns the single-process sums (1e-12); parallel.sharded_knn_dotp takes the same route under MCE_PAIRS_ONCE=1."""
    import socket
    import torch.multiprocessing as mp
    capi = sym
    n, d, kmax = 90000, 27, 10
    rng = np.random.default_rng(n)
    Y = _data(n, d, n + 1)
    w = rng.integers(1, 6, n).astype(np.float64)
    fs = -0.5 * rng.random(n) * 5.0
    capi.set_sym_mode(capi.SYM_OFF)
    whole = capi.knn_dotp(Y, None, w, fs, kmax, 1)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pairs_once_rank, args=(r, world, port, q, n, d, kmax)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=600) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sent = recv = 0
    for r in range(world):
        direct, dotp, stats, kernel = got[r]
        assert np.allclose(direct[1:], whole[1:], rtol=1e-12, atol=0) and np.allclose(dotp[1:], whole[1:], rtol=1e-12, atol=0)
        assert np.array_equal(direct, got[0][0])                          # the all-reduce hands every rank the same sums
        assert "pairs-once" in kernel
        sent += stats["sent"]; recv += stats["received"]
    assert sent == recv > 0


@pytest.mark.parametrize("chains", [2, 5])
def test_chains_of_units_per_block_are_bit_identical(sym, monkeypatch, chains):
    """MCE_SYM_CHAINS = S: a block's panels dealt to S independent chains of units with their own list sets, merged afterwards (what
    a rank of the all-pairs-once partition does; off by default on one GPU).  Same distances, same rows, same sums -- also with
    blocks that have fewer panels than chains, with give-ups (repaired blocks: the other sets are emptied) and all self modes."""
    capi = sym
    monkeypatch.setenv("MCE_SYM_CHAINS", str(chains))
    for n, d, K in ((70001, 27, 10), (20000, 6, 4), (150000, 15, 12)):
        Y = _data(n, d, n + chains)
        for sm in (capi.SELF_EXCLUDE, capi.SELF_INCLUDE):
            (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, K, self_mode=sm))
            assert "chains" in capi.last_kernel(), capi.last_kernel()
            assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    w = np.ones(n); fs = np.zeros(n)
    (a,), (b,) = _both(capi, lambda: (capi.knn_dotp(Y, None, w, fs, K + 1, 1),))
    assert np.allclose(a[1:], b[1:], rtol=1e-12, atol=0)          # (the same terms, summed in the sorted rows' order)
    monkeypatch.setenv("MCE_SYM_PANEL", "3")
    monkeypatch.setenv("MCE_SYM_SPIN_LIMIT", "0")
    Y = _data(60000, 20, 7)
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, 6, self_mode=capi.SELF_EXCLUDE))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)


def test_unit_that_gives_up_waiting_is_repaired(sym, monkeypatch):
    """The wait of a unit for its block's previous unit is bounded (knn_panel.hpp).  With the bound at 0 every wait that
    is not already satisfied gives up at once: the unit starts from empty lists, flags its block, and the repair launch
    searches that block again exhaustively -- the result must not change.  Small panels make many units per block."""
    capi = sym
    Y = _data(60000, 20, 7)
    monkeypatch.setenv("MCE_SYM_PANEL", "2")
    (d0, i0), (d1, i1) = _both(capi, lambda: capi.knn(Y, Y, 6, self_mode=capi.SELF_EXCLUDE))
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
    monkeypatch.setenv("MCE_SYM_SPIN_LIMIT", "0")
    capi.set_sym_mode(capi.SYM_FORCE)
    d2, i2 = capi.knn(Y, Y, 6, self_mode=capi.SELF_EXCLUDE)
    assert "symmetric" in capi.last_kernel()
    assert np.array_equal(d0, d2) and np.array_equal(i0, i2)


def test_some_waves_of_a_unit_give_up(sym, monkeypatch):
    """The eight waves of a unit wait for the block's previous unit independently, so under load any subset of them can give
    up: each must flag the block (found with tools/stress_concurrent.py at a 50 us limit -- whole waves of 64 queries with
    incomplete lists in a block that only wave 0 would have flagged).  MCE_PANEL_DEBUG=16 makes the odd waves of every
    block's second unit give up at once, which reproduces it without a loaded chip (22 332 wrong rows at 60 000 x 20 before
    the fix).  It also covers what a give-up does to the rest of the block: its units no longer run one after the other,
    and lists written by two of them at once are not trusted (a flagged block's units start from empty lists)."""
    capi = sym
    monkeypatch.setenv("MCE_SYM_PANEL", "2")
    for n, d, K in ((60000, 20, 6), (45000, 45, 8), (80000, 6, 4)):
        Y = _data(n, d, n + 1)
        capi.set_sym_mode(capi.SYM_OFF)
        d0, i0 = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
        capi.set_sym_mode(capi.SYM_FORCE)
        monkeypatch.setenv("MCE_PANEL_DEBUG", "16")
        for _ in range(3):
            d1, i1 = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
            assert "panel-kernel" in capi.last_kernel()
            assert np.array_equal(d0, d1) and np.array_equal(i0, i1)
        monkeypatch.delenv("MCE_PANEL_DEBUG")


def test_every_candidate_through_the_redo_list(sym, monkeypatch):
    """A tile whose candidates do not fit the wave's queue is deferred to a redo list and multiplied again after the
    next drain.  MCE_PANEL_DEBUG=8 sends EVERY candidate of the sweep that way: identical results."""
    capi = sym
    for n, d, K in ((7777, 27, 9), (20000, 6, 10), (1024, 2, 1)):
        Y = _data(n, d, n)
        capi.set_sym_mode(capi.SYM_OFF)
        d0, i0 = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
        monkeypatch.setenv("MCE_PANEL_DEBUG", "8")
        capi.set_sym_mode(capi.SYM_FORCE)
        d1, i1 = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
        monkeypatch.delenv("MCE_PANEL_DEBUG")
        assert "panel-kernel" in capi.last_kernel()
        assert np.array_equal(d0, d1) and np.array_equal(i0, i1)


# --------------------------------------------------------------------------- many searches at once (tools/stress_concurrent.py, in the suite)
@pytest.mark.parametrize("spin_limit", [50, None])
def test_concurrent_searches_on_many_streams_and_threads_are_bit_identical(spin_limit, monkeypatch):
    """Fourteen searches of every kind -- symmetric sweep (panel kernel), exhaustive sweep with splits and seeds, pruned walk, and
    (round 6) the deep filter -- enqueued at once on as many streams, first from one thread, then from four threads (modes through the thread-scoped
    mce_options), several rounds in shuffled order: distances and sums bit-identical to the same searches run one at a time.
    With MCE_SYM_SPIN_LIMIT = 50 (~50 us) units of the symmetric sweep that meet a loaded chip give up waiting for their
    block's previous unit -- any subset of a unit's waves -- and the repair launch must make up for it (the hole
    tools/stress_concurrent.py found in round 3: only wave 0 flagged the block); with the default limit nothing gives up."""
    import threading
    import torch
    from mcevidence_amd import _capi as capi
    if spin_limit is not None:
        monkeypatch.setenv("MCE_SYM_SPIN_LIMIT", str(spin_limit))
    else:
        monkeypatch.delenv("MCE_SYM_SPIN_LIMIT", raising=False)
    rng = np.random.default_rng(11)
    shapes = [(9000, 6, 2, 0), (26862, 6, 2, 0), (60000, 8, 3, 0), (40000, 27, 10, 0),          # kind 0: exhaustive (automatic splits, seeds)
              (150000, 27, 10, 2), (120000, 45, 6, 2), (200000, 20, 4, 2), (70000, 27, 10, 2),    # kind 2: symmetric sweep
              (300000, 3, 5, 1), (400000, 6, 4, 1), (250000, 2, 3, 1), (131072, 10, 5, 0),         # kind 1: pruned walk
              (50000, 70, 5, 0), (60000, 100, 10, 0),                                               # (round 6) kind 0 at d >= 64: the deep filter
              (20000, 200, 6, 0)]                                                                   # ... at d >= 128: the long-row fp64 sweep
    modes = {0: dict(prune_mode=capi.PRUNE_OFF, sym_mode=capi.SYM_OFF), 1: dict(prune_mode=capi.PRUNE_FORCE, sym_mode=capi.SYM_OFF),
             2: dict(prune_mode=capi.PRUNE_OFF, sym_mode=capi.SYM_FORCE)}
    jobs = []
    for (n, d, kmax, kind) in shapes:
        X = torch.from_numpy(rng.standard_normal((n, d)) @ (np.eye(d) + 0.3 * rng.standard_normal((d, d)))).cuda()
        K = kmax - 1
        with capi.options(**modes[kind]):
            wsb = capi.knn_workspace_bytes(n, n, d, K) + capi.dotp_workspace_bytes(n, kmax)
        jobs.append(dict(n=n, d=d, kmax=kmax, kind=kind, X=X, w=torch.ones(n, dtype=torch.float64, device="cuda"),
                         fs=torch.zeros(n, dtype=torch.float64, device="cuda"), ws=torch.empty(wsb, dtype=torch.uint8, device="cuda"), wsb=wsb,
                         out=torch.zeros(kmax, dtype=torch.float64, device="cuda"), dd=torch.zeros((n, K), dtype=torch.float64, device="cuda"),
                         st=torch.cuda.Stream()))

    def enqueue(j, stream):
        with capi.options(**modes[j["kind"]]):
            capi.knn_dotp_dev(j["X"].data_ptr(), j["n"], j["X"].data_ptr(), j["n"], j["d"], j["kmax"], 1, 0, j["w"].data_ptr(), j["fs"].data_ptr(),
                              j["out"].data_ptr(), j["dd"].data_ptr(), j["ws"].data_ptr(), j["wsb"], stream)
            return capi.last_kernel()

    ref = []
    for j in jobs:                                  # one at a time
        k = enqueue(j, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ref.append((j["dd"].cpu().numpy().copy(), j["out"].cpu().numpy().copy()))
        assert ("pruned" in k) == (j["kind"] == 1) and ("symmetric" in k) == (j["kind"] == 2), k
        if j["kind"] == 2:
            assert "panel-kernel" in k
        assert ("knn_deep_kernel" in k) == (64 <= j["d"] <= 127) and ("knn_long_kernel" in k) == (j["d"] >= 128), k

    def check(tag):
        bad = []
        for i, j in enumerate(jobs):
            dd, out = j["dd"].cpu().numpy(), j["out"].cpu().numpy()
            if not (np.array_equal(dd, ref[i][0]) and np.array_equal(out, ref[i][1])):
                bad.append((tag, j["n"], j["d"], j["kind"], int(np.any(dd != ref[i][0], axis=1).sum())))
        assert not bad, bad

    for rnd in range(3):                            # all at once, from ONE thread
        for j in jobs:
            j["dd"].zero_(); j["out"].zero_()
        torch.cuda.synchronize()
        for i in rng.permutation(len(jobs)):
            enqueue(jobs[i], jobs[i]["st"].cuda_stream)
        torch.cuda.synchronize()
        check("streams round %d" % rnd)
    errors = []

    def worker(mine):
        try:
            for _ in range(2):
                for i in mine:
                    enqueue(jobs[i], jobs[i]["st"].cuda_stream)
        except Exception as e:       # noqa: BLE001
            errors.append(repr(e))

    for rnd in range(2):                            # ... and from FOUR threads, every thread a mix of kinds
        for j in jobs:
            j["dd"].zero_(); j["out"].zero_()
        torch.cuda.synchronize()
        order = rng.permutation(len(jobs))
        th = [threading.Thread(target=worker, args=([int(i) for i in order[t::4]],)) for t in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        assert not errors, errors
        check("threads round %d" % rnd)
