"""The run-time certificate of a search (mce_options.verify, mce_verify_knn_f64[_dev]; round 5): sampled query rows are
re-checked by an exact fp64 scan of all reference rows that shares nothing with the search kernels -- no matrix cores, no
packed fp16 operands, no bounds, no lists.  It must pass on every search path, and it must catch a list that misses a
neighbour, holds a distance that is too small, or carries a wrong own row (reference: the output of
`nbrs.kneighbors(samples)`, MCEvidence.py:1104)."""
import numpy as np
import pytest

from helpers import LNE_TOL, orc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def capi():
    from mcevidence_amd import _capi
    _capi.require_device()
    yield _capi
    _capi.set_prune_mode(_capi.PRUNE_AUTO)
    _capi.set_sym_mode(_capi.SYM_AUTO)
    _capi.set_search_mode(0)


def _data(n, d, seed, dup=0):
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((n, d)) @ (np.eye(d) + 0.3 * rng.standard_normal((d, d)))
    if dup:
        Y[rng.integers(0, n, dup)] = Y[rng.integers(0, n, dup)]        # exact duplicates: ties at distance 0
    return Y


@pytest.mark.parametrize("n,d,K", [(20000, 6, 4), (40000, 27, 9), (9000, 45, 16), (30000, 3, 9)])
def test_recheck_accepts_the_search_on_every_self_mode(capi, n, d, K):
    Y = _data(n, d, n + d, dup=20)
    X = _data(3000, d, 5)
    for sm, Q in ((capi.SELF_EXCLUDE, Y), (capi.SELF_INCLUDE, Y), (capi.SELF_NONE, X)):
        dist, _ = capi.knn(Q, Y, K, self_mode=sm)
        assert capi.verify_knn(Q, Y, dist, self_mode=sm, nsample=len(Q)) == 0          # every row
        assert capi.verify_knn(Q, Y, dist, self_mode=sm, nsample=500, seed=12345) == 0
    # a query shard of the set (multi-GPU row shards): self_offset = the shard's first row
    lo = 5000
    dist, _ = capi.knn(Y[lo:lo + 4000], Y, K, self_mode=capi.SELF_EXCLUDE, self_offset=lo)
    assert capi.verify_knn(Y[lo:lo + 4000], Y, dist, self_mode=capi.SELF_EXCLUDE, self_offset=lo, nsample=4000) == 0


def test_recheck_catches_wrong_lists(capi):
    n, d, K = 30000, 10, 6
    Y = _data(n, d, 77)
    full, _ = capi.knn(Y, Y, K + 1, self_mode=capi.SELF_EXCLUDE)
    good = np.ascontiguousarray(full[:, :K])
    assert capi.verify_knn(Y, Y, good, self_mode=capi.SELF_EXCLUDE, nsample=n) == 0
    rng = np.random.default_rng(1)
    rows = rng.choice(n, 37, replace=False)
    # (a) a MISSED neighbour: the third entry dropped, the (K+1)-th moved up -- what a filter that loses a candidate would return
    bad = good.copy()
    bad[rows] = np.delete(full[rows], 2, axis=1)
    assert capi.verify_knn(Y, Y, bad, self_mode=capi.SELF_EXCLUDE, nsample=n) == len(rows)
    # (b) a distance that is too small (nothing lies that close)
    bad = good.copy()
    bad[rows, K - 1] *= 0.97
    bad[rows] = np.sort(bad[rows], axis=1)
    assert capi.verify_knn(Y, Y, bad, self_mode=capi.SELF_EXCLUDE, nsample=n) == len(rows)
    # (c) the last bits do not matter (the scan sums in another order than the search): 1e-12 relative passes
    assert capi.verify_knn(Y, Y, good * (1.0 + 1e-12), self_mode=capi.SELF_EXCLUDE, nsample=n) == 0
    # (d) the own row taken for a neighbour: lists of a search WITHOUT self-exclusion judged under it
    incl, _ = capi.knn(Y, Y, K, self_mode=capi.SELF_INCLUDE)
    assert capi.verify_knn(Y, Y, incl, self_mode=capi.SELF_EXCLUDE, nsample=n) == n
    # a sample only sees its own rows: 1 corrupted row in 30 000, 300 sampled -> found or not, never more than one
    bad = good.copy()
    bad[rows[0], K - 1] *= 1.5
    assert capi.verify_knn(Y, Y, bad, self_mode=capi.SELF_EXCLUDE, nsample=300) in (0, 1)
    assert capi.verify_knn(Y, Y, bad, self_mode=capi.SELF_EXCLUDE, nsample=n) == 1


def test_options_verify_runs_behind_every_host_entry_point_and_path(capi):
    """mce_options.verify: the re-check behind mce_knn_f64, the fused mce_knn_dotp_f64 and the evidence feed -- through the
    exhaustive sweep, the symmetric sweep and the pruned walk; results unchanged, no error raised."""
    rng = np.random.default_rng(3)
    for n, d, K, prune, sym, want in ((60000, 27, 9, capi.PRUNE_OFF, capi.SYM_FORCE, "symmetric"), (120000, 5, 4, capi.PRUNE_FORCE, capi.SYM_AUTO, "pruned"),
                                     (50000, 15, 4, capi.PRUNE_OFF, capi.SYM_OFF, "rsplit")):
        Y = rng.standard_normal((n, d))
        w = rng.integers(1, 4, n).astype(float)
        fs = -rng.random(n)
        capi.set_prune_mode(prune)
        capi.set_sym_mode(sym)
        plain = capi.knn_dotp(Y, None, w, fs, K + 1, 1)
        with capi.options(verify=2048):
            checked = capi.knn_dotp(Y, None, w, fs, K + 1, 1)
            assert want in capi.last_kernel(), capi.last_kernel()
            d1, _ = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
        assert np.array_equal(plain, checked)
        d0, _ = capi.knn(Y, Y, K, self_mode=capi.SELF_EXCLUDE)
        assert np.array_equal(d0, d1)
    # the fp64 sweep (mode 1) selects on GEMM-form keys: near-ties may swap, the distances agree to 1e-10 -- the re-check's
    # tolerance (1e-9 on the squared distance) accepts that
    capi.set_prune_mode(capi.PRUNE_OFF)
    capi.set_sym_mode(capi.SYM_OFF)
    Y = rng.standard_normal((30000, 20))
    capi.knn(Y, Y, 9, self_mode=capi.SELF_EXCLUDE, options=capi.Options(search_mode=capi.MODE_F64, verify=30000))


def test_class_with_recheck_rows_gives_the_same_lnE(capi):
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    chain = gaussian_chain(seed=4, n=150000, d=8, weights="int", cov="corr")
    plain = pkg.MCEvidence([chain], kmax=5, verbose=0).evidence()
    checked = pkg.MCEvidence([chain], kmax=5, verbose=0, backend=pkg.HipBackend(recheck_rows=1024)).evidence()
    assert np.array_equal(plain, checked)
    ref = orc.evidence_from_chain(chain[:20000], kmax=5, knn="brute")
    small = pkg.MCEvidence([chain[:20000]], kmax=5, verbose=0, backend=pkg.HipBackend(recheck_rows=20000)).evidence()
    assert np.max(np.abs(small - ref["lnE"])) < LNE_TOL
    # cross evidence and the batched route
    two = gaussian_chain(seed=9, n=60000, d=6)
    mk = lambda **kw: pkg.MCEvidence([two], kmax=4, verbose=0, **kw).set_split(np.arange(0, 30000), np.arange(30000, 60000))
    assert np.array_equal(mk().evidence(), mk(backend=pkg.HipBackend(recheck_rows=512)).evidence())
    chains = [gaussian_chain(seed=30 + i, n=7000 + 500 * i, d=6) for i in range(4)]
    a = pkg.evidence_many([pkg.MCEvidence([c], kmax=3, verbose=0) for c in chains])
    be = pkg.HipBackend(recheck_rows=256)
    b = pkg.evidence_many([pkg.MCEvidence([c], kmax=3, verbose=0, backend=be) for c in chains])
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_certificate_runs_by_default_behind_the_filter_and_not_behind_the_fp64_kernels(capi):
    """Round 6 (parity footnote b): a production call carries the check.  With no option set a host-pointer
    search that went through the fp16 filter is re-checked on 256 rows (mce_last_verify_rows says so); the fp64 sweep and the
    generic kernel -- fp64 arithmetic throughout -- are not; verify=0 turns it off, a number replaces the default; shapes the
    re-check does not handle (K > 32: the generic kernel) are searched without it and do NOT become an error."""
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    Y = _data(70000, 12, 3)                                   # from 65 536 query rows on: every call
    capi.knn(Y, Y, 5, self_mode=capi.SELF_EXCLUDE)
    assert capi.last_verify_rows() == 256 and "f16" in capi.last_kernel()
    capi.knn(Y, Y, 5, self_mode=capi.SELF_EXCLUDE)
    assert capi.last_verify_rows() == 256
    capi.knn(Y, Y, 5, self_mode=capi.SELF_EXCLUDE, options=capi.Options(verify=0))
    assert capi.last_verify_rows() == 0
    capi.knn(Y, Y, 5, self_mode=capi.SELF_EXCLUDE, options=capi.Options(verify=1000))
    assert capi.last_verify_rows() == 1000
    capi.knn(Y, Y, 5, self_mode=capi.SELF_EXCLUDE, options=capi.Options(search_mode=capi.MODE_F64))
    assert capi.last_verify_rows() == 0 and "mfma" in capi.last_kernel()
    # smaller searches (the check is launch overhead there): one call in eight
    small = Y[:20000]
    ran = []
    for _ in range(16):
        capi.knn(small, small, 5, self_mode=capi.SELF_EXCLUDE)
        ran.append(capi.last_verify_rows())
    assert sorted(set(ran)) == [0, 256] and ran.count(256) == 2
    capi.knn(small, small, 5, self_mode=capi.SELF_EXCLUDE, options=capi.Options(verify=300))      # asked for: always
    assert capi.last_verify_rows() == 300
    Y = _data(30000, 12, 3)
    # K = 40: the generic kernel; with an explicit row count the call still succeeds, unchecked
    d40, _ = capi.knn(Y[:4000], Y, 40, options=capi.Options(verify=500))
    assert capi.last_verify_rows() == 0 and d40.shape == (4000, 40) and "generic" in capi.last_kernel()
    # the fused entry points and the class
    big = _data(70000, 12, 5)
    w, fs = np.ones(len(big)), np.zeros(len(big))
    capi.knn_dotp(big, big, w, fs, 5, 1)
    assert capi.last_verify_rows() == 256
    chain = gaussian_chain(seed=4, n=70000, d=8, weights="int", cov="corr")
    a = pkg.MCEvidence([chain], kmax=5, verbose=0).evidence()
    assert capi.last_verify_rows() == 256
    b = pkg.MCEvidence([chain], kmax=5, verbose=0, backend=pkg.HipBackend(recheck_rows=0)).evidence()
    assert capi.last_verify_rows() == 0 and np.array_equal(a, b)
