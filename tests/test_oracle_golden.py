"""Pins the CPU oracle against golden vectors produced by the reference itself
(oracle/gen_golden.py imported /root/reference/MCEvidence.py in the build container).
CPU only."""
import math

import numpy as np
import pytest

from helpers import DIST_RTOL, LNE_TOL, chain_of, explicit_split_of, load_golden, orc

G = load_golden()
SMALL = [n for n, c in G.items() if c["tag"] == "small"]
MEDIUM = [n for n, c in G.items() if c["tag"] == "medium"]


def _oracle_run(case, knn):
    a = case["arrays"]
    mk, ek = case["mce"], case["ev"]
    kw = dict(ndim=mk.get("ndim"), kmax=mk.get("kmax", 5), priorvolume=ek.get("pvolume") or mk.get("priorvolume", 1.0),
              pos_lnp=ek.get("pos_lnp", False), knn=knn)
    cov = ek.get("covtype", "all")
    kw["covtype"] = mk.get("covtype", "single") if cov is None else cov
    if mk.get("split"):
        kw["s1_idx"], kw["s2_idx"] = a["s1_idx"], a["s2_idx"]
    return orc.evidence_from_chain(chain_of(case), **kw)


@pytest.mark.parametrize("name", SMALL)
def test_oracle_reproduces_reference_lnE_sklearn(name):
    """numpy restatement + the reference's own sklearn call == reference output."""
    case = G[name]
    out = _oracle_run(case, "sklearn")
    assert np.allclose(out["lnE"], case["lnE"], rtol=0, atol=1e-11)
    assert math.isclose(out["J"], case["J"], rel_tol=1e-12)
    assert math.isclose(out["SumW"], case["SumW"], rel_tol=1e-14)
    assert math.isclose(out["logLmax"], case["logLmax"], rel_tol=1e-14, abs_tol=1e-14)
    assert np.allclose(out["dotp"][case["k0"]:], np.array(case["dotp"])[case["k0"]:], rtol=1e-11)
    assert out["S"] == case["S"] and out["k0"] == case["k0"]


@pytest.mark.parametrize("name", SMALL + MEDIUM)
def test_oracle_bruteforce_knn_matches_reference(name):
    """the independent exact C search gives the reference's distances and ln E."""
    case = G[name]
    out = _oracle_run(case, "brute")
    assert np.allclose(out["lnE"], case["lnE"], rtol=0, atol=LNE_TOL)
    rows = case["arrays"]["rows"]
    ref_rows = case["arrays"]["DkNN_rows"]
    got = out["DkNN"][rows]
    k0 = case["k0"]      # auto mode: sklearn's column 0 is the point itself (0 or ~1e-8 on the GEMM path)
    assert np.allclose(got[:, k0:], ref_rows[:, k0:], rtol=DIST_RTOL, atol=0)
    assert np.allclose(out["X"][rows], case["arrays"]["X_rows"], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("name", SMALL[:6])
def test_dotp_forms_agree(name):
    """literal pow/gamma form (reference) == log-domain form (HIP kernel) == C restatement."""
    case = G[name]
    out = _oracle_run(case, "brute")
    k0, kmax = out["k0"], out["kmax"]
    lit = orc.dotp_literal(out["DkNN"], out["w"], out["fs"], out["ndim"], k0, kmax)
    logd = orc.dotp_logdomain(out["DkNN"], out["w"], out["fs"], out["ndim"], k0, kmax)
    cc = orc.dotp_c(out["DkNN"], out["w"], out["fs"], out["ndim"], k0, kmax)
    assert np.allclose(logd[k0:], lit[k0:], rtol=1e-12)
    assert np.allclose(cc[k0:], lit[k0:], rtol=1e-12)


def test_dotp_signed_weights_and_zero_likelihood_rows():
    """a negative weight gives the reference's signed term volume/weight*exp(fs); fs = -inf (likelihood 0) and
    r = 0 give zero terms -- in the literal form and in both log-domain restatements"""
    rng = np.random.default_rng(11)
    n, ndim, k0, kmax = 400, 5, 1, 4
    dist = np.abs(rng.standard_normal((n, kmax))) + 0.1
    dist[5, 2] = 0.0
    w = rng.integers(1, 5, n).astype(float)
    w[[3, 77, 200]] = [-2.0, -1.0, -0.5]
    fs = -rng.random(n)
    fs[[9, 77]] = -np.inf
    with np.errstate(divide="ignore"):
        lit = orc.dotp_literal(dist, w, fs, ndim, k0, kmax)
    logd = orc.dotp_logdomain(dist, w, fs, ndim, k0, kmax)
    cc = orc.dotp_c(dist, w, fs, ndim, k0, kmax)
    assert np.all(np.isfinite(lit[k0:]))
    assert np.allclose(logd[k0:], lit[k0:], rtol=1e-12) and np.allclose(cc[k0:], lit[k0:], rtol=1e-12)
    pos = orc.dotp_literal(dist, np.abs(w), fs, ndim, k0, kmax)
    assert np.all(lit[k0:] < pos[k0:])


def test_knn_brute_modes_and_numpy_agree():
    rng = np.random.default_rng(5)
    X = rng.standard_normal((700, 5))
    d0, i0 = orc.knn_brute(X, X, 6)
    d2, i2 = orc.knn_brute(X, X, 5, self_mode=2)
    assert np.all(d0[:, 0] == 0.0) and np.array_equal(i0[:, 0], np.arange(700))
    assert np.array_equal(d0[:, 1:], d2) and np.array_equal(i0[:, 1:], i2)
    dn, inn = orc.knn_numpy(X, X, 5, exclude_self=True)
    assert np.allclose(dn, d2, rtol=1e-13) and np.array_equal(inn, i2)
    # sharded self-exclusion: shard [200,450) with self_offset=200 == rows of the full result
    ds, is_ = orc.knn_brute(X[200:450], X, 5, self_mode=2, self_offset=200)
    assert np.array_equal(ds, d2[200:450]) and np.array_equal(is_, i2[200:450])
    with pytest.raises(ValueError):
        orc.knn_brute(X[:3], X[:3], 3, self_mode=2)


def test_duplicates_give_zero_volume_terms():
    """repeated rows (thinning replicates rows): r=0 -> zero volume, no NaN (SURVEY hard parts)."""
    rng = np.random.default_rng(6)
    X = rng.standard_normal((300, 4))
    X[10] = X[11] = X[12]
    d, _ = orc.knn_brute(X, X, 3, self_mode=2)
    assert d[10, 0] == 0.0 and d[10, 1] == 0.0
    full = np.zeros((300, 4)); full[:, 1:] = d
    w = np.ones(300); fs = np.zeros(300)
    for f in (orc.dotp_literal, orc.dotp_logdomain, orc.dotp_c):
        out = f(full, w, fs, 4, 1, 4)
        assert np.all(np.isfinite(out[1:])) and np.all(out[1:] > 0)


def test_oracle_reproduces_the_reference_on_an_explicit_pair_C4_shape():
    """BASELINE configs[3] (C4) at 1/50 of its size: two independent chains, s1 = the first, s2 = the second, d = 15,
    kmax = 4, cross evidence (k0 = 0).  The reference was handed that split (oracle/gen_golden.py --c4 --c4-n 20000); the
    oracle with the same pair reproduces its ln E (all kmax columns), its dotp and its kd_tree distances."""
    case = G["cross_n20000_d15_k4_C4"]
    r1, r2 = explicit_split_of(case)
    for knn in ("sklearn", "brute"):
        out = orc.evidence_from_chain(chain_of(case), kmax=4, knn=knn, s1_idx=r1, s2_idx=r2)
        assert out["k0"] == 0 and out["S"] == case["S"] == 20000
        assert np.allclose(out["lnE"], case["lnE"], rtol=0, atol=LNE_TOL)
        assert np.allclose(out["dotp"], case["dotp"], rtol=1e-10)
        rows = case["arrays"]["rows"]
        assert np.allclose(out["DkNN"][rows][:, :4], case["arrays"]["DkNN_rows"][:, :4], rtol=DIST_RTOL, atol=0)
    assert case["fit_method"] == "kd_tree"


def test_oracle_reproduces_the_reference_at_C5_shape():
    """BASELINE configs[4] (C5) at 1/50 of its size: synth.config_chain('C5', n=200000) -- d = 6, kmax = 10, auto evidence,
    kd_tree in the reference (oracle/gen_golden.py --c5 --c5-n 200000).  The oracle reproduces ln E for the whole k sweep,
    the per-k dotp and the sampled distance rows, through the reference's sklearn call and through the exact C search."""
    case = G["auto_n200000_d6_k10_C5"]
    for knn in ("sklearn", "brute"):
        out = orc.evidence_from_chain(chain_of(case), kmax=10, knn=knn)
        assert out["k0"] == 1 and out["S"] == case["S"] == 200000
        assert np.allclose(out["lnE"], case["lnE"], rtol=0, atol=LNE_TOL)
        assert np.allclose(out["dotp"][1:], case["dotp"][1:], rtol=1e-10)
        rows = case["arrays"]["rows"]
        assert np.allclose(out["DkNN"][rows][:, 1:10], case["arrays"]["DkNN_rows"][:, 1:10], rtol=DIST_RTOL, atol=0)
    assert case["fit_method"] == "kd_tree"
