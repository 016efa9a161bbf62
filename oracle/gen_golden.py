#!/usr/bin/env python
"""Golden-vector generator -- TEST INFRASTRUCTURE, runs only in the build
container (needs /root/reference, which does not exist on the GPU box).

Imports the *reference* MCEvidence module (read-only, through a three-line
compatibility shim because the reference targets old numpy/sklearn names:
SURVEY.md section 8c) and records, for seeded synthetic chains made by
``mcevidence_amd.synth``:

  * the ln-evidence vector the reference returns,
  * the scalars the hot path needs (J, SumW, logLmax, S, k0),
  * per-k ``dotp`` (``MCEvidence.py:1117``) recovered by re-running the
    reference's own pieces (get_covariance / get_samples / diagonalise_chain)
    and the same sklearn call as ``MCEvidence.py:1093-1104``,
  * a few sampled rows of the whitened samples and of DkNN.

Outputs are data only (JSON/NPZ under tests/golden/).  No reference source is
copied.  Usage:  python oracle/gen_golden.py [--small] [--medium] [--big] [--sym] [--sym2] [--host] [--c4 [--c4-n N]] [--c5 [--c5-n N]]
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def import_reference():
    import sklearn.metrics
    import sklearn.neighbors

    sklearn.neighbors.DistanceMetric = sklearn.metrics.DistanceMetric  # imported by name only
    for name, val in (("int", int), ("float", float), ("Infinity", np.inf)):
        if not hasattr(np, name):
            setattr(np, name, val)
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    import MCEvidence as ref  # noqa: E402

    # The reference globs chain files in directory-listing order, which is arbitrary; the
    # build reads them sorted.  Pin the reference to the sorted order too (harness-side
    # wrapper around glob, the reference source is untouched) so file-based pins are
    # reproducible.
    import glob as _glob
    _orig = _glob.glob

    class _SortedGlob(object):
        def __getattr__(self, k):
            return getattr(_glob, k)

        @staticmethod
        def glob(pat, *a, **k):
            return sorted(_orig(pat, *a, **k))

    ref.glob = _SortedGlob()
    return ref


def run_case(ref, name, chain_kw, mce_kw=None, ev_kw=None, seed_split=None, nrows=16):
    """One in-memory-chain case.  Returns (json_dict, npz_dict)."""
    from sklearn.neighbors import NearestNeighbors
    from mcevidence_amd.synth import gaussian_chain

    mce_kw = dict(mce_kw or {})
    ev_kw = dict(ev_kw or {})
    chain = gaussian_chain(**chain_kw)
    if seed_split is not None:
        np.random.seed(seed_split)
    t0 = time.perf_counter()
    mce = ref.MCEvidence([chain], verbose=0, **mce_kw)
    lnE = mce.evidence(**ev_kw)
    wall = time.perf_counter() - t0

    # --- intermediates, via the reference's own pieces -------------------
    covtype = ev_kw.get("covtype", "all")
    if covtype is None:
        covtype = mce.covtype
    S = int(mce.nchain[0][0])
    s1, logL, w, _ = mce.get_samples(S, istart=0, rand=False, prewhiten=False, name="s1")
    cov = mce.get_covariance() if covtype == "all" else mce.get_covariance(s=s1)
    J = cov["J"]
    X = mce.diagonalise_chain(s1.copy(), cov["eVec"], cov["eVal"])
    if ev_kw.get("pos_lnp", False):
        logL = -logL
    logLmax = float(np.amax(logL))
    fs = logL - logLmax
    kmax = mce.kmax
    if mce.split:
        s2, _, _, _ = mce.get_samples(0, istart=0, rand=False, prewhiten=False, name="s2")
        cov2 = cov if covtype == "all" else mce.get_covariance(s=s2)
        Y = mce.diagonalise_chain(s2.copy(), cov2["eVec"], cov2["eVal"])
        k0 = 0
    else:
        Y = X
        k0 = 1
    nb = NearestNeighbors(n_neighbors=kmax + 1, metric="euclidean", leaf_size=20, algorithm="auto", n_jobs=-1).fit(Y)
    DkNN, _ = nb.kneighbors(X)
    D = mce.ndim
    SumW = float(np.sum(mce.gd.data["s1"].adjusted_weights))
    lnPV = math.log(ev_kw.get("pvolume") or mce.priorvolume)
    dotp = np.zeros(kmax)
    lnE_re = np.zeros(kmax)
    for k in range(k0, kmax):
        vol = math.pi ** (D / 2) * DkNN[:, k] ** D / math.gamma(1 + D / 2)
        dotp[k] = np.dot(vol / w, np.exp(fs))
        knn = k if k0 == 1 else k + 1
        lnE_re[k] = math.log(SumW * dotp[k] / (S * knn + 1.0) * J) + logLmax - lnPV
    # the re-run must reproduce what evidence() returned
    assert np.allclose(lnE_re[1:], lnE, rtol=0, atol=1e-12), (name, lnE_re[1:], lnE)

    rows = np.linspace(0, S - 1, nrows).astype(np.int64)
    js = dict(
        name=name, chain=chain_kw, mce=mce_kw, ev=ev_kw, seed_split=seed_split,
        S=S, N_ref=int(Y.shape[0]), ndim=int(D), kmax=int(kmax), k0=k0,
        J=float(J), SumW=SumW, logLmax=logLmax, lnPriorVolume=lnPV,
        dotp=[float(x) for x in dotp], lnE=[float(x) for x in lnE],
        fit_method=str(nb._fit_method), ref_wall_s=wall,
        versions=dict(numpy=np.__version__, sklearn=__import__("sklearn").__version__),
    )
    npz = {
        name + "__rows": rows,
        name + "__X_rows": X[rows],
        name + "__DkNN_rows": DkNN[rows],
    }
    if mce.split:
        npz[name + "__s1_idx"] = np.asarray(mce.gd.data["s1"].__dict__.get("ichain", []))  # not kept by data_set
    return js, npz, mce


def split_indices(ref, chain, seed, s1frac):
    """Realised s1/s2 row indices of the reference's random split
    (``MCEvidence.py:225-226``), by replaying the same global-RNG call."""
    np.random.seed(seed)
    nrow = len(chain)
    ix = np.random.choice(range(nrow), size=int(nrow * s1frac), replace=False)
    not_ix = np.setxor1d(range(nrow), ix)
    return ix, not_ix


SMALL_CASES = [
    # name, chain kwargs, MCEvidence kwargs, evidence kwargs, split seed
    ("auto_n5000_d6_k5", dict(seed=0, n=5000, d=6), dict(kmax=5), {}, None),
    ("auto_n4000_d3_k5_intw", dict(seed=1, n=4000, d=3, weights="int"), dict(kmax=5), {}, None),
    ("auto_n5000_d6_k5_corr_intw", dict(seed=2, n=5000, d=6, weights="int", cov="corr"), dict(kmax=5), {}, None),
    ("auto_n6000_d8_k4_extra_pv", dict(seed=3, n=6000, d=5, cov="corr", nextra=3), dict(kmax=4, ndim=5, priorvolume=3.0), {}, None),
    ("auto_n5000_d6_k3_poslnp", dict(seed=4, n=5000, d=6), dict(kmax=3), dict(pos_lnp=True), None),
    ("auto_n5000_d6_k3_covsingle", dict(seed=5, n=5000, d=6, cov="corr"), dict(kmax=3), dict(covtype="single"), None),
    ("auto_n5000_d6_k3_covnone", dict(seed=6, n=5000, d=6, cov="corr"), dict(kmax=3, covtype="single"), dict(covtype=None), None),
    ("auto_n3000_d2_k2", dict(seed=7, n=3000, d=2), dict(kmax=1), {}, None),   # kmax=max(2,kmax)
    ("auto_n20000_d15_k4", dict(seed=8, n=20000, d=15, cov="corr"), dict(kmax=4), {}, None),
    ("auto_n20000_d27_k10", dict(seed=9, n=20000, d=27, cov="corr"), dict(kmax=10), {}, None),
    ("auto_n8000_d33_k6", dict(seed=10, n=8000, d=33), dict(kmax=6), {}, None),
    ("cross_n8000_d6_k4", dict(seed=11, n=8000, d=6, cov="corr", weights="int"), dict(kmax=4, split=True), {}, 101),
    ("cross_n8000_d6_k4_s1frac03", dict(seed=12, n=8000, d=6), dict(kmax=4, split=True, s1frac=0.3), {}, 102),
    ("cross_n10000_d15_k4", dict(seed=13, n=10000, d=15), dict(kmax=4, split=True), {}, 103),
    ("cross_n8000_d6_k3_covsingle", dict(seed=14, n=8000, d=6, cov="corr"), dict(kmax=3, split=True), dict(covtype="single"), 104),
    ("auto_n5000_d6_k5_pvolume_arg", dict(seed=15, n=5000, d=6), dict(kmax=5, priorvolume=2.0), dict(pvolume=7.5), None),
]

MEDIUM_CASES = [
    ("auto_n100000_d6_k4_C2", dict(seed=2, n=100_000, d=6, cov="corr"), dict(kmax=4), {}, None),
    ("auto_n100000_d6_k4_unit", dict(seed=0, n=100_000, d=6), dict(kmax=4), {}, None),
    ("auto_n100000_d27_k10_unit", dict(seed=0, n=100_000, d=27), dict(kmax=10), {}, None),
    ("auto_n50000_d15_k4", dict(seed=16, n=50_000, d=15, cov="corr"), dict(kmax=4), {}, None),
]

BIG_CASES = [
    ("auto_n1000000_d6_k4_unit", dict(seed=0, n=1_000_000, d=6), dict(kmax=4), {}, None),
    ("auto_n1000000_d27_k10_C3", dict(seed=3, n=1_000_000, d=27, cov="corr"), dict(kmax=10), {}, None),
]


# medium sizes around the symmetric sweep's automatic range (capi.hip: kSymAutoMinBlocks -- from 257 blocks of 512 rows at
# two 16-wide k-steps, d = 16..30; from 193 blocks beyond): 135 k x 27 and 140 k x 20 are inside it, 70 k x 45 (137 blocks)
# is below it and takes the exhaustive sweep unless the symmetric one is forced -- the tests run it both ways
SYM_CASES = [
    ("auto_n135000_d27_k10_corr", dict(seed=21, n=135_000, d=27, cov="corr"), dict(kmax=10), {}, None),
    ("auto_n70000_d45_k6", dict(seed=22, n=70_000, d=45), dict(kmax=6), {}, None),
    ("auto_n140000_d20_k5_corr_intw", dict(seed=23, n=140_000, d=20, cov="corr", weights="int"), dict(kmax=5), {}, None),
]


# 16 < K <= 32 neighbours at a size where the symmetric sweep is the automatic choice (two symmetric passes, round 5)
SYM2_CASES = [
    ("auto_n140000_d20_k21_corr", dict(seed=24, n=140_000, d=20, cov="corr"), dict(kmax=21), {}, None),
]


def gen_inmemory(ref, cases, tag):
    out_js, out_npz = [], {}
    for name, ckw, mkw, ekw, sseed in cases:
        t0 = time.perf_counter()
        js, npz, mce = run_case(ref, name, ckw, mkw, ekw, sseed)
        if mkw.get("split"):
            from mcevidence_amd.synth import gaussian_chain
            ix, not_ix = split_indices(ref, gaussian_chain(**ckw), sseed, mkw.get("s1frac", 0.5))
            # sanity: the replayed split reproduces the reference's s1 weights
            ch = gaussian_chain(**ckw)
            assert np.array_equal(ch[ix, 0], mce.gd.data["s1"].weights)
            npz[name + "__s1_idx"] = ix.astype(np.int64)
            npz[name + "__s2_idx"] = not_ix.astype(np.int64)
        out_js.append(js)
        out_npz.update(npz)
        print("%-36s lnE=%s  (%.1fs)" % (name, np.array2string(np.asarray(js["lnE"]), precision=8), time.perf_counter() - t0), flush=True)
    with open(os.path.join(GOLD, "evidence_%s.json" % tag), "w") as fh:
        json.dump(out_js, fh, indent=1)
    np.savez_compressed(os.path.join(GOLD, "evidence_%s.npz" % tag), **out_npz)


def gen_c4(ref, n, tag):
    """BASELINE configs[3] (C4): cross evidence of two INDEPENDENT chains of ``n`` rows each, d = 15, kmax = 4 --
    ``synth.config_chain('C4')``: the two chains stacked, s1 = the first, s2 = the second.

    The reference only knows a random split (``MCEvidence.py:221-226``: ``np.random.choice`` over the rows), so the
    harness hands it the wanted one by wrapping ``np.random.choice`` for the duration of the constructor (harness-side,
    like the sorted glob above; the reference source is untouched).  Its kNN call (``:1093-1104``, kd_tree at d = 15:
    hours at n = 1M on 8 cores) runs ONCE: a spy on ``NearestNeighbors.kneighbors`` keeps the distances it returned,
    from which ``dotp`` (``:1117``) and a few rows are recorded."""
    from sklearn.neighbors import NearestNeighbors
    from mcevidence_amd.synth import config_chain

    chain, (r1, r2) = config_chain("C4", n=n)
    kmax = 4
    seen = {}
    orig_kn, orig_choice = NearestNeighbors.kneighbors, np.random.choice

    def spy(self, X=None, *a, **k):
        out = orig_kn(self, X, *a, **k)
        seen["DkNN"], seen["fit_method"], seen["n_fit"] = out[0], str(self._fit_method), int(self.n_samples_fit_)
        return out

    def first_half(rows, size=None, replace=True, p=None):
        assert len(rows) == len(chain) and size == n and not replace
        return np.arange(n)

    NearestNeighbors.kneighbors = spy
    np.random.choice = first_half
    t0 = time.perf_counter()
    try:
        mce = ref.MCEvidence([chain], kmax=kmax, split=True, verbose=0)
    finally:
        np.random.choice = orig_choice
    try:
        lnE = mce.evidence()
    finally:
        NearestNeighbors.kneighbors = orig_kn
    wall = time.perf_counter() - t0
    assert np.array_equal(mce.gd.data["s1"].samples, chain[r1, 2:]) and np.array_equal(mce.gd.data["s2"].samples, chain[r2, 2:])
    DkNN = seen["DkNN"]
    S, D = int(mce.nchain[0][0]), int(mce.ndim)
    s1, logL, w, _ = mce.get_samples(S, istart=0, rand=False, prewhiten=False, name="s1")
    cov = mce.get_covariance()
    X = mce.diagonalise_chain(s1.copy(), cov["eVec"], cov["eVal"])
    logLmax = float(np.amax(logL))
    fs = logL - logLmax
    SumW = float(np.sum(mce.gd.data["s1"].adjusted_weights))
    dotp, lnE_re = np.zeros(kmax), np.zeros(kmax)
    for k in range(kmax):
        vol = math.pi ** (D / 2) * DkNN[:, k] ** D / math.gamma(1 + D / 2)
        dotp[k] = np.dot(vol / w, np.exp(fs))
        lnE_re[k] = math.log(SumW * dotp[k] / (S * (k + 1) + 1.0) * cov["J"]) + logLmax - math.log(mce.priorvolume)
    assert np.allclose(lnE_re[1:], lnE, rtol=0, atol=1e-12), (lnE_re, lnE)
    rows = np.linspace(0, S - 1, 64).astype(np.int64)
    name = "cross_n%d_d15_k4_C4" % n
    js = dict(name=name, config="C4", n_per_chain=n, mce=dict(kmax=kmax, split=True), ev={}, S=S, N_ref=seen["n_fit"], ndim=D,
              kmax=kmax, k0=0, J=float(cov["J"]), SumW=SumW, logLmax=logLmax, lnPriorVolume=math.log(mce.priorvolume),
              dotp=[float(x) for x in dotp], lnE=[float(x) for x in lnE], lnE_all_k=[float(x) for x in lnE_re],
              fit_method=seen["fit_method"], ref_wall_s=wall,
              versions=dict(numpy=np.__version__, sklearn=__import__("sklearn").__version__))
    with open(os.path.join(GOLD, "evidence_%s.json" % tag), "w") as fh:
        json.dump([js], fh, indent=1)
    np.savez_compressed(os.path.join(GOLD, "evidence_%s.npz" % tag), **{name + "__rows": rows, name + "__X_rows": X[rows],
                                                                      name + "__DkNN_rows": DkNN[rows]})
    print("%-36s lnE=%s fit=%s (%.1fs)" % (name, np.array2string(np.asarray(lnE), precision=10), seen["fit_method"], wall), flush=True)


def gen_c5(ref, n, tag):
    """BASELINE configs[4] (C5): auto evidence of ``synth.config_chain('C5')`` -- n rows, d = 6, kmax = 10 (ONE K = 11
    search serves the k = 2..10 sweep the config names: the reference returns ln E for k_nn = 1..kmax-1 from one call,
    ``MCEvidence.py:1107-1131, :1157``).  kd_tree at d = 6 (``:1099-1104``); a spy on ``NearestNeighbors.kneighbors`` keeps
    the distances the reference's own call returned, from which ``dotp`` (``:1117``) and a few rows are recorded."""
    from sklearn.neighbors import NearestNeighbors
    from mcevidence_amd.synth import config_chain, CONFIGS

    chain, _ = config_chain("C5", n=n)
    kmax = CONFIGS["C5"]["kmax"]
    seen = {}
    orig_kn = NearestNeighbors.kneighbors

    def spy(self, X=None, *a, **k):
        t = time.perf_counter()
        out = orig_kn(self, X, *a, **k)
        seen["DkNN"], seen["fit_method"], seen["n_fit"] = out[0], str(self._fit_method), int(self.n_samples_fit_)
        seen["knn_s"] = time.perf_counter() - t
        return out

    NearestNeighbors.kneighbors = spy
    t0 = time.perf_counter()
    try:
        mce = ref.MCEvidence([chain], kmax=kmax, verbose=0)
        lnE = mce.evidence()
    finally:
        NearestNeighbors.kneighbors = orig_kn
    wall = time.perf_counter() - t0
    DkNN = seen["DkNN"]
    S, D = int(mce.nchain[0][0]), int(mce.ndim)
    s1, logL, w, _ = mce.get_samples(S, istart=0, rand=False, prewhiten=False, name="s1")
    cov = mce.get_covariance()
    X = mce.diagonalise_chain(s1.copy(), cov["eVec"], cov["eVal"])
    logLmax = float(np.amax(logL))
    fs = logL - logLmax
    SumW = float(np.sum(mce.gd.data["s1"].adjusted_weights))
    dotp, lnE_re = np.zeros(kmax), np.zeros(kmax)
    for k in range(1, kmax):
        vol = math.pi ** (D / 2) * DkNN[:, k] ** D / math.gamma(1 + D / 2)
        dotp[k] = np.dot(vol / w, np.exp(fs))
        lnE_re[k] = math.log(SumW * dotp[k] / (S * k + 1.0) * cov["J"]) + logLmax - math.log(mce.priorvolume)
    assert np.allclose(lnE_re[1:], lnE, rtol=0, atol=1e-12), (lnE_re, lnE)
    rows = np.linspace(0, S - 1, 64).astype(np.int64)
    name = "auto_n%d_d6_k10_C5" % n
    js = dict(name=name, config="C5", n_per_chain=n, chain=dict(seed=CONFIGS["C5"]["seed"], n=n, d=6, cov="corr"), mce=dict(kmax=kmax), ev={},
              seed_split=None, S=S, N_ref=seen["n_fit"], ndim=D, kmax=kmax, k0=1, J=float(cov["J"]), SumW=SumW, logLmax=logLmax,
              lnPriorVolume=math.log(mce.priorvolume), dotp=[float(x) for x in dotp], lnE=[float(x) for x in lnE],
              fit_method=seen["fit_method"], ref_wall_s=wall, ref_kneighbors_s=seen["knn_s"], ref_cores=os.cpu_count(),
              versions=dict(numpy=np.__version__, sklearn=__import__("sklearn").__version__))
    with open(os.path.join(GOLD, "evidence_%s.json" % tag), "w") as fh:
        json.dump([js], fh, indent=1)
    np.savez_compressed(os.path.join(GOLD, "evidence_%s.npz" % tag), **{name + "__rows": rows, name + "__X_rows": X[rows],
                                                                      name + "__DkNN_rows": DkNN[rows]})
    print("%-36s lnE=%s fit=%s (%.1fs, kneighbors %.1fs)" % (name, np.array2string(np.asarray(lnE), precision=10), seen["fit_method"],
                                                          wall, seen["knn_s"]), flush=True)


def gen_host_pins(ref):
    """Host-bookkeeping pins: file loading, burn-in,
    thinning, idchain, prior volume, error behaviour."""
    from mcevidence_amd.synth import planck_like_chains, write_cosmomc_chains, gaussian_chain

    pins = {}
    with tempfile.TemporaryDirectory() as td:
        chains, names, ranges = planck_like_chains(seed=1)
        ranges2 = list(ranges)
        ranges2[3] = (ranges2[3][0], ranges2[3][1], None)          # unbounded upper ('N') entry
        ranges2.append(("fixedpar", 1.0, 1.0))                      # a fixed parameter
        root = os.path.join(td, "base_plikHM_TT_lowTEB")
        write_cosmomc_chains(root, chains, ranges)
        # --- C1 plumbing config: all 4 chains, ndim=6, kmax=2 -----------
        pi = ref.params_info(root, cosmo=True)
        pins["C1_params_info"] = dict(ndim=int(pi["ndim"]), volume=float(pi["volume"]), names=list(pi["name"]))
        pi_all = ref.params_info(root, cosmo=False)
        pins["C1_params_info_all"] = dict(ndim=int(pi_all["ndim"]), volume=float(pi_all["volume"]))
        mce = ref.MCEvidence(root, ndim=pi["ndim"], priorvolume=pi["volume"], kmax=2, verbose=0)
        lnE, info = mce.evidence(info=True)
        pins["C1_all"] = dict(lnE=[float(x) for x in lnE], N=int(mce.nsample[0]), info={k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in info.items()})
        for ic in (1, 2, 3, 4):
            m = ref.MCEvidence(root, ndim=6, priorvolume=pi["volume"], kmax=2, verbose=0, idchain=ic)
            pins["C1_chain%d" % ic] = dict(lnE=[float(x) for x in m.evidence()], N=int(m.nsample[0]))
        # --- burn-in / thinning on files ---------------------------------
        for tag, kw in (
            ("burn0.3", dict(burnlen=0.3)),
            ("burn500", dict(burnlen=500)),
            ("thin2", dict(thinlen=2)),
            ("thin5", dict(thinlen=5)),
            ("thin10", dict(thinlen=10)),
            ("burn0.2_thin3", dict(burnlen=0.2, thinlen=3)),
        ):
            m = ref.MCEvidence(root, ndim=6, priorvolume=1.0, kmax=3, verbose=0, **kw)
            d = m.gd.data["s1"]
            pins["file_" + tag] = dict(
                kw=kw, N=int(m.nsample[0]), sumw=float(np.sum(d.weights)),
                sumlike=float(np.sum(d.loglikes)), sum_p0=float(np.sum(d.samples[:, 0])),
                lnE=[float(x) for x in m.evidence()],
            )
        # poisson thinning under a fixed global seed
        np.random.seed(7)
        m = ref.MCEvidence(root, ndim=6, priorvolume=1.0, kmax=3, verbose=0, thinlen=0.5)
        d = m.gd.data["s1"]
        pins["file_thin0.5_seed7"] = dict(N=int(m.nsample[0]), sumw=float(np.sum(d.weights)), sumlike=float(np.sum(d.loglikes)), lnE=[float(x) for x in m.evidence()])
        # float weights -> weighted_thin path
        fch = [c.copy() for c in chains]
        rng = np.random.default_rng(5)
        for c in fch:
            c[:, 0] = c[:, 0] * (0.5 + rng.random(len(c)))
        rootf = os.path.join(td, "floatw")
        write_cosmomc_chains(rootf, fch, None)
        m = ref.MCEvidence(rootf, ndim=6, priorvolume=1.0, kmax=3, verbose=0, thinlen=4)
        d = m.gd.data["s1"]
        pins["file_floatw_thin4"] = dict(N=int(m.nsample[0]), sumw=float(np.sum(d.weights)), sumlike=float(np.sum(d.loglikes)), lnE=[float(x) for x in m.evidence()])
        # ranges with 'N' and a fixed parameter
        root2 = os.path.join(td, "withN")
        write_cosmomc_chains(root2, chains[:1], ranges2)
        pi2 = ref.params_info(root2, cosmo=False)
        pins["ranges_N_fixed"] = dict(ndim=int(pi2["ndim"]), volume=float(pi2["volume"]) if np.isfinite(pi2["volume"]) else "inf")
        # in-memory chains ignore burn/thin (MCEvidence.py:151)
        ch = gaussian_chain(seed=0, n=4000, d=4)
        a = ref.MCEvidence([ch], kmax=3, verbose=0).evidence()
        b = ref.MCEvidence([ch], kmax=3, verbose=0, burnlen=0.5, thinlen=3).evidence()
        pins["inmemory_ignores_burn_thin"] = dict(plain=[float(x) for x in a], with_burn_thin=[float(x) for x in b])
        # two in-memory chains are concatenated
        ch2 = gaussian_chain(seed=1, n=3000, d=4)
        pins["inmemory_two_chains"] = dict(lnE=[float(x) for x in ref.MCEvidence([ch, ch2], kmax=3, verbose=0).evidence()])
        # importance weights change SumW only (MCEvidence.py:270 vs :1117,1126)
        isf = lambda s: 0.5 * ((s[:, 0] - 0.3) / 2.0) ** 2  # noqa: E731
        pins["isfunc"] = dict(lnE=[float(x) for x in ref.MCEvidence([ch], kmax=3, verbose=0, isfunc=isf).evidence()])
        # error behaviour
        errs = {}
        for tag, fn in (
            ("dict_input", lambda: ref.MCEvidence({"a": ch}, verbose=0)),
            ("ndarray_input", lambda: ref.MCEvidence(ch, verbose=0)),
            ("thinlen1_file", lambda: ref.MCEvidence(root, ndim=6, verbose=0, thinlen=1)),
            ("thinlen_negative_file", lambda: ref.MCEvidence(root, ndim=6, verbose=0, thinlen=-2)),
            ("bscale_linear", lambda: ref.MCEvidence([ch], verbose=0, nbatch=2, brange=[100, 1000], bscale="linear")),
            ("kmax_gt_n", lambda: ref.MCEvidence([ch[:5]], kmax=10, verbose=0).evidence()),
        ):
            try:
                fn()
                errs[tag] = None
            except BaseException as e:  # noqa: BLE001
                errs[tag] = type(e).__name__
        pins["errors"] = errs
        # logpower batching (the only batching mode that works in the reference)
        m = ref.MCEvidence([ch], kmax=3, verbose=0, nbatch=3, brange=[2.5, 3.5], bscale="logpower")
        pins["batch_logpower"] = dict(nchain=m.nchain.tolist(), lnE=np.asarray(m.evidence()).tolist())
    with open(os.path.join(GOLD, "host_pins.json"), "w") as fh:
        json.dump(pins, fh, indent=1, default=str)
    for k, v in pins.items():
        print(k, v if len(str(v)) < 200 else str(v)[:200] + "...")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--medium", action="store_true")
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--sym", action="store_true")
    ap.add_argument("--sym2", action="store_true")
    ap.add_argument("--host", action="store_true")
    ap.add_argument("--c4", action="store_true", help="BASELINE configs[3] at full size: hours of kd_tree on 8 cores")
    ap.add_argument("--c4-n", type=int, default=1_000_000, help="rows per chain for --c4 (smaller: a quick harness check)")
    ap.add_argument("--c5", action="store_true", help="BASELINE configs[4] at full size: ~20 min of kd_tree + the Python volume loop")
    ap.add_argument("--c5-n", type=int, default=10_000_000, help="rows for --c5 (smaller: a quick harness check)")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    ref = import_reference()
    import logging
    logging.disable(logging.CRITICAL)
    if a.small:
        gen_inmemory(ref, SMALL_CASES, "small")
    if a.medium:
        gen_inmemory(ref, MEDIUM_CASES, "medium")
    if a.big:
        gen_inmemory(ref, BIG_CASES, "big")
    if a.sym:
        gen_inmemory(ref, SYM_CASES, "sym")
    if a.sym2:
        gen_inmemory(ref, SYM2_CASES, "sym2")
    if a.host:
        gen_host_pins(ref)
    if a.c4:
        gen_c4(ref, a.c4_n, "c4" if a.c4_n == 1_000_000 else "c4_n%d" % a.c4_n)
    if a.c5:
        gen_c5(ref, a.c5_n, "c5" if a.c5_n == 10_000_000 else "c5_n%d" % a.c5_n)


if __name__ == "__main__":
    main()
