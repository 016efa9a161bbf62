"""Host side of the drop-in (MCEvidence class, chain reader, burn/thin, prior volume),
checked against golden vectors recorded from the reference.  The kNN/reduction backend
is replaced by the CPU oracle through the `backend=` hook, so these run without a GPU."""
import logging
import math
import os

import numpy as np
import pytest

import mcevidence_amd as pkg
from mcevidence_amd.synth import gaussian_chain, planck_like_chains, write_cosmomc_chains
from helpers import LNE_TOL, OracleBackend, OracleFeedBackend, build_mce, chain_of, host_pins, load_golden

logging.disable(logging.CRITICAL)
G = load_golden()
PINS = host_pins()
SMALL = [n for n, c in G.items() if c["tag"] == "small"]


@pytest.mark.parametrize("name", SMALL)
def test_class_reproduces_reference(name):
    case = G[name]
    if case["seed_split"] is not None:
        np.random.seed(case["seed_split"])          # the reference's split uses the global RNG
    be = OracleBackend()
    mce = pkg.MCEvidence([chain_of(case)], verbose=0, backend=be, **case["mce"])
    lnE, info = mce.evidence(info=True, **case["ev"])
    assert np.allclose(lnE, case["lnE"], rtol=0, atol=LNE_TOL)
    assert len(lnE) == case["kmax"] - 1
    assert info["Nsamples_read"] == case["S"]
    # exactly one hot-path call with the sizes the reference would search
    assert len(be.calls) == 1
    assert be.calls[0]["nq"] == case["S"] and be.calls[0]["nr"] == case["N_ref"] and be.calls[0]["k0"] == case["k0"]
    if case["mce"].get("split"):
        a = case["arrays"]
        assert np.array_equal(np.asarray(mce.gd.data["s1"].ichain), a["s1_idx"])
        assert np.array_equal(np.asarray(mce.gd.data["s2"].ichain), a["s2_idx"])


def test_kmax_floor_and_signature():
    import inspect
    ch = gaussian_chain(0, 500, 3)
    m = pkg.MCEvidence([ch], kmax=1, verbose=0, backend=OracleBackend())
    assert m.kmax == 2 and len(m.evidence()) == 1
    ini = list(inspect.signature(pkg.MCEvidence.__init__).parameters)
    assert ini[:20] == ["self", "method", "ischain", "isfunc", "thinlen", "burnlen", "split", "s1frac", "shuffle", "ndim",
                        "kmax", "priorvolume", "debug", "nsample", "covtype", "nbatch", "brange", "bscale", "verbose", "args"]
    ev = list(inspect.signature(pkg.MCEvidence.evidence).parameters)
    assert ev == ["self", "verbose", "rand", "info", "covtype", "profile", "pvolume", "pos_lnp", "nproc", "prewhiten"]


def test_inmemory_quirks():
    ch = gaussian_chain(seed=0, n=4000, d=4)
    p = PINS["inmemory_ignores_burn_thin"]
    a = pkg.MCEvidence([ch], kmax=3, verbose=0, backend=OracleBackend()).evidence()
    b = pkg.MCEvidence([ch], kmax=3, verbose=0, burnlen=0.5, thinlen=3, backend=OracleBackend()).evidence()
    assert np.allclose(a, p["plain"], atol=LNE_TOL) and np.allclose(b, p["with_burn_thin"], atol=LNE_TOL)
    ch2 = gaussian_chain(seed=1, n=3000, d=4)
    c = pkg.MCEvidence([ch, ch2], kmax=3, verbose=0, backend=OracleBackend()).evidence()
    assert np.allclose(c, PINS["inmemory_two_chains"]["lnE"], atol=LNE_TOL)
    # dict input: accepted here (the reference raises TypeError on py3 by accident)
    d = pkg.MCEvidence({"a": ch, "b": ch2}, kmax=3, verbose=0, backend=OracleBackend()).evidence()
    assert np.allclose(c, d, atol=1e-12)
    isf = lambda s: 0.5 * ((s[:, 0] - 0.3) / 2.0) ** 2  # noqa: E731
    e = pkg.MCEvidence([ch], kmax=3, verbose=0, isfunc=isf, backend=OracleBackend()).evidence()
    assert np.allclose(e, PINS["isfunc"]["lnE"], atol=LNE_TOL)


def test_verbose_debug_route_same_numbers(caplog):
    """verbose > 1 takes the host-feeder route with the distances returned (debug median volume, reference
    :1143-1145) and must give the same ln E."""
    ch = gaussian_chain(seed=0, n=3000, d=4)
    quiet = pkg.MCEvidence([ch], kmax=4, verbose=0, backend=OracleBackend()).evidence()
    logging.disable(logging.NOTSET)
    try:
        with caplog.at_level(logging.DEBUG, logger="mcevidence_amd"):
            loud = pkg.MCEvidence([ch], kmax=4, verbose=2, backend=OracleBackend()).evidence()
    finally:
        logging.disable(logging.CRITICAL)
    assert np.array_equal(quiet, loud)
    assert any("median_volume" in r.getMessage() for r in caplog.records)
    assert any("ln(B)[k=1]" in r.getMessage() for r in caplog.records)


def test_evidence_many_host_logic():
    """evidence_many(): one batched backend call for the objects the feed route covers, per-object
    evidence() for the rest, results identical to evidence() one by one, per-object prior volumes."""
    be = OracleFeedBackend()
    chains = [gaussian_chain(seed=10 + i, n=900 + 50 * i, d=3 + (i % 3)) for i in range(6)]
    ms = [pkg.MCEvidence([c], kmax=3 + (i % 2), verbose=0, backend=be, priorvolume=2.0 + i) for i, c in enumerate(chains)]
    ms.append(pkg.MCEvidence([chains[0]], kmax=3, verbose=0, backend=OracleBackend()))            # no batch route
    ms.append(pkg.MCEvidence([chains[1]], kmax=3, verbose=0, backend=be, nbatch=2, brange=[2.0, 2.5], bscale="logpower"))
    np.random.seed(5)
    ms.append(pkg.MCEvidence([chains[2]], kmax=3, verbose=0, backend=be, split=True))              # cross evidence
    many = pkg.evidence_many(ms)
    assert be.batches == [7]                      # 6 autos + the split object, ONE call
    for m, got in zip(ms, many):
        assert np.array_equal(got, m.evidence())
    # the feed route agrees with the host-feeder route of the same oracle
    host = pkg.MCEvidence([chains[3]], kmax=4, verbose=0, backend=OracleBackend(), priorvolume=5.0).evidence()
    assert np.max(np.abs(many[3] - host)) < 1e-10
    # per-object pvolume override and info=True
    pv = [1.5] * len(ms)
    outs = pkg.evidence_many(ms, pvolume=pv, info=True)
    for m, (lnE, info) in zip(ms, outs):
        assert np.array_equal(lnE, m.evidence(pvolume=1.5)) and info is m.info
    with pytest.raises(ValueError):
        pkg.evidence_many(ms, pvolume=[1.0, 2.0])
    assert pkg.evidence_many([]) == []


def test_batch_logpower():
    ch = gaussian_chain(seed=0, n=4000, d=4)
    p = PINS["batch_logpower"]
    m = pkg.MCEvidence([ch], kmax=3, verbose=0, nbatch=3, brange=[2.5, 3.5], bscale="logpower", backend=OracleBackend())
    assert m.nchain.tolist() == p["nchain"]
    assert np.allclose(m.evidence(), np.array(p["lnE"]), atol=LNE_TOL)


def test_error_behaviour():
    ch = gaussian_chain(seed=0, n=400, d=3)
    with pytest.raises(TypeError):
        pkg.MCEvidence(ch, verbose=0, backend=OracleBackend())            # bare ndarray
    with pytest.raises(ValueError):                                      # K > n: sklearn's ValueError
        pkg.MCEvidence([ch[:5]], kmax=10, verbose=0, backend=OracleBackend()).evidence()
    with pytest.raises(ValueError):
        pkg.MCEvidence([ch], verbose=0, nbatch=2, brange=[100, 1000], bscale="linear", backend=OracleBackend())


@pytest.fixture(scope="module")
def planck_root(tmp_path_factory):
    td = tmp_path_factory.mktemp("chains")
    chains, names, ranges = planck_like_chains(seed=1)
    root = os.path.join(str(td), "base_plikHM_TT_lowTEB")
    write_cosmomc_chains(root, chains, ranges)
    return root, chains, ranges


def test_params_info_and_C1_plumbing(planck_root):
    """config C1 (BASELINE.json configs[0]): CosmoMC-format chains, ndim=6, kmax=2."""
    root, chains, ranges = planck_root
    pi = pkg.params_info(root, cosmo=True)
    p = PINS["C1_params_info"]
    assert pi["ndim"] == p["ndim"] == 6 and math.isclose(pi["volume"], p["volume"], rel_tol=1e-12)
    assert list(pi["name"]) == [str(x) for x in p["names"]]
    pa = pkg.params_info(root, cosmo=False)
    assert pa["ndim"] == PINS["C1_params_info_all"]["ndim"] and math.isclose(pa["volume"], PINS["C1_params_info_all"]["volume"], rel_tol=1e-12)
    m = pkg.MCEvidence(root, ndim=pi["ndim"], priorvolume=pi["volume"], kmax=2, verbose=0, backend=OracleBackend())
    lnE, info = m.evidence(info=True)
    assert m.nsample[0] == PINS["C1_all"]["N"] == 26862
    assert np.allclose(lnE, PINS["C1_all"]["lnE"], atol=1e-8)     # text round trip of the chains: a few 1e-10
    assert info["NparamsMC"] == 21 and info["NparamsCosmo"] == 6 and info["Nsamples"] == "26862"
    for ic in (1, 2, 3, 4):
        mi = pkg.MCEvidence(root, ndim=6, priorvolume=pi["volume"], kmax=2, verbose=0, idchain=ic, backend=OracleBackend())
        assert mi.nsample[0] == PINS["C1_chain%d" % ic]["N"]
        assert np.allclose(mi.evidence(), PINS["C1_chain%d" % ic]["lnE"], atol=1e-8)


@pytest.mark.parametrize("tag", ["burn0.3", "burn500", "thin2", "thin5", "thin10", "burn0.2_thin3"])
def test_file_burn_thin(planck_root, tag):
    root, _, _ = planck_root
    p = PINS["file_" + tag]
    m = pkg.MCEvidence(root, ndim=6, priorvolume=1.0, kmax=3, verbose=0, backend=OracleBackend(), **p["kw"])
    d = m.gd.data["s1"]
    assert m.nsample[0] == p["N"]
    assert math.isclose(float(np.sum(d.weights)), p["sumw"], rel_tol=1e-12)
    assert math.isclose(float(np.sum(d.loglikes)), p["sumlike"], rel_tol=1e-12)
    assert math.isclose(float(np.sum(d.samples[:, 0])), p["sum_p0"], rel_tol=1e-10)
    assert np.allclose(m.evidence(), p["lnE"], atol=1e-8)


def test_file_poisson_and_float_weight_thin(planck_root, tmp_path):
    root, chains, _ = planck_root
    p = PINS["file_thin0.5_seed7"]
    np.random.seed(7)
    m = pkg.MCEvidence(root, ndim=6, priorvolume=1.0, kmax=3, verbose=0, thinlen=0.5, backend=OracleBackend())
    assert m.nsample[0] == p["N"] and math.isclose(float(np.sum(m.gd.data["s1"].weights)), p["sumw"])
    assert np.allclose(m.evidence(), p["lnE"], atol=1e-8)
    fch = [c.copy() for c in chains]
    rng = np.random.default_rng(5)
    for c in fch:
        c[:, 0] = c[:, 0] * (0.5 + rng.random(len(c)))
    rootf = os.path.join(str(tmp_path), "floatw")
    write_cosmomc_chains(rootf, fch, None)
    p = PINS["file_floatw_thin4"]
    m = pkg.MCEvidence(rootf, ndim=6, priorvolume=1.0, kmax=3, verbose=0, thinlen=4, backend=OracleBackend())
    assert m.nsample[0] == p["N"] and math.isclose(float(np.sum(m.gd.data["s1"].weights)), p["sumw"], rel_tol=1e-9)
    assert np.allclose(m.evidence(), p["lnE"], atol=1e-8)
    with pytest.raises(ValueError):
        pkg.MCEvidence(root, ndim=6, verbose=0, thinlen=-2, backend=OracleBackend())
    # thinlen == 1: no-op here (the reference raises TypeError by accident)
    m1 = pkg.MCEvidence(root, ndim=6, verbose=0, thinlen=1, backend=OracleBackend())
    assert m1.nsample[0] == 26862


def test_ranges_unbounded_and_fixed(tmp_path):
    chains, _, ranges = planck_like_chains(seed=1)
    ranges = list(ranges)
    ranges[3] = (ranges[3][0], ranges[3][1], None)
    ranges.append(("fixedpar", 1.0, 1.0))
    root = os.path.join(str(tmp_path), "withN")
    write_cosmomc_chains(root, chains[:1], ranges)
    pi = pkg.params_info(root, cosmo=False)
    assert pi["ndim"] == PINS["ranges_N_fixed"]["ndim"] == 21 and np.isinf(pi["volume"])
    with pytest.raises(Exception):
        pkg.params_info(os.path.join(str(tmp_path), "nothing_here"))


def test_montepython_log_param(tmp_path):
    d = tmp_path / "mp"
    d.mkdir()
    (d / "log.param").write_text(
        "data.parameters['omega_b'] = [2.2, 1.8, 3.0, 0.03, 0.01, 'cosmo']\n"
        "data.parameters['h'] = [0.7, 0.5, 0.9, 0.01, 1, 'cosmo']\n"
        "data.parameters['A_nuis'] = [1.0, 0.0, 2.0, 0.1, 1, 'nuisance']\n"
        "data.parameters['sigma8'] = [0, None, None, 0, 1, 'derived']\n"
        "#data.parameters['commented'] = [0, 0, 1, 0, 1, 'cosmo']\n")
    pc = pkg.params_info(str(d), cosmo=True)
    assert pc["name"] == ["omega_b", "h"] and math.isclose(pc["volume"], 1.2 * 0.4)
    pa = pkg.params_info(str(d), cosmo=False)
    assert pa["ndim"] == 3 and math.isclose(pa["volume"], 1.2 * 0.4 * 2.0)


def test_cli_runs_cross(planck_root, monkeypatch, capsys):
    from mcevidence_amd import cli, evidence
    root, _, _ = planck_root
    monkeypatch.setattr(evidence, "HipBackend", lambda *a, **k: OracleBackend())
    np.random.seed(3)
    out = cli.main([root, "-k", "3", "--cross", "-vb", "0"])
    assert out.shape == (2,) and np.all(np.isfinite(out))
    assert "Using file" in capsys.readouterr().out


def test_ndim_is_clamped_once_and_both_routes_agree():
    """ndim beyond the parameter columns means "all of them" (the reference slices s[:, 0:ndim]); ndim below them
    cuts the leading columns.  The device-feeder route (ndim goes to the library) and the host route (NumPy
    whitening, then the hot path) must see the same number and give the same ln E."""
    ch = gaussian_chain(seed=4, n=3000, d=5, cov="corr")
    for ndim, used in ((3, 3), (5, 5), (9, 5)):
        feed = pkg.MCEvidence([ch], ndim=ndim, kmax=4, verbose=0, backend=OracleFeedBackend())
        host = pkg.MCEvidence([ch], ndim=ndim, kmax=4, verbose=0, backend=OracleBackend())
        assert feed.ndim == host.ndim == used and feed.info["NparamsCosmo"] == used
        a, b = feed.evidence(), host.evidence()
        assert feed.backend.calls[0]["d"] == host.backend.calls[0]["d"] == used
        assert np.allclose(a, b, rtol=0, atol=LNE_TOL)
    with pytest.raises(ValueError):
        pkg.MCEvidence([ch], ndim=0, verbose=0, backend=OracleBackend())


def test_missing_native_reader_falls_back_to_loadtxt(tmp_path, monkeypatch):
    """an install without libmcechains.so still reads chain files (np.loadtxt, the reference's reader), with a warning"""
    from mcevidence_amd import chain_io, chains
    ch = gaussian_chain(seed=2, n=300, d=3)
    root = str(tmp_path / "c")
    np.savetxt(root + "_1.txt", ch[:150])
    np.savetxt(root + "_2.txt", ch[150:])
    want = pkg.MCEvidence(root, kmax=3, verbose=0, backend=OracleBackend()).evidence()
    monkeypatch.setattr(chain_io, "LIB_PATH", str(tmp_path / "nowhere.so"))
    monkeypatch.setattr(chains, "_NATIVE_READER", None)
    with pytest.warns(RuntimeWarning, match="libmcechains|nowhere"):
        got = pkg.MCEvidence(root, kmax=3, verbose=0, backend=OracleBackend()).evidence()
    assert np.array_equal(want, got)
    monkeypatch.setattr(chains, "_NATIVE_READER", None)


def test_class_with_an_explicit_pair_reproduces_the_reference_C4_shape():
    """``MCEvidence.set_split``: a caller-chosen (s1, s2) pair -- config C4's two independent chains -- instead of the
    reference's random split; the reference's own ln E for that pair (reduced-size golden) through the host route and
    through the device-feeder route's CPU double."""
    case = G["cross_n20000_d15_k4_C4"]
    for be in (OracleBackend(), OracleFeedBackend()):
        mce = build_mce(case, backend=be)
        assert mce.split and mce.nsample == [20000, 20000] and mce.snames == ["s1", "s2"]
        lnE = mce.evidence()
        assert np.allclose(lnE, case["lnE"], rtol=0, atol=LNE_TOL)
        assert be.calls[0]["nq"] == 20000 and be.calls[0]["nr"] == 20000 and be.calls[0]["k0"] == 0
