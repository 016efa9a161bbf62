"""Shared test helpers (tests only; the oracle is never imported by the product)."""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (REPO, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import oracle_np as orc  # noqa: E402
from mcevidence_amd.synth import gaussian_chain  # noqa: E402

GOLD = os.path.join(HERE, "golden")

#: stated fp64 tolerance on ln E (|dlnE| <= 1e-9)
LNE_TOL = 1e-9
#: row-level tolerance on kNN distances (GEMM-form fp64 vs exact differences)
DIST_RTOL = 1e-10


def load_golden():
    out = {}
    for tag in ("small", "medium", "big", "sym", "sym2", "c4_n20000", "c4", "c5_n200000", "c5"):
        jp = os.path.join(GOLD, "evidence_%s.json" % tag)
        if not os.path.exists(jp):
            continue
        arrays = np.load(os.path.join(GOLD, "evidence_%s.npz" % tag))
        for case in json.load(open(jp)):
            case = dict(case)
            case["tag"] = tag
            case.setdefault("seed_split", None)          # (config cases -- C4 -- carry an explicit split instead: explicit_split_of)
            case["arrays"] = {k[len(case["name"]) + 2:]: arrays[k] for k in arrays.files if k.startswith(case["name"] + "__")}
            out[case["name"]] = case
    return out


def host_pins():
    return json.load(open(os.path.join(GOLD, "host_pins.json")))


def chain_of(case):
    if case.get("config"):                      # a BASELINE.json config recipe (synth.config_chain): C4 = two chains stacked
        from mcevidence_amd.synth import config_chain
        return config_chain(case["config"], n=case.get("n_per_chain"))[0]
    return gaussian_chain(**case["chain"])


def explicit_split_of(case):
    """(s1_rows, s2_rows) of a config case with a caller-chosen split, else None"""
    if case.get("config"):
        from mcevidence_amd.synth import config_chain
        r1, r2 = config_chain(case["config"], n=case.get("n_per_chain"))[1]
        return None if r1 is None else (r1, r2)
    return None


class OracleBackend(object):
    """Test double for ``HipBackend``: the same interface served by the CPU oracle, so the
    host bookkeeping of ``MCEvidence`` can be pinned against the goldens without a GPU."""

    name = "oracle"

    def __init__(self, knn="brute"):
        self.knn = knn
        self.calls = []

    def knn_dotp(self, X, Y, weight, fs, kmax, k0, want_dist=False):
        ref = X if Y is None else Y
        K = kmax - k0
        if self.knn == "sklearn":
            d, _ = orc.knn_sklearn(X, ref, kmax + 1)
            d = d[:, k0:kmax]
        else:
            d, _ = orc.knn_brute(X, ref, K, self_mode=2 if k0 == 1 else 0)
        ndim = X.shape[1]
        full = np.zeros((X.shape[0], kmax))
        full[:, k0:] = d
        dotp = orc.dotp_literal(full, weight, fs, ndim, k0, kmax)
        self.calls.append(dict(nq=X.shape[0], nr=ref.shape[0], d=ndim, kmax=kmax, k0=k0))
        return dotp, (d if want_dist else None)


class OracleFeedBackend(OracleBackend):
    """OracleBackend that also offers the device-feeder routes (``evidence_feed`` and its batched
    form) -- NumPy covariance/eigen-system/whitening in front of the same oracle search."""

    def evidence_feed(self, S1, S2, ndim, cov_mode, kmax, weight, fs):
        s1 = np.asarray(S1)[:, :ndim]
        s2 = None if S2 is None else np.asarray(S2)[:, :ndim]

        def eig(rows):
            ev, U = np.linalg.eigh(np.atleast_2d(np.cov(rows.T)))
            if (ev <= 0).any():
                raise ValueError("math domain error")
            # the library's documented canonical form (include/mcevidence_hip.h): eigenvalues
            # descending, largest component of each eigenvector positive
            ev, U = ev[::-1], U[:, ::-1]
            sgn = np.sign(U[np.argmax(np.abs(U), axis=0), np.arange(U.shape[1])])
            return ev, U * sgn
        if cov_mode == 0:
            ev, U = eig(s1 if s2 is None else np.concatenate([s1, s2]))
            ev2, U2 = ev, U
        else:
            ev, U = eig(s1)
            ev2, U2 = (ev, U) if s2 is None else eig(s2)
        X = (s1 @ U) / np.sqrt(ev)
        Y = None if s2 is None else (s2 @ U2) / np.sqrt(ev2)
        dotp, _ = self.knn_dotp(X, Y, weight, fs, kmax, 0 if s2 is not None else 1)
        return dotp, math.sqrt(float(np.prod(ev)))

    def evidence_feed_batch(self, problems):
        self.batches = getattr(self, "batches", []) + [len(problems)]
        return [self.evidence_feed(*p) for p in problems]


def lnE_from_dotp(case, dotp):
    return orc.mle_from_dotp(np.asarray(dotp), case["S"], case["k0"], case["kmax"], case["SumW"], case["J"],
                             case["logLmax"], case["lnPriorVolume"])


def build_mce(case, **kw):
    """The drop-in class set up as the reference was when it produced ``case``: the global RNG seeded before a random
    split, or the explicit split of a config case (C4: two independent chains stacked, s1 = the first)."""
    import mcevidence_amd as pkg
    if case["seed_split"] is not None:
        np.random.seed(case["seed_split"])
    mk = dict(case["mce"])
    split = explicit_split_of(case)
    if split is not None:
        mk.pop("split", None)
    mce = pkg.MCEvidence([chain_of(case)], verbose=0, **mk, **kw)
    if split is not None:
        mce.set_split(*split)
    return mce
