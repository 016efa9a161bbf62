"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain NumPy restatement of the reference's kNN evidence hot path
(``/root/reference/MCEvidence.py:1041-1168``).  It exists so the HIP path can
be checked against something that runs anywhere; it is never imported by the
product package (``mcevidence_amd``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against golden vectors produced by importing the reference itself in the
build container (``oracle/gen_golden.py`` -> ``tests/golden/*.json|npz``).

Third-party arithmetic on the path: scikit-learn ``NearestNeighbors``
(unpinned in the reference's setup.py:34; 1.7.2 in this image) -- KDTree for
D<=15, chunked GEMM-form brute force for D>15.  ``knn_sklearn`` below is that
exact call (``MCEvidence.py:1093-1104``); ``knn_brute`` is an independent
exact direct-difference search (C + OpenMP, ``oracle/knn_brute.c``) used for
row-level parity.
"""
from __future__ import annotations

import ctypes
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# feeders (host side in the reference too)
# --------------------------------------------------------------------------
def covariance_eig(samples):
    """``get_covariance`` (MCEvidence.py:851-882): UNWEIGHTED np.cov, np.linalg.eig,
    J = sqrt(det) -- or J=1 / posdef=False if an eigenvalue is negative."""
    cov = np.cov(samples.T)
    cov = np.atleast_2d(cov)
    eval_, evec = np.linalg.eig(cov)
    if (eval_ < 0).any():
        return dict(cov=cov, posdef=False, J=1, eVec=evec, eVal=eval_)
    return dict(cov=cov, posdef=True, J=math.sqrt(np.linalg.det(cov)), eVec=evec, eVal=eval_)


def whiten(samples, evec, eval_):
    """``diagonalise_chain`` (MCEvidence.py:842-849)."""
    s = np.dot(samples, evec)
    return s / np.sqrt(eval_)[None, :]


# --------------------------------------------------------------------------
# a1/a2: the neighbour search
# --------------------------------------------------------------------------
def knn_sklearn(X, Y, K, n_jobs=-1, algorithm="auto"):
    """The reference's exact third-party call (MCEvidence.py:1093-1104)."""
    from sklearn.neighbors import NearestNeighbors

    nb = NearestNeighbors(n_neighbors=K, metric="euclidean", leaf_size=20, algorithm=algorithm, n_jobs=n_jobs).fit(Y)
    d, i = nb.kneighbors(X)
    return d, i


_lib = None


def _load_c():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle_knn.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle C library not built: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.oracle_knn_f64.restype = ctypes.c_int
        lib.oracle_knn_f64.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
            ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
        ]
        lib.oracle_dotp_f64.restype = ctypes.c_int
        lib.oracle_dotp_f64.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ]
        _lib = lib
    return _lib


def knn_brute(X, Y, K, self_mode=0, self_offset=0, nthreads=0):
    """Exact brute force in C: d^2 = sum_i (x_i - y_i)^2 (no GEMM trick), ties
    broken by smaller reference index.  self_mode as in include/mcevidence_hip.h:
    0 = plain search, 2 = skip reference row (self_offset + query row)."""
    lib = _load_c()
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    nq, d = X.shape
    nr = Y.shape[0]
    dist = np.empty((nq, K), dtype=np.float64)
    idx = np.empty((nq, K), dtype=np.int64)
    rc = lib.oracle_knn_f64(X.ctypes.data, nq, Y.ctypes.data, nr, d, K, self_mode, self_offset,
                            dist.ctypes.data, idx.ctypes.data, nthreads)
    if rc != 0:
        raise ValueError("oracle_knn_f64 failed rc=%d (K > number of usable reference rows?)" % rc)
    return dist, idx


def knn_numpy(X, Y, K, exclude_self=False, chunk=512):
    """Pure NumPy exact direct-difference search for SMALL inputs."""
    nq = X.shape[0]
    dist = np.empty((nq, K))
    idx = np.empty((nq, K), dtype=np.int64)
    for s in range(0, nq, chunk):
        xs = X[s:s + chunk]
        d2 = ((xs[:, None, :] - Y[None, :, :]) ** 2).sum(-1)
        if exclude_self:
            r = np.arange(xs.shape[0])
            d2[r, s + r] = np.inf
        order = np.argsort(d2, axis=1, kind="stable")[:, :K]
        idx[s:s + chunk] = order
        dist[s:s + chunk] = np.sqrt(np.take_along_axis(d2, order, axis=1))
    return dist, idx


# --------------------------------------------------------------------------
# a3/a4: volume + weighted reduction
# --------------------------------------------------------------------------
def dotp_literal(DkNN, weight, fs, ndim, k0, kmax):
    """Literal restatement of MCEvidence.py:1107-1117 (pow / gamma / np.dot),
    vectorised over j.  Note py3 true division in ndim/2."""
    out = np.zeros(kmax)
    efs = np.exp(fs)
    for k in range(k0, kmax):
        vol = math.pow(math.pi, ndim / 2) * np.power(DkNN[:, k], ndim) / math.gamma(1 + ndim / 2)
        out[k] = np.dot(vol / weight, efs)
    return out


def ln_unit_ball(ndim):
    """ln of the volume of the unit ndim-ball, ln(pi^(D/2)/Gamma(1+D/2))."""
    return 0.5 * ndim * math.log(math.pi) - math.lgamma(1.0 + 0.5 * ndim)


def dotp_logdomain(DkNN, weight, fs, ndim, k0, kmax):
    """Same sum in the log domain -- the form the HIP reduction kernel uses:
    dotp_k = sum_j sign(w_j) exp(lnC_D + D ln r_jk - ln |w_j| + fs_j)  (the sign keeps the reference's
    volume/weight for a negative weight; fs = -inf and r = 0 give zero terms)."""
    out = np.zeros(kmax)
    lnc = ln_unit_ball(ndim)
    weight = np.asarray(weight, dtype=np.float64)
    with np.errstate(divide="ignore"):
        base = fs - np.log(np.abs(weight))
        sgn = np.where(weight < 0, -1.0, 1.0)
        for k in range(k0, kmax):
            out[k] = np.sum(sgn * np.exp(lnc + ndim * np.log(DkNN[:, k]) + base))
    return out


def dotp_c(DkNN, weight, fs, ndim, k0, kmax):
    """C restatement of the same reduction (serial, fixed order)."""
    lib = _load_c()
    DkNN = np.ascontiguousarray(DkNN, dtype=np.float64)
    weight = np.ascontiguousarray(weight, dtype=np.float64)
    fs = np.ascontiguousarray(fs, dtype=np.float64)
    out = np.zeros(kmax)
    rc = lib.oracle_dotp_f64(DkNN.ctypes.data, DkNN.shape[0], DkNN.shape[1], k0, kmax, ndim,
                             weight.ctypes.data, fs.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise ValueError("oracle_dotp_f64 rc=%d" % rc)
    return out


# --------------------------------------------------------------------------
# a5: MLE assembly
# --------------------------------------------------------------------------
def mle_from_dotp(dotp, S, k0, kmax, SumW, J, logLmax, lnPriorVolume):
    """MCEvidence.py:1119-1131 and the [1:] slice of :1156-1157."""
    mle = np.zeros(kmax)
    for k in range(k0, kmax):
        k_nn = k if k0 == 1 else k + 1
        amax = dotp[k] / (S * k_nn + 1.0)
        mle[k] = math.log(SumW * amax * J) + logLmax - lnPriorVolume
    return mle[1:]


# --------------------------------------------------------------------------
# the whole path, from chain arrays
# --------------------------------------------------------------------------
def evidence_from_chain(chain, ndim=None, kmax=5, priorvolume=1.0, pos_lnp=False, covtype="all",
                        s1_idx=None, s2_idx=None, adjusted_weights=None, knn="sklearn"):
    """Restatement of MCEvidence([chain]).evidence() for an in-memory chain
    (columns: weight, -lnL, params...).  Cross-evidence when s1_idx/s2_idx (the
    realised random split, MCEvidence.py:225-226) are given.
    Returns dict(lnE, dotp, DkNN, X, Y, J, SumW, logLmax, S, k0)."""
    kmax = max(2, kmax)                                  # MCEvidence.py:694
    w_all, nll_all, th_all = chain[:, 0], chain[:, 1], chain[:, 2:]
    if ndim is None:
        ndim = th_all.shape[1]
    split = s1_idx is not None
    if split:
        s1 = th_all[s1_idx][:, :ndim]
        s2 = th_all[s2_idx][:, :ndim]
        w, lnp = w_all[s1_idx], -nll_all[s1_idx]
        allrows = np.concatenate((s1, s2))               # all_sample_arrays order, :413
    else:
        s1 = th_all[:, :ndim]
        s2 = None
        w, lnp = w_all, -nll_all                         # arrays(): lnp = -loglikes, :399
        allrows = s1
    if covtype == "all":
        cs = covariance_eig(allrows)
        cs2 = cs
    else:  # 'single': s1's own eigen-system; s2 whitened with ITS own, J stays s1's (:1052-1086)
        cs = covariance_eig(s1)
        cs2 = covariance_eig(s2) if split else cs
    J = cs["J"]
    X = whiten(s1, cs["eVec"], cs["eVal"])
    logL = -lnp if pos_lnp else lnp
    logLmax = float(np.amax(logL))
    fs = logL - logLmax
    if split:
        Y = whiten(s2, cs2["eVec"], cs2["eVal"])
        k0 = 0
    else:
        Y = X
        k0 = 1
    K = kmax + 1
    if knn == "sklearn":
        DkNN, _ = knn_sklearn(X, Y, K)
    elif knn == "brute":
        DkNN, _ = knn_brute(X, Y, K)
    else:
        DkNN, _ = knn(X, Y, K)
    S = X.shape[0]
    dotp = dotp_literal(DkNN, w, fs, ndim, k0, kmax)
    SumW = float(np.sum(w if adjusted_weights is None else adjusted_weights))
    lnE = mle_from_dotp(dotp, S, k0, kmax, SumW, J, logLmax, math.log(priorvolume))
    return dict(lnE=lnE, dotp=dotp, DkNN=DkNN, X=X, Y=Y, J=J, SumW=SumW, logLmax=logLmax, S=S, k0=k0,
                w=w, fs=fs, ndim=ndim, kmax=kmax)
