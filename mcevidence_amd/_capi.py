"""ctypes binding of ``libmcevidence_hip.so`` (C ABI: ``include/mcevidence_hip.h``).

This is the only place the package touches native code.  There is no CPU
fallback: if the shared library is missing, or no MI355X is visible when a
compute entry point is called, an exception is raised.

Error mapping follows what the replaced scikit-learn call would raise
(``/root/reference/MCEvidence.py:1093-1104``): argument problems ->
``ValueError`` (sklearn raises ValueError for ``n_neighbors > n_samples_fit``),
runtime/device problems -> ``RuntimeError``.
"""
from __future__ import annotations

import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MCE_LIB: another build of the same library -- same-box A/B of kernel variants, tools/; never a different backend)
LIB_PATH = os.environ.get("MCE_LIB") or os.path.join(_HERE, "libmcevidence_hip.so")

MCE_OK = 0
MCE_ERR_INVALID = -1
MCE_ERR_K_RANGE = -2
MCE_ERR_HIP = -3
MCE_ERR_NO_DEVICE = -4
MCE_ERR_WORKSPACE = -5
MCE_ERR_DIM_RANGE = -6
MCE_ERR_VERIFY = -7
MCE_MAX_K = 32
MCE_MAX_DIM = 63
SELF_NONE, SELF_INCLUDE, SELF_EXCLUDE = 0, 1, 2

#: every symbol include/mcevidence_hip.h declares: name -> (restype, argtypes)
_c = ctypes
_P = _c.c_void_p
SIGNATURES = {
    "mce_abi_version": (_c.c_int, []),
    "mce_device_count": (_c.c_int, []),
    "mce_last_error": (_c.c_char_p, []),
    "mce_last_kernel": (_c.c_char_p, []),
    "mce_last_verify_rows": (_c.c_int32, []),
    "mce_source_hash": (_c.c_char_p, []),
    "mce_release_device_memory": (None, []),
    "mce_set_search_mode": (_c.c_int, [_c.c_int]),
    "mce_get_search_mode": (_c.c_int, []),
    "mce_set_prune_mode": (_c.c_int, [_c.c_int]),
    "mce_get_prune_mode": (_c.c_int, []),
    "mce_set_sym_mode": (_c.c_int, [_c.c_int]),
    "mce_get_sym_mode": (_c.c_int, []),
    "mce_last_prune_stats": (_c.c_int, [_c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
    "mce_set_profiling": (None, [_c.c_int]),
    "mce_last_kernel_ms": (_c.c_double, []),
    "mce_last_search_stats": (_c.c_int, [_c.c_void_p, _c.c_int32]),
    "mce_debug_mfma_tile_f16": (_c.c_int, [_P, _P, _c.c_int32, _P, _c.c_int32]),
    "mce_debug_mfma_tiles_f16": (_c.c_int, [_P, _P, _c.c_int32, _c.c_int32, _P, _c.c_int32]),
    "mce_options_push": (_c.c_int, [_P]),
    "mce_options_pop": (_c.c_int, []),
    "mce_knn_f64_opt": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int64, _P, _P, _c.c_int32, _P]),
    "mce_knn_dotp_f64_opt": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int64, _P, _P, _P, _P, _P, _c.c_int32, _P]),
    "mce_knn_dotp_f64_dev_opt": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int64, _P, _P, _P, _P, _P, _c.c_size_t, _P, _P]),
    "mce_knn_workspace_bytes_opt": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _P]),
    "mce_knn_f64": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int64, _P, _P, _c.c_int32]),
    "mce_dotp_f64": (_c.c_int, [_P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _P, _P, _P, _c.c_int32]),
    "mce_knn_dotp_f64": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int64, _P, _P, _P, _P, _P, _c.c_int32]),
    "mce_evidence_feed_f64": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32,
                                         _P, _P, _P, _P, _P, _c.c_int32]),
    "mce_evidence_feed_part_f64": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32,
                                              _P, _P, _c.c_int32, _c.c_int32, _P, _P, _P, _P, _c.c_int32]),
    "mce_evidence_feed_whiten_f64": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _P, _P, _P, _P, _P, _c.POINTER(_c.c_double), _P,
                                                _c.POINTER(_c.c_uint64), _c.c_int32]),
    "mce_evidence_feed_part_dev_f64": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32,
                                                  _P, _P, _c.c_int32, _c.c_int32, _P, _P, _P, _P, _c.c_int32]),
    "mce_evidence_feed_whiten_dev_f64": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32, _P, _P, _P, _P, _P, _c.POINTER(_c.c_double), _P,
                                                    _c.POINTER(_c.c_uint64), _c.c_int32]),
    "mce_knn_dotp_part_f64_dev": (_c.c_int, [_P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _P, _P, _P, _P, _c.c_size_t, _P]),
    "mce_knn_dotp_part_f64": (_c.c_int, [_P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _P, _P, _P, _c.c_int32]),
    "mce_knn_dotp_part_prepared_f64_dev": (_c.c_int, [_P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _P, _P, _P, _P, _c.c_size_t, _P]),
    "mce_prune_part_applies": (_c.c_int32, [_c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32]),
    "mce_prune_part_prepare_dev": (_c.c_int, [_P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_int64),
                                              _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64), _P, _c.c_size_t, _P]),
    "mce_pairs_once_blocks": (_c.c_int32, [_c.c_int64, _c.c_int32, _c.c_int32]),
    "mce_pairs_once_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32]),
    "mce_pairs_once_prepare_dev": (_c.c_int, [_P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_int64), _P,
                                              _c.c_size_t, _P]),
    "mce_pairs_once_sweep_dev": (_c.c_int, [_P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _P, _P, _P, _c.c_size_t, _P]),
    "mce_pairs_once_export_dev": (_c.c_int, [_c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _P, _P, _c.c_size_t, _P]),
    "mce_pairs_once_finish_dev": (_c.c_int, [_P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _P, _P, _P, _c.c_int64, _P, _P, _P,
                                             _c.c_size_t, _P]),
    "mce_verify_workspace_bytes": (_c.c_size_t, [_c.c_int32, _c.c_int32]),
    "mce_verify_knn_f64_dev": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int64, _P, _c.c_int32, _c.c_int32,
                                          _c.c_uint64, _P, _P, _c.c_size_t, _P]),
    "mce_verify_knn_f64": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int64, _P, _c.c_int32, _c.c_int32,
                                      _c.c_uint64, _P, _c.c_int32]),
    "mce_knn_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int32, _c.c_int32]),
    "mce_dotp_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int32]),
    "mce_knn_f64_dev": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int64, _P, _P, _P, _c.c_size_t, _P]),
    "mce_dotp_f64_dev": (_c.c_int, [_P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _P, _P, _P, _P, _c.c_size_t, _P]),
    "mce_knn_dotp_f64_dev": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int64, _P, _P, _P, _P, _P, _c.c_size_t, _P]),
}



class FeedProblem(ctypes.Structure):
    """``mce_feed_problem`` of include/mcevidence_hip.h (one evidence problem of a batch)."""
    _fields_ = [("S1", _P), ("n1", _c.c_int64), ("ld1", _c.c_int64),
                ("S2", _P), ("n2", _c.c_int64), ("ld2", _c.c_int64),
                ("d", _c.c_int32), ("cov_mode", _c.c_int32), ("kmax", _c.c_int32), ("status", _c.c_int32),
                ("w", _P), ("fs", _P), ("dotp", _P), ("eigenvalues", _P), ("jacobian", _c.c_double)]


SIGNATURES["mce_evidence_feed_batch_f64"] = (_c.c_int, [_c.POINTER(FeedProblem), _c.c_int64, _P, _c.c_int32])
SIGNATURES["mce_feed_problem_size"] = (_c.c_size_t, [])

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is None:
        path = LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                "mcevidence_amd: %s not found -- build it with `make -C mcevidence_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % path)
        # PyTorch-ROCm wheels bundle their own HIP/HSA runtime (torch/lib/libamdhip64.so, same
        # soname as /opt/rocm's).  Two HSA runtimes in one process cannot both own the GPU, so
        # when torch is installed let it load its runtime FIRST; our library then binds to the
        # already-loaded libamdhip64.so.7.  torch is plumbing here, never a compute fallback.
        if "torch" not in sys.modules and os.environ.get("MCE_SKIP_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:  # torch absent: plain ROCm runtime from /opt/rocm
                pass
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("MCE_LIB") and not hasattr(lib, name):
                continue                                         # (an older build in an A/B run lacks the newest entry points)
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.mce_abi_version() not in (2, 3) or lib.mce_feed_problem_size() != ctypes.sizeof(FeedProblem):
            raise RuntimeError("mcevidence_amd: ABI version mismatch")
        _lib = lib
    return _lib


def last_error():
    return load().mce_last_error().decode("utf-8", "replace")


def last_kernel():
    return load().mce_last_kernel().decode("utf-8", "replace")


def source_hash():
    """SHA-256 of the kernel sources the loaded library was built from (csrc/Makefile: src_hash.h)"""
    return load().mce_source_hash().decode("ascii")


MODE_AUTO, MODE_F64, MODE_F16_FILTER = 0, 1, 2


def set_search_mode(mode):
    """0/2: fp16-MFMA filter + exact fp64 refine where supported; 1: fp64 MFMA sweep only."""
    check(load().mce_set_search_mode(int(mode)))


def get_search_mode():
    return int(load().mce_get_search_mode())


PRUNE_AUTO, PRUNE_OFF, PRUNE_FORCE = 0, 1, 2


def set_prune_mode(mode):
    """spatial pruning of the search (low d, large reference sets): 0 auto, 1 never, 2 whenever possible"""
    check(load().mce_set_prune_mode(int(mode)))


def get_prune_mode():
    return int(load().mce_get_prune_mode())


SYM_AUTO, SYM_OFF, SYM_FORCE = 0, 1, 2


def set_sym_mode(mode):
    """symmetric sweep of auto-evidence searches (each pair of rows multiplied once): 0 auto, 1 never, 2 whenever possible"""
    check(load().mce_set_sym_mode(int(mode)))


def get_sym_mode():
    return int(load().mce_get_sym_mode())


def last_prune_stats():
    """(fraction of block x chunk pairs staged, fraction of wave x tile products multiplied) of the last
    pruned search run through a ``*_dev`` call on this thread (its workspace must still exist)."""
    a, b = _c.c_double(), _c.c_double()
    check(load().mce_last_prune_stats(ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def release_device_memory():
    load().mce_release_device_memory()


def set_profiling(on):
    load().mce_set_profiling(1 if on else 0)


def last_kernel_ms():
    return float(load().mce_last_kernel_ms())


class Options(_c.Structure):
    """``mce_options``: the search / prune / symmetric modes of ONE call (-1: the process default); ``verify`` > 0:
    re-check that many query rows after the search by an exact fp64 scan of all reference rows (RuntimeError if one
    disagrees; host-pointer entry points)."""
    _fields_ = [("size", _c.c_int32), ("search_mode", _c.c_int32), ("prune_mode", _c.c_int32), ("sym_mode", _c.c_int32),
                ("same_set", _c.c_int32), ("verify", _c.c_int32), ("reserved", _c.c_int32 * 2)]

    def __init__(self, search_mode=-1, prune_mode=-1, sym_mode=-1, same_set=-1, verify=-1):
        super().__init__(_c.sizeof(Options), int(search_mode), int(prune_mode), int(sym_mode), int(same_set), int(verify))


class options(object):
    """``with _capi.options(sym_mode=_capi.SYM_FORCE): ...`` -- the modes apply to the library calls THIS THREAD makes
    inside the block (mce_options_push / mce_options_pop); other threads are not affected."""

    def __init__(self, **modes):
        self.opt = Options(**modes)

    def __enter__(self):
        check(load().mce_options_push(_c.byref(self.opt)))
        return self

    def __exit__(self, *exc):
        check(load().mce_options_pop())
        return False


def last_search_stats():
    """dict(flops_main, flops_all, search_ms, kernel_ms) of the last search on this thread (mce_last_search_stats)."""
    out = (_c.c_double * 4)()
    check(load().mce_last_search_stats(_c.cast(out, _c.c_void_p), 4))
    return dict(flops_main=out[0], flops_all=out[1], search_ms=out[2], kernel_ms=out[3])


def check(rc):
    if rc == MCE_OK:
        return
    msg = last_error()
    if rc in (MCE_ERR_INVALID, MCE_ERR_K_RANGE, MCE_ERR_WORKSPACE, MCE_ERR_DIM_RANGE):
        raise ValueError(msg)
    raise RuntimeError("mcevidence_amd HIP backend: " + msg)


def device_count():
    return int(load().mce_device_count())


def require_device():
    n = device_count()
    if n < 1:
        raise RuntimeError("mcevidence_amd: no HIP device visible (this package has no CPU fallback)")
    return n


def _f64(a, name):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if not np.all(np.isfinite(a)):
        raise ValueError("%s contains NaN or infinity" % name)   # sklearn's check_array does the same
    return a


def _f64_fs(a, name="fs"):
    """fs = logL - max(logL) <= 0 (reference :1062-1064).  -inf is a legitimate entry -- a row with zero
    likelihood, exp(fs) = 0, a harmless zero term in the reference's sum and in the kernel's -- so only NaN
    and +inf are refused."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if np.isnan(a).any() or (a == np.inf).any():
        raise ValueError("%s contains NaN or +infinity" % name)
    return a


# ---------------------------------------------------------------------------
# host-pointer wrappers (NumPy in, NumPy out)
# ---------------------------------------------------------------------------
def knn(X, Y, K, self_mode=SELF_NONE, self_offset=0, return_index=True, device=0, options=None):
    """K nearest reference rows (Euclidean) for every query row.  Returns
    (dist[nq,K] ascending, idx[nq,K] int64 or None).  ``options``: an ``Options`` for THIS call (mce_knn_f64_opt)."""
    lib = load()
    X = _f64(X, "X")
    Y = _f64(Y, "Y")
    if X.ndim != 2 or Y.ndim != 2 or X.shape[1] != Y.shape[1]:
        raise ValueError("X and Y must be 2-D with the same number of columns, got %r and %r" % (X.shape, Y.shape))
    nq, d = X.shape
    nr = Y.shape[0]
    K = int(K)
    dist = np.empty((nq, K), dtype=np.float64)
    idx = np.empty((nq, K), dtype=np.int64) if return_index else None
    if options is None:
        check(lib.mce_knn_f64(X.ctypes.data, nq, Y.ctypes.data, nr, d, K, int(self_mode), int(self_offset),
                              dist.ctypes.data, idx.ctypes.data if idx is not None else None, int(device)))
    else:
        check(lib.mce_knn_f64_opt(X.ctypes.data, nq, Y.ctypes.data, nr, d, K, int(self_mode), int(self_offset),
                                  dist.ctypes.data, idx.ctypes.data if idx is not None else None, int(device), _c.addressof(options)))
    return dist, idx


def last_verify_rows():
    """rows the run-time certificate of the most recent host-pointer search re-checked (0: it did not run)"""
    return int(load().mce_last_verify_rows())


def verify_knn(X, Y, dist, self_mode=SELF_NONE, self_offset=0, nsample=1024, seed=0, device=0):
    """Run-time certificate of a finished search (mce_verify_knn_f64): ``nsample`` query rows spread over X are re-checked
    against ``dist`` (what ``knn`` returned for them) by an exact fp64 scan of all of Y -- none of the search's machinery.
    Returns the number of rows that failed (0 = the sample agrees)."""
    lib = load()
    X = _f64(X, "X")
    Y = _f64(Y, "Y")
    dist = _f64_allow_inf(dist)
    if X.ndim != 2 or Y.ndim != 2 or X.shape[1] != Y.shape[1] or dist.ndim != 2 or dist.shape[0] != X.shape[0]:
        raise ValueError("shape mismatch: X %r, Y %r, dist %r" % (X.shape, Y.shape, dist.shape))
    failed = _c.c_int32(0)
    rc = lib.mce_verify_knn_f64(X.ctypes.data, X.shape[0], Y.ctypes.data, Y.shape[0], X.shape[1], dist.shape[1], int(self_mode), int(self_offset),
                                dist.ctypes.data, dist.shape[1], int(nsample), int(seed) & (2 ** 64 - 1), _c.addressof(failed), int(device))
    if rc not in (MCE_OK, MCE_ERR_VERIFY):
        check(rc)
    return int(failed.value)


def dotp(dist, w, fs, d, k0, kmax, device=0):
    """sum_j V_d(dist[j,k]) / w[j] * exp(fs[j]) for k in [k0,kmax) (entries < k0 are 0)."""
    lib = load()
    dist = _f64_allow_inf(dist)
    w = _f64(w, "weight")
    fs = _f64_fs(fs)
    if dist.ndim != 2 or w.shape != (dist.shape[0],) or fs.shape != w.shape:
        raise ValueError("shape mismatch: dist %r, w %r, fs %r" % (dist.shape, w.shape, fs.shape))
    out = np.zeros(int(kmax), dtype=np.float64)
    check(lib.mce_dotp_f64(dist.ctypes.data, dist.shape[0], dist.shape[1], int(k0), int(kmax), int(d),
                           w.ctypes.data, fs.ctypes.data, out.ctypes.data, int(device)))
    return out


def _f64_allow_inf(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def knn_dotp(X, Y, w, fs, kmax, k0, self_offset=0, return_dist=False, devices=None):
    """Fused search + reduction.  Returns dotp[kmax] (and dist[nq,kmax-k0] if asked)."""
    lib = load()
    X = _f64(X, "X")
    Y = X if Y is None else _f64(Y, "Y")
    w = _f64(w, "weight")
    fs = _f64_fs(fs)
    if X.ndim != 2 or Y.ndim != 2 or X.shape[1] != Y.shape[1]:
        raise ValueError("X and Y must be 2-D with the same number of columns")
    nq, d = X.shape
    if w.shape != (nq,) or fs.shape != (nq,):
        raise ValueError("weight and fs must have one entry per query row")
    out = np.zeros(int(kmax), dtype=np.float64)
    dist = np.empty((nq, int(kmax) - int(k0)), dtype=np.float64) if return_dist else None
    if devices is None:
        devs, ndev = None, 0
    else:
        arr = (ctypes.c_int32 * len(devices))(*[int(x) for x in devices])
        devs, ndev = ctypes.cast(arr, ctypes.c_void_p), len(devices)
    check(lib.mce_knn_dotp_f64(X.ctypes.data, nq, Y.ctypes.data, Y.shape[0], d, int(kmax), int(k0), int(self_offset),
                               w.ctypes.data, fs.ctypes.data, out.ctypes.data,
                               dist.ctypes.data if dist is not None else None, devs, ndev))
    return (out, dist) if return_dist else out


def knn_dotp_part(Y, w, fs, kmax, part, nparts, device=0):
    """Partial auto-evidence sums over part ``part`` of ``nparts`` of the queries (the library chooses the
    partition; the parts add up to ``knn_dotp(Y, None, w, fs, kmax, 1)``).  One call per rank / device."""
    lib = load()
    Y = _f64(Y, "Y")
    w = _f64(w, "weight")
    fs = _f64_fs(fs)
    if Y.ndim != 2 or w.shape != (Y.shape[0],) or fs.shape != w.shape:
        raise ValueError("Y must be 2-D, weight and fs one entry per row")
    out = np.zeros(int(kmax), dtype=np.float64)
    check(lib.mce_knn_dotp_part_f64(Y.ctypes.data, Y.shape[0], Y.shape[1], int(kmax), int(part), int(nparts),
                                    w.ctypes.data, fs.ctypes.data, out.ctypes.data, int(device)))
    return out


def _rows_f64(a, name, d, check=True):
    """2-D fp64 array whose rows are contiguous in their first d columns (row stride arbitrary).
    ``check=False``: the callee detects NaN / infinity itself (the device covariance turns non-finite and
    the call fails with ValueError), sparing a host pass over the whole matrix."""
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] < d:
        raise ValueError("%s must be 2-D with at least %d columns" % (name, d))
    if a.dtype != np.float64 or a.strides[1] != 8 or a.strides[0] % 8 != 0 or a.strides[0] < 8 * d:
        a = np.ascontiguousarray(a[:, :d], dtype=np.float64)
    if check and not np.all(np.isfinite(a[:, :d])):
        raise ValueError("%s contains NaN or infinity" % name)
    return a


def evidence_feed(S1, S2, d, cov_mode, kmax, w, fs, device=0):
    """Covariance + whitening + kNN + reduction on the device from ONE upload of the raw parameter
    rows.  Returns (dotp[kmax], jacobian, eigenvalues[d])."""
    lib = load()
    S1 = _rows_f64(S1, "samples", d, check=False)
    S2 = None if S2 is None else _rows_f64(S2, "samples2", d, check=False)
    w = _f64(w, "weight")
    fs = _f64_fs(fs)
    if w.shape != (S1.shape[0],) or fs.shape != w.shape:
        raise ValueError("weight and fs must have one entry per s1 row")
    out = np.zeros(int(kmax))
    jac = ctypes.c_double(0.0)
    ev = np.zeros(int(d))
    check(lib.mce_evidence_feed_f64(S1.ctypes.data, S1.shape[0], S1.strides[0] // 8,
                                    S2.ctypes.data if S2 is not None else None, 0 if S2 is None else S2.shape[0],
                                    0 if S2 is None else S2.strides[0] // 8, int(d), int(cov_mode), int(kmax),
                                    w.ctypes.data, fs.ctypes.data, out.ctypes.data, ctypes.byref(jac), ev.ctypes.data, int(device)))
    return out, float(jac.value), ev


def evidence_feed_part(S1, S2, d, cov_mode, kmax, w, fs, part, nparts, device=0, want_checksum=True):
    """One rank's share of ``evidence_feed`` (``mce_evidence_feed_part_f64``): every rank passes the same arrays, uploads them
    once, whitens on its device and searches its share.  Returns (dotp_part[kmax], jacobian, eigenvalues[d], checksum) --
    the partial sums still to be all-reduced over the ranks, and the device-side fingerprint of the uploaded inputs
    (None if not asked for)."""
    lib = load()
    S1 = _rows_f64(S1, "samples", d, check=False)
    S2 = None if S2 is None else _rows_f64(S2, "samples2", d, check=False)
    w = _f64(w, "weight")
    fs = _f64_fs(fs)
    if w.shape != (S1.shape[0],) or fs.shape != w.shape:
        raise ValueError("weight and fs must have one entry per s1 row")
    out = np.zeros(int(kmax))
    jac = ctypes.c_double(0.0)
    ev = np.zeros(int(d))
    csum = ctypes.c_uint64(0)
    check(lib.mce_evidence_feed_part_f64(S1.ctypes.data, S1.shape[0], S1.strides[0] // 8,
                                         S2.ctypes.data if S2 is not None else None, 0 if S2 is None else S2.shape[0],
                                         0 if S2 is None else S2.strides[0] // 8, int(d), int(cov_mode), int(kmax),
                                         w.ctypes.data, fs.ctypes.data, int(part), int(nparts), out.ctypes.data, ctypes.byref(jac),
                                         ev.ctypes.data, ctypes.byref(csum) if want_checksum else None, int(device)))
    return out, float(jac.value), ev, (int(csum.value) if want_checksum else None)


def evidence_feed_part_dev(dS1, n1, ld1, dS2, n2, ld2, d, cov_mode, kmax, d_w, d_fs, part, nparts, device=0, want_checksum=True):
    """``evidence_feed_part`` with the inputs on the device already (``mce_evidence_feed_part_dev_f64``): device POINTERS (rows
    ``ld`` doubles apart; ``dS2`` = 0 for auto evidence), produced on a stream the caller has synchronised.  Same returns."""
    lib = load()
    out = np.zeros(int(kmax))
    jac = ctypes.c_double(0.0)
    ev = np.zeros(int(d))
    csum = ctypes.c_uint64(0)
    check(lib.mce_evidence_feed_part_dev_f64(dS1, int(n1), int(ld1), dS2 or None, int(n2) if dS2 else 0, int(ld2) if dS2 else 0, int(d), int(cov_mode), int(kmax),
                                             d_w, d_fs, int(part), int(nparts), out.ctypes.data, ctypes.byref(jac), ev.ctypes.data,
                                             ctypes.byref(csum) if want_checksum else None, int(device)))
    return out, float(jac.value), ev, (int(csum.value) if want_checksum else None)


def evidence_feed_whiten_dev(dS1, n1, ld1, d, kmax, d_w, d_fs, d_X_out, d_w_out, d_fs_out, device=0, want_checksum=True):
    """``evidence_feed_whiten`` with the inputs on the device already (``mce_evidence_feed_whiten_dev_f64``)."""
    lib = load()
    jac = ctypes.c_double(0.0)
    ev = np.zeros(int(d))
    csum = ctypes.c_uint64(0)
    check(lib.mce_evidence_feed_whiten_dev_f64(dS1, int(n1), int(ld1), int(d), int(kmax), d_w, d_fs, d_X_out, d_w_out, d_fs_out, ctypes.byref(jac),
                                               ev.ctypes.data, ctypes.byref(csum) if want_checksum else None, int(device)))
    return float(jac.value), ev, (int(csum.value) if want_checksum else None)


def evidence_feed_whiten(S1, d, kmax, w, fs, d_X_out, d_w_out, d_fs_out, device=0, want_checksum=True):
    """The feeders alone (``mce_evidence_feed_whiten_f64``, auto evidence): upload, covariance, eigen-system, whitening; the
    whitened rows, weights and likelihood terms are left in the caller's DEVICE buffers (pointers; [n, d], [n], [n] doubles).
    Returns (jacobian, eigenvalues[d], checksum)."""
    lib = load()
    S1 = _rows_f64(S1, "samples", d, check=False)
    w = _f64(w, "weight")
    fs = _f64_fs(fs)
    if w.shape != (S1.shape[0],) or fs.shape != w.shape:
        raise ValueError("weight and fs must have one entry per row")
    jac = ctypes.c_double(0.0)
    ev = np.zeros(int(d))
    csum = ctypes.c_uint64(0)
    check(lib.mce_evidence_feed_whiten_f64(S1.ctypes.data, S1.shape[0], S1.strides[0] // 8, int(d), int(kmax), w.ctypes.data, fs.ctypes.data,
                                           d_X_out, d_w_out, d_fs_out, ctypes.byref(jac), ev.ctypes.data,
                                           ctypes.byref(csum) if want_checksum else None, int(device)))
    return float(jac.value), ev, (int(csum.value) if want_checksum else None)


def _devices_arg(devices):
    if devices is None:
        return None, 0
    arr = (ctypes.c_int32 * len(devices))(*[int(x) for x in devices])
    return arr, len(devices)


def _raise_for(rc, msg):
    if rc in (MCE_ERR_INVALID, MCE_ERR_K_RANGE, MCE_ERR_WORKSPACE, MCE_ERR_DIM_RANGE):
        return ValueError(msg)
    return RuntimeError("mcevidence_amd HIP backend: " + msg)


def evidence_feed_batch(problems, devices=None, return_exceptions=False):
    """Many independent evidence problems in ONE library call (``mce_evidence_feed_batch_f64``).

    ``problems``: sequence of ``(S1, S2, d, cov_mode, kmax, w, fs)`` with the meaning of
    :func:`evidence_feed`.  Returns a list of ``(dotp[kmax], jacobian, eigenvalues[d])`` in the same
    order.  A failing problem raises (the first one, like a loop of single calls would) unless
    ``return_exceptions`` is set, in which case its slot holds the exception instead."""
    lib = load()
    n = len(problems)
    arr = (FeedProblem * max(n, 1))()
    keep = []                                  # arrays the C struct points into
    outs = []
    for i, (S1, S2, d, cov_mode, kmax, w, fs) in enumerate(problems):
        d, kmax = int(d), int(kmax)
        S1 = _rows_f64(S1, "samples", d, check=False)
        S2 = None if S2 is None else _rows_f64(S2, "samples2", d, check=False)
        w = _f64(w, "weight")
        fs = _f64_fs(fs)
        if w.shape != (S1.shape[0],) or fs.shape != w.shape:
            raise ValueError("problem %d: weight and fs must have one entry per s1 row" % i)
        out = np.zeros(max(kmax, 0))
        ev = np.zeros(max(d, 0))
        keep.append((S1, S2, w, fs))
        outs.append((out, ev))
        q = arr[i]
        q.S1, q.n1, q.ld1 = S1.ctypes.data, S1.shape[0], S1.strides[0] // 8
        if S2 is not None:
            q.S2, q.n2, q.ld2 = S2.ctypes.data, S2.shape[0], S2.strides[0] // 8
        q.d, q.cov_mode, q.kmax = d, int(cov_mode), kmax
        q.w, q.fs, q.dotp, q.eigenvalues = w.ctypes.data, fs.ctypes.data, out.ctypes.data, ev.ctypes.data
    devs, ndev = _devices_arg(devices)
    rc = lib.mce_evidence_feed_batch_f64(arr, n, ctypes.cast(devs, _P) if devs is not None else None, ndev)
    if rc != MCE_OK and all(arr[i].status == MCE_OK for i in range(n)):
        check(rc)                              # the machinery failed (allocation, device), not a problem
    results = []
    for i in range(n):
        st = int(arr[i].status)
        if st == MCE_OK:
            results.append((outs[i][0], float(arr[i].jacobian), outs[i][1]))
            continue
        # the library keeps only the first failure's text; later ones get a generic message
        first = all(int(arr[j].status) == MCE_OK for j in range(i))
        exc = _raise_for(st, last_error() if first else "problem %d failed with status %d" % (i, st))
        if not return_exceptions:
            raise exc
        results.append(exc)
    return results


# ---------------------------------------------------------------------------
# device-pointer wrappers (raw addresses: torch tensors' data_ptr(), stream handle)
# ---------------------------------------------------------------------------
def knn_workspace_bytes(nq, nr, d, K, options=None):
    """Scratch bytes a *_dev search needs.  ``options=Options(same_set=0)``: queries and references will be two buffers
    (cross evidence with equal halves): no scratch for the symmetric sweep is reserved."""
    if options is None:
        n = int(load().mce_knn_workspace_bytes(int(nq), int(nr), int(d), int(K)))
    else:
        n = int(load().mce_knn_workspace_bytes_opt(int(nq), int(nr), int(d), int(K), _c.addressof(options)))
    if n == 0:
        raise ValueError(last_error())
    return n


def dotp_workspace_bytes(nq, kmax):
    return int(load().mce_dotp_workspace_bytes(int(nq), int(kmax)))


def knn_dev(dX, nq, dY, nr, d, K, self_mode, self_offset, d_dist, d_idx, ws, ws_bytes, stream=0):
    check(load().mce_knn_f64_dev(dX, nq, dY, nr, d, K, self_mode, self_offset, d_dist, d_idx or None, ws, ws_bytes, stream or None))


def dotp_dev(d_dist, nq, ld, k0, kmax, d, d_w, d_fs, d_dotp, ws, ws_bytes, stream=0):
    check(load().mce_dotp_f64_dev(d_dist, nq, ld, k0, kmax, d, d_w, d_fs, d_dotp, ws, ws_bytes, stream or None))


def prune_part_applies(nr, d, kmax, nparts):
    """Does an auto-evidence search of this shape on ``nparts`` ranks take the pruned walk with a preparation that can be distributed?"""
    return bool(load().mce_prune_part_applies(int(nr), int(d), int(kmax), int(nparts)))


def prune_part_prepare_dev(dY, nr, d, kmax, part, nparts, ws, ws_bytes, stream=0, want_range=False):
    """This rank's part of the DISTRIBUTED k-d preparation of a pruned auto-evidence search (``mce_prune_part_prepare_dev``): returns
    (byte offset into ``ws``, count) of the int32 permutation array -- final inside the rank's range, zeros elsewhere: gather the
    ranges (or all-reduce(SUM) the array) over the ranks, then ``knn_dotp_part_prepared_dev`` -- or (0, 0) when there is nothing to
    exchange (call ``knn_dotp_part_dev``).  ``want_range``: also the rank's range (lo, hi) in positions."""
    off, cnt, lo, hi = _c.c_size_t(0), _c.c_int64(0), _c.c_int64(0), _c.c_int64(0)
    check(load().mce_prune_part_prepare_dev(dY, nr, d, kmax, part, nparts, _c.byref(off), _c.byref(cnt), _c.byref(lo), _c.byref(hi), ws, ws_bytes, stream or None))
    if want_range:
        return int(off.value), int(cnt.value), int(lo.value), int(hi.value)
    return int(off.value), int(cnt.value)


def knn_dotp_part_prepared_dev(dY, nr, d, kmax, part, nparts, d_w, d_fs, d_dotp, ws, ws_bytes, stream=0):
    check(load().mce_knn_dotp_part_prepared_f64_dev(dY, nr, d, kmax, part, nparts, d_w, d_fs, d_dotp, ws, ws_bytes, stream or None))


def knn_dotp_part_dev(dY, nr, d, kmax, part, nparts, d_w, d_fs, d_dotp, ws, ws_bytes, stream=0):
    check(load().mce_knn_dotp_part_f64_dev(dY, nr, d, kmax, part, nparts, d_w, d_fs, d_dotp, ws, ws_bytes, stream or None))


def pairs_once_blocks(nr, d, kmax):
    """Sorted 512-row blocks of the all-pairs-once partition for this shape (the length of its flags array); 0: the shape
    does not take the one-pass symmetric sweep and the partition does not exist (``mce_pairs_once_blocks``)."""
    return int(load().mce_pairs_once_blocks(int(nr), int(d), int(kmax)))


def pairs_once_workspace_bytes(nr, d, kmax, nparts):
    return int(load().mce_pairs_once_workspace_bytes(int(nr), int(d), int(kmax), int(nparts)))


def pairs_once_prepare_dev(dY, nr, d, kmax, part, nparts, ws, ws_bytes, stream=0):
    """-> (byte offset of the rows' bounds inside the workspace, their number): float64, to be all-reduced with MIN"""
    off, cnt = _c.c_size_t(0), _c.c_int64(0)
    check(load().mce_pairs_once_prepare_dev(dY, nr, d, kmax, part, nparts, _c.byref(off), _c.byref(cnt), ws, ws_bytes, stream or None))
    return int(off.value), int(cnt.value)


def pairs_once_sweep_dev(dY, nr, d, kmax, part, nparts, d_counts, d_flags, ws, ws_bytes, stream=0):
    check(load().mce_pairs_once_sweep_dev(dY, nr, d, kmax, part, nparts, d_counts, d_flags, ws, ws_bytes, stream or None))


def pairs_once_export_dev(nr, d, kmax, part, nparts, d_send, ws, ws_bytes, stream=0):
    check(load().mce_pairs_once_export_dev(nr, d, kmax, part, nparts, d_send or None, ws, ws_bytes, stream or None))


def pairs_once_finish_dev(dY, nr, d, kmax, part, nparts, d_w, d_fs, d_recv, nrecv, d_flags, d_dotp, ws, ws_bytes, stream=0):
    check(load().mce_pairs_once_finish_dev(dY, nr, d, kmax, part, nparts, d_w, d_fs, d_recv or None, nrecv, d_flags, d_dotp, ws, ws_bytes,
                                           stream or None))


def knn_dotp_dev(dX, nq, dY, nr, d, kmax, k0, self_offset, d_w, d_fs, d_dotp, d_dist_out, ws, ws_bytes, stream=0):
    check(load().mce_knn_dotp_f64_dev(dX, nq, dY, nr, d, kmax, k0, self_offset, d_w, d_fs, d_dotp, d_dist_out or None,
                                      ws, ws_bytes, stream or None))


def debug_mfma_tile(yprime, xprime, device=0):
    """Test hook (mce_debug_mfma_tile_f16): the 32 x 32 fp32 tile ``yprime @ xprime.T`` as the filter kernels' MFMA
    sequence computes it; ``yprime``, ``xprime``: float16 arrays [32, 16 * kst]."""
    y = np.ascontiguousarray(yprime, dtype=np.float16)
    x = np.ascontiguousarray(xprime, dtype=np.float16)
    if y.shape != x.shape or y.shape[0] != 32 or y.shape[1] % 16 or not 1 <= y.shape[1] // 16 <= 8:
        raise ValueError("yprime and xprime must be float16 [32, 16*kst], kst = 1..8")
    out = np.empty((32, 32), dtype=np.float32)
    check(load().mce_debug_mfma_tile_f16(y.ctypes.data, x.ctypes.data, y.shape[1] // 16, out.ctypes.data, int(device)))
    return out


def debug_mfma_tiles(yprime, xprime, device=0):
    """``debug_mfma_tile`` for a batch: float16 arrays [ntiles, 32, 16 * kst] -> float32 [ntiles, 32, 32] (row, query)."""
    y = np.ascontiguousarray(yprime, dtype=np.float16)
    x = np.ascontiguousarray(xprime, dtype=np.float16)
    if y.ndim != 3 or y.shape != x.shape or y.shape[1] != 32 or y.shape[2] % 16 or not 1 <= y.shape[2] // 16 <= 8:
        raise ValueError("yprime and xprime must be float16 [ntiles, 32, 16*kst], kst = 1..8")
    out = np.empty((y.shape[0], 32, 32), dtype=np.float32)
    check(load().mce_debug_mfma_tiles_f16(y.ctypes.data, x.ctypes.data, y.shape[2] // 16, y.shape[0], out.ctypes.data, int(device)))
    return out
