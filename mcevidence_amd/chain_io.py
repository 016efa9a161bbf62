"""ctypes binding of ``libmcechains.so`` (C ABI: ``include/mcechains.h``): a multi-threaded,
mmap-based reader for chain text files.

``loadtxt(path)`` replaces ``np.loadtxt(f)`` at ``/root/reference/MCEvidence.py:564`` for CosmoMC /
MontePython chains (whitespace-separated numbers, ``#`` comments): same array, bit for bit
(every field is the correctly rounded fp64 value, like Python's ``float()``), ~100x faster.
Host-only native code; no GPU involved.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmcechains.so")

MCC_OK, MCC_ERR_IO, MCC_ERR_PARSE, MCC_ERR_RAGGED, MCC_ERR_INVALID = 0, -1, -2, -3, -4

_c = ctypes
_P = _c.c_void_p
#: every symbol include/mcechains.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mce_chain_abi_version": (_c.c_int, []),
    "mce_chain_last_error": (_c.c_char_p, []),
    "mce_chain_open": (_c.c_int, [_c.c_char_p, _c.c_int32, _c.POINTER(_P), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
    "mce_chain_read": (_c.c_int, [_P, _P]),
    "mce_chain_close": (None, [_P]),
    "mce_chain_parse_token": (_c.c_int, [_c.c_char_p, _c.c_int64, _c.POINTER(_c.c_double)]),
    "mce_chain_fingerprint_f64": (_c.c_uint64, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_uint64, _c.c_int32]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("mcevidence_amd: %s not found -- build it with `make -C mcevidence_amd/csrc`" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.mce_chain_abi_version() != 1:
            raise RuntimeError("mcevidence_amd: libmcechains ABI version mismatch")
        _lib = lib
    return _lib


def _raise(rc, lib):
    msg = lib.mce_chain_last_error().decode("utf-8", "replace")
    if rc == MCC_ERR_IO:
        raise OSError(msg)
    raise ValueError(msg)


def loadtxt(path, ndmin=2, nthreads=0):
    """The array ``np.loadtxt(path, ndmin=ndmin)`` returns for a numeric text file, as fp64.
    ``nthreads=0`` lets the library choose (one thread per ~4 MB, up to the core count)."""
    lib = load()
    handle = _P()
    nrows, ncols = _c.c_int64(), _c.c_int64()
    rc = lib.mce_chain_open(os.fsencode(path), int(nthreads), ctypes.byref(handle), ctypes.byref(nrows), ctypes.byref(ncols))
    if rc != MCC_OK:
        _raise(rc, lib)
    try:
        out = np.empty((nrows.value, ncols.value), dtype=np.float64)
        rc = lib.mce_chain_read(handle, out.ctypes.data)
        if rc != MCC_OK:
            _raise(rc, lib)
    finally:
        lib.mce_chain_close(handle)
    if out.shape[0] == 0:                       # np.loadtxt: empty input -> shape (0,), (0, 1) with ndmin=2
        out = np.empty((0, 1) if ndmin == 2 else (0,))
    elif ndmin < 2:
        out = np.squeeze(out)
        if ndmin == 1 and out.ndim == 0:
            out = out.reshape(1)
    return out


def parse_token(text):
    """One field -> float, exactly as the reader converts it (tests)."""
    lib = load()
    b = text.encode("ascii")
    v = _c.c_double()
    rc = lib.mce_chain_parse_token(b, len(b), ctypes.byref(v))
    if rc != MCC_OK:
        _raise(rc, lib)
    return v.value


def feed_fingerprint(S1, S2, d, w, fs, nthreads=0):
    """The 64-bit fingerprint ``mce_evidence_feed_part_f64`` computes on the device over what it uploaded -- the first ``d``
    columns of ``S1`` followed by those of ``S2`` (if any), the weights, the likelihood terms -- computed on the HOST's cores
    from the caller's arrays (``mce_chain_fingerprint_f64``; ~10 ms per 200 MB on 8 threads; releases the GIL, so it can run
    in a thread beside the GPU work).  Equal on two ranks iff they hold the same inputs."""
    lib = load()
    S1 = np.asarray(S1)
    n1 = S1.shape[0]
    mask = (1 << 64) - 1
    tot = 0

    def rows(a, first_row):
        a = np.asarray(a)
        if a.dtype != np.float64 or a.ndim != 2 or a.strides[1] != 8 or a.strides[0] % 8 or a.strides[0] < 8 * d:
            a = np.ascontiguousarray(a[:, :d], dtype=np.float64)
        return int(lib.mce_chain_fingerprint_f64(a.ctypes.data, a.shape[0], int(d), a.strides[0] // 8, (salt0 + first_row * d) & mask, int(nthreads)))

    salt0 = ((1 << 56) ^ (n1 << 8) ^ int(d)) & mask
    tot += rows(S1, 0)
    if S2 is not None:
        tot += rows(S2, n1)             # (the device holds s2's rows right behind s1's: word index n1 d + ...)
    for b, v in ((2, w), (3, fs)):
        v = np.ascontiguousarray(v, dtype=np.float64)
        salt = ((b << 56) ^ (n1 << 8) ^ int(d)) & mask
        tot += int(lib.mce_chain_fingerprint_f64(v.ctypes.data, v.shape[0], 1, 1, salt, int(nthreads)))
    return tot & mask
