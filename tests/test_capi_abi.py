"""The C-ABI library loads and exports every symbol include/mcevidence_hip.h declares;
argument validation works without a GPU; compute calls fail LOUDLY without one
(no CPU fallback in the product).  CPU only."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import REPO
from mcevidence_amd import _capi

HEADER = os.path.join(REPO, "include", "mcevidence_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mce_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_capi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), "libmcevidence_hip.so does not export %s" % s
    # and the binding covers exactly the header
    assert sorted(_capi.SIGNATURES) == syms


def test_header_constants_match_binding():
    txt = open(HEADER).read()
    for name in ("MCE_MAX_K", "MCE_MAX_DIM", "MCE_ERR_INVALID", "MCE_ERR_K_RANGE", "MCE_ERR_HIP", "MCE_ERR_NO_DEVICE",
                 "MCE_ERR_WORKSPACE", "MCE_ERR_DIM_RANGE"):
        m = re.search(r"#define\s+%s\s+\(?(-?\d+)\)?" % name, txt)
        assert m and int(m.group(1)) == getattr(_capi, name), name
    assert _capi.load().mce_abi_version() == 3


def test_argument_validation_without_gpu():
    X = np.zeros((5, 3))
    with pytest.raises(ValueError, match="n_neighbors <= n_samples_fit"):
        _capi.knn(X, X, 9)                                      # sklearn's error for K > n
    with pytest.raises(ValueError, match="n_neighbors <= n_samples_fit"):
        _capi.knn(X, X, 5, self_mode=_capi.SELF_EXCLUDE)        # self excluded: only 4 usable rows
    with pytest.raises(ValueError):
        _capi.knn(np.zeros((5, 1025)), np.zeros((5, 1025)), 2)  # d > MCE_GENERIC_MAX_DIM
    with pytest.raises(ValueError):
        _capi.knn(np.zeros((2000, 3)), np.zeros((2000, 3)), 1025)   # K > MCE_GENERIC_MAX_K
    with pytest.raises(ValueError):
        _capi.knn(np.zeros((5, 3)), np.zeros((5, 4)), 2)        # column mismatch
    with pytest.raises(ValueError):
        _capi.knn(np.full((5, 3), np.nan), X, 2)                # NaN input (sklearn check_array)
    with pytest.raises(ValueError):
        _capi.knn_dotp(X, None, np.ones(5), np.zeros(5), kmax=2, k0=3)
    with pytest.raises(ValueError):
        _capi.knn_dotp(X, None, np.ones(4), np.zeros(5), kmax=2, k0=1)
    assert _capi.knn_workspace_bytes(1000, 1000, 6, 4) > 0
    assert _capi.dotp_workspace_bytes(1000, 4) >= 4 * 4 * 8
    assert _capi.knn_workspace_bytes(10, 10, 100, 4) > 0         # 64 <= d <= 127: the fp64 sweep's wide form
    assert _capi.knn_workspace_bytes(10, 10, 300, 4) > 0         # d > 127: the long-row fp64 sweep (K > 32: the generic exact kernel)
    assert _capi.knn_workspace_bytes(10, 100, 300, 40) > 0
    with pytest.raises(ValueError):
        _capi.knn_workspace_bytes(10, 10, 2000, 4)


def test_mode_switches_validate_without_gpu():
    assert _capi.get_search_mode() in (0, 1, 2) and _capi.get_prune_mode() in (0, 1, 2)
    for setter, getter in ((_capi.set_search_mode, _capi.get_search_mode), (_capi.set_prune_mode, _capi.get_prune_mode)):
        old = getter()
        for m in (1, 2, 0):
            setter(m)
            assert getter() == m
        with pytest.raises(ValueError):
            setter(3)
        with pytest.raises(ValueError):
            setter(-1)
        setter(old)
    # the pruned walk needs more scratch (k-d ordering, boxes, chunk lists); the plan is a pure function
    _capi.set_prune_mode(_capi.PRUNE_OFF)
    plain = _capi.knn_workspace_bytes(100000, 100000, 6, 4)
    _capi.set_prune_mode(_capi.PRUNE_FORCE)
    pruned = _capi.knn_workspace_bytes(100000, 100000, 6, 4)
    assert _capi.knn_workspace_bytes(200000, 200000, 27, 4) > 0          # d > 13: never pruned
    _capi.set_prune_mode(_capi.PRUNE_AUTO)
    assert pruned > plain and _capi.knn_workspace_bytes(100000, 100000, 6, 4) == plain   # 100k x 6: below the automatic threshold
    assert _capi.knn_workspace_bytes(1000000, 1000000, 6, 4) > 12 * plain                # 1M x 6: automatic
    with pytest.raises(ValueError, match="no pruned search"):
        _capi.last_prune_stats()


def test_compute_fails_loudly_without_gpu():
    if _capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    X = np.random.default_rng(0).standard_normal((50, 3))
    with pytest.raises(RuntimeError, match="no HIP device"):
        _capi.knn(X, X, 3)
    with pytest.raises(RuntimeError, match="no HIP device"):
        _capi.knn_dotp(X, None, np.ones(50), np.zeros(50), kmax=3, k0=1)
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    with pytest.raises(RuntimeError):
        pkg.MCEvidence([gaussian_chain(0, 300, 3)], verbose=0).evidence()
    from mcevidence_amd.neighbors import NearestNeighbors
    with pytest.raises(RuntimeError):
        NearestNeighbors(n_neighbors=3).fit(X).kneighbors(X)


def test_product_never_imports_the_oracle():
    """the package must not reference oracle/ anywhere (parity claims depend on it)."""
    pkgdir = os.path.join(REPO, "mcevidence_amd")
    for root, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.lower(), "%s mentions the oracle" % f


def test_symmetric_sweep_unit_enumeration(tmp_path):
    """the symmetric sweep's workgroups are numbered panel by panel (mcevidence_amd/csrc/sym_types.hpp); host and
    kernel share the decode: every (panel, block) once, a block's units tile its rows without gaps, its p-th unit is
    panel p (the list hand-over counter), the count is the grid size -- checked on the CPU, with UBSan"""
    import subprocess
    exe = str(tmp_path / "sym_units_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fsanitize=undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(REPO, "mcevidence_amd", "csrc"),
                           os.path.join(REPO, "tests", "native", "sym_units_check.cpp"), "-o", exe])
    out = subprocess.check_output([exe]).decode()
    assert out.startswith("ok ")


def test_workspace_query_with_the_same_set_hint():
    """mce_knn_workspace_bytes_opt: a caller who knows that queries and references are two buffers (cross evidence with
    equal halves) says so and is not charged the symmetric sweep's scratch (~1.7 GB at 1M rows, K = 9); sizes only, no
    device needed."""
    from mcevidence_amd import _capi
    n, d, K = 1_000_000, 27, 9
    unknown = _capi.knn_workspace_bytes(n, n, d, K)
    same = _capi.knn_workspace_bytes(n, n, d, K, options=_capi.Options(same_set=1))
    apart = _capi.knn_workspace_bytes(n, n, d, K, options=_capi.Options(same_set=0))
    assert unknown == same and apart < same - 1_000_000_000
    never = _capi.knn_workspace_bytes(n, n, d, K, options=_capi.Options(sym_mode=_capi.SYM_OFF))
    assert never == apart
    with pytest.raises(ValueError):
        _capi.knn_workspace_bytes(n, n, d, K, options=_capi.Options(prune_mode=5))
