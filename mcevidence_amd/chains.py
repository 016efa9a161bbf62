"""Host-side chain container: reading, burn-in, thinning, s1/s2 split.

Mirrors the interface of the reference's ``MCSamples`` / ``SamplesMIXIN``
(``/root/reference/MCEvidence.py:107-607``) -- same attribute names
(``data['s1'].samples / .weights / .loglikes / .adjusted_weights``), same column
convention (col 0 weight, col 1 -ln L, cols 2.. parameters; overridable with
``iw/ilike/itheta``), same burn-in and thinning rules -- written from scratch.
This stays on the host by design (BASELINE.json north_star).

Behaviour kept on purpose (pinned by tests/golden/host_pins.json):
  * chains passed in memory (list/tuple of arrays) IGNORE ``burnlen``/``thinlen``;
    only chains read from files are burned/thinned (reference :151 vs :606);
  * the random split uses the global NumPy RNG (``np.random.choice``), so
    ``np.random.seed(s)`` before construction reproduces the reference split;
  * integer-weight thinning can replicate rows (getdist's algorithm).
Deliberate deviations (the reference crashes by accident there): a dict of arrays
is accepted (reference: TypeError on py3), ``thinlen == 1`` is a no-op (reference:
TypeError), a bare ndarray raises TypeError with a message (reference: bare ``raise``).
"""
from __future__ import annotations

import glob
import logging
import os

import numpy as np

logger = logging.getLogger("mcevidence_amd")


def rank0_draw(draw):
    """Every random decision of the host bookkeeping (the s1/s2 split, Poisson thinning, random batches)
    comes from the process's own global NumPy RNG, as in the reference (:225, :417-445, :917).  Under
    ``torchrun`` each rank is its own process with its own unseeded RNG, and the query-sharded hot path
    (``parallel.sharded_knn_dotp``) needs every rank to hold the SAME rows: rank 0 draws, the others
    receive its result (one ``broadcast_object_list``; the other ranks' RNGs are not advanced).
    Outside a process group this is just ``draw()``."""
    from . import parallel
    if not parallel.is_distributed():
        return draw()
    import torch.distributed as dist
    group = parallel.current_group()           # the group of the evidence computation (parallel.set_group), default: world
    lead = dist.get_rank(group) == 0
    box = [draw() if lead else None]
    dist.broadcast_object_list(box, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
    return box[0]


def read_chain_file(path):
    """One chain text file -> fp64 array [rows, columns]: what ``np.loadtxt(f)`` gives the reference
    (:564), read by the native multi-threaded reader (``chain_io`` / ``libmcechains.so``).
    ``MCE_CHAIN_READER=numpy`` selects NumPy's reader instead."""
    if os.environ.get("MCE_CHAIN_READER", "native") == "numpy" or not _native_reader():
        return np.loadtxt(path, ndmin=2)
    from . import chain_io
    return chain_io.loadtxt(path)


_NATIVE_READER = None


def _native_reader():
    """Is libmcechains.so there?  Reading chains is host bookkeeping, not the hot path: without the native
    reader (an install that lost the library) the files are read by ``np.loadtxt`` -- the reference's own
    reader (:564), same values, ~10x slower -- with ONE warning.  (The HIP hot path has no such fallback.)"""
    global _NATIVE_READER
    if _NATIVE_READER is None:
        from . import chain_io
        _NATIVE_READER = os.path.exists(chain_io.LIB_PATH)
        if not _NATIVE_READER:
            import warnings
            warnings.warn("mcevidence_amd: %s not found (build it with `make -C mcevidence_amd/csrc`); "
                          "reading chain files with numpy.loadtxt instead" % chain_io.LIB_PATH, RuntimeWarning)
    return _NATIVE_READER


def read_chain_files(paths):
    """The chain files of one root, in order.  Small files (a Planck chain is ~3 MB: one reader thread
    each) are parsed concurrently -- the native reader runs outside the GIL."""
    paths = list(paths)
    if len(paths) < 2 or os.environ.get("MCE_CHAIN_READER", "native") == "numpy" or not _native_reader():
        return [read_chain_file(f) for f in paths]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(paths), 8)) as pool:
        return list(pool.map(read_chain_file, paths))


class Partition(object):
    """One partition (s1 or s2) of the samples."""

    def __init__(self, samples=None, weights=None, loglikes=None, rows=None):
        self.samples = samples
        # the two bookkeeping columns as contiguous vectors: as strided views of the chain array every pass over
        # them (negation, max, upload) costs 3-7 ms per million rows
        self.weights = None if weights is None else np.ascontiguousarray(weights)
        self.loglikes = None if loglikes is None else np.ascontiguousarray(loglikes)
        self.ichain = rows
        # a copy that importance sampling may alter independently (reference :244-247)
        self.adjusted_weights = None if weights is None else np.array(weights, copy=True)

    def __len__(self):
        return 0 if self.samples is None else len(self.samples)


# ---------------------------------------------------------------------------
# thinning index rules
# ---------------------------------------------------------------------------
def integer_weight_thin(weights, factor):
    """Indices that keep one row per ``factor`` units of (integer) weight; rows
    heavier than ``factor`` are repeated.  Same rule as getdist's
    ``WeightedSamples.thin_indices`` which the reference adopts (:481-532).
    Returns (indices, weights[indices])."""
    w = np.asarray(weights)
    wi = w.astype(int)
    if abs(float(np.sum(wi)) - float(np.sum(w))) > 1e-4:
        raise ValueError("integer-weight thinning needs integer weights")
    if factor != int(factor):
        raise ValueError("thin factor must be an integer")
    factor = int(factor)
    n = len(wi)
    if factor >= np.max(wi):
        _, keep = np.unique(np.cumsum(wi) // factor, return_index=True)
    else:
        # unroll every row into w unit-weight copies and keep the copy that completes each
        # group of `factor` units: group m ends at unit m*factor, which lies in the first row
        # whose cumulative weight reaches it (rows heavier than `factor` are kept repeatedly).
        csum = np.cumsum(wi)
        ends = factor * np.arange(1, int(csum[-1]) // factor + 1)
        keep = np.searchsorted(csum, ends, side="left")
    return keep, wi[keep]


def max_weight_bin_thin(weights, unit):
    """Non-integer weights: cut the chain into ~N/unit equal index bins and keep the
    heaviest row of each (reference ``weighted_thin`` :447-479)."""
    w = np.asarray(weights)
    n = len(w)
    if unit == 0:
        return np.arange(n), w
    nbins = int(n * unit) if unit < 1 else int(n // unit)
    edges = np.linspace(-1, n, nbins + 1)
    which = np.digitize(np.arange(n), edges)
    # first index of the maximum inside every bin, bins in ascending order
    order = np.lexsort((np.arange(n), -w, which))
    first = np.ones(n, dtype=bool)
    first[1:] = which[order][1:] != which[order][:-1]
    keep = order[first].astype(np.intp)
    return keep, w[keep]


def poisson_thin(weights, retain_fraction):
    """0 < thinlen < 1: new weight ~ Poisson(w * fraction) drawn row by row from the
    global NumPy RNG; rows with zero draws are dropped (reference :417-445)."""
    w = np.asarray(weights) * retain_fraction
    draws = rank0_draw(lambda: np.array([float(np.random.poisson(x)) for x in w]))
    keep = np.where(draws > 0)[0]
    return keep, draws[keep]


def thin_rows(weights, nthin):
    """Dispatch of reference ``get_thin_index`` (:272-287). Returns (indices, new_weights)."""
    if nthin < 1:
        return poisson_thin(weights, nthin)
    try:
        return integer_weight_thin(weights, nthin)
    except ValueError:
        return max_weight_bin_thin(weights, nthin)


# ---------------------------------------------------------------------------
class MCSamples(object):
    """Container for one or more MCMC chains.

    str_or_dict : chain file root / file name / wildcard (str), or list/tuple/dict of
                  2-D arrays (one per chain).
    csplit      : object with .split/.frac/.shuffle, or None (no split).
    kwargs      : iw, ilike, itheta, log_level, burnlen, thinlen, idchain, idpattern.
    """

    def __init__(self, str_or_dict, trueval=None, debug=False, csplit=None, names=None, labels=None,
                 px="x", **kwargs):
        self.debug = debug
        self.names = None
        self.labels = None
        self.trueval = trueval
        self.px = px
        self.split = bool(csplit.split) if csplit is not None else False
        self.s1frac = csplit.frac if csplit is not None else 0.5
        self.shuffle = csplit.shuffle if csplit is not None else True
        self.iw = kwargs.pop("iw", 0)
        self.ilike = kwargs.pop("ilike", 1)
        self.itheta = kwargs.pop("itheta", 2)
        kwargs.pop("log_level", None)
        self.logger = logger
        self.chains = None

        if isinstance(str_or_dict, str):
            self.logger.info("Loading chain from " + str_or_dict)
            self.data = self.load_from_file(str_or_dict, **kwargs)
        elif isinstance(str_or_dict, (list, tuple, dict)):
            seq = list(str_or_dict.values()) if isinstance(str_or_dict, dict) else list(str_or_dict)
            if len(seq) and isinstance(seq[0], str):
                # a list of file names: read them, then treat like a file root (burn/thin honoured)
                self.data = self.load_from_file(seq, **kwargs)
            else:
                self.chains = [np.asarray(c, dtype=np.float64) for c in seq]
                self.data = self.chains2samples()          # NO kwargs: burn/thin ignored (reference :151)
        else:
            raise TypeError("first argument must be a chain file name (str) or a list/tuple/dict of 2-D "
                            "chain arrays, got %s" % type(str_or_dict).__name__)
        self.nparamMC = self.get_shape()[1]
        ndim = self.nparamMC
        self.names = ["p%s" % i for i in range(ndim)]
        self.labels = ["%s_%s" % (self.px, i) for i in range(ndim)]

    # -- reading ----------------------------------------------------------
    def load_from_file(self, fname, **kwargs):
        """CosmoMC text chains: ``root_1.txt .. root_n.txt`` (all, or ``idchain``), an
        explicit file, a list of files, or a wildcard (reference :567-606)."""
        if isinstance(fname, (list, tuple)):
            flist = list(fname)
        elif os.path.isfile(fname):
            flist = [fname]
        elif "*" in fname or "?" in fname:
            flist = sorted(glob.glob(fname))
        else:
            idchain = kwargs.pop("idchain", 0)
            if idchain > 0:
                flist = ["%s_%d.txt" % (fname, idchain)]
            else:
                pattern = kwargs.pop("idpattern", "_?.txt")
                flist = sorted(glob.glob(fname + pattern))
        kwargs.pop("idchain", None)
        kwargs.pop("idpattern", None)
        if not flist:
            raise IOError("no chain files found for %r" % (fname,))
        self.logger.debug("Reading from files: " + ", ".join(flist))
        self.chains = read_chain_files(flist)
        return self.chains2samples(**kwargs)

    # -- burn / concatenate / thin / split -----------------------------------
    def chains2samples(self, **kwargs):
        if self.chains is None or len(self.chains) == 0:
            raise ValueError("the chains array is empty")
        burnlen = kwargs.pop("burnlen", 0)
        thinlen = kwargs.pop("thinlen", 0)
        self.nchains = len(self.chains)
        if burnlen > 0:
            self.chains = [self.removeBurn(burnlen, chain=c) for c in self.chains]
        self.chain_offsets = np.cumsum([0] + [c.shape[0] for c in self.chains])
        self.ichain = np.concatenate([(i + 1) * np.ones(len(c)) for i, c in enumerate(self.chains)])
        self.samples = np.concatenate(self.chains)
        if abs(thinlen) > 0:
            self.samples = self.thin(nthin=thinlen, chain=self.samples)
        self.chains = None
        return self.chain_split(self.samples)

    def removeBurn(self, remove, chain):
        """burnlen < 1 is a fraction of the chain, otherwise a row count (reference :350-391)."""
        start = int(chain.shape[0] * remove) if remove < 1 else int(remove)
        self.logger.info("Removing %s lines as burn in" % start)
        return chain[start:, :]

    def thin(self, nthin=1, chain=None):
        if nthin == 1:
            return chain
        if nthin < 0:
            raise ValueError("negative thinlen (autocorrelation-length thinning) is not supported")
        w = chain[:, self.iw]
        keep, neww = thin_rows(w, nthin)
        out = chain[keep, :]
        out[:, self.iw] = neww
        self.logger.info("Thinning with thin length=%s: #old_chain=%s, #new_chain=%s" % (nthin, len(w), len(neww)))
        return out

    def chain_split(self, s):
        """split=True: s1 = random ``int(N*s1frac)`` rows (global RNG), s2 = the rest in
        ascending row order (reference :221-249)."""
        if self.split:
            nrow = len(s)
            pick = rank0_draw(lambda: np.random.choice(range(nrow), size=int(nrow * self.s1frac), replace=False))
            rest = np.setxor1d(range(nrow), pick)
            self.logger.info("%s chain with nrow=%s split to ns1=%s, ns2=%s" % (self.nchains, nrow, len(pick), len(rest)))
            return self._partitions(s, pick, rest)
        return self._partitions(s, None, None)

    def set_split(self, s1_rows, s2_rows):
        """Use an explicit, caller-chosen split (e.g. two independent chains) instead of
        the random one; rows index the concatenated sample array."""
        self.split = True
        self.data = self._partitions(self.samples, np.asarray(s1_rows), np.asarray(s2_rows))

    def _partitions(self, s, rows1, rows2):
        def part(rows):
            a = s if rows is None else s[rows, :]
            return Partition(a[:, self.itheta:], a[:, self.iw], a[:, self.ilike],
                             range(len(s)) if rows is None else rows)
        if rows1 is None:
            return {"s1": part(None), "s2": Partition()}
        return {"s1": part(rows1), "s2": part(rows2)}

    # -- accessors ----------------------------------------------------------
    def get_shape(self, name="s1"):
        def shp(p):
            return (0, 0) if p.samples is None else p.samples.shape
        if name in ("s1", "s2"):
            return shp(self.data[name])
        a, b = shp(self.data["s1"]), shp(self.data["s2"])
        return (a[0] + b[0], a[1])

    def arrays(self, name="s1"):
        """(samples, lnp, weights) with lnp = -loglikes (reference :394-405)."""
        if name in ("s1", "s2"):
            p = self.data[name]
            if p.samples is None:
                return None, None, None
            return p.samples, -p.loglikes, p.weights
        return self.all_sample_arrays()

    def all_sample_arrays(self):
        s, lnp, w = self.arrays("s1")
        s2, lnp2, w2 = self.arrays("s2")
        if s2 is None:
            return s, lnp, w
        return np.concatenate((s, s2)), np.concatenate((lnp, lnp2)), np.concatenate((w, w2))

    def importance_sample(self, func, name="s1"):
        """adjusted_weights *= exp(-func(samples)); the original weights (used in the
        volume sum) are untouched (reference :265-270)."""
        self.data[name].adjusted_weights *= np.exp(-func(self.data[name].samples))
