"""``MCEvidence`` -- drop-in for the reference class of the same name.

Same constructor and ``evidence()`` signatures as
``/root/reference/MCEvidence.py:614-624`` and ``:950-952``.  Everything around
the hot path (chain handling, whitening, prior volume, the final ln-evidence
assembly) is host NumPy, as in the reference and as BASELINE.json's north_star
keeps it; the hot path itself --

    nbrs = NearestNeighbors(n_neighbors=kmax+1, ...).fit(Y)      # :1093-1101
    DkNN, indices = nbrs.kneighbors(samples)                      # :1104
    volume[j,k] = pi^(D/2) DkNN[j,k]^D / Gamma(1+D/2)             # :1107-1110
    dotp = np.dot(volume[:,k]/weight, np.exp(fs))                 # :1117

-- is ONE call into the HIP library (``_capi.knn_dotp`` ->
``mce_knn_dotp_f64``), or, under ``torchrun``, one call per rank on its query
shard followed by a single RCCL all-reduce of the ``kmax`` partial sums
(``parallel.sharded_knn_dotp``).  There is no CPU implementation of the hot
path in this package; without the shared library or a GPU it raises.
"""
from __future__ import annotations

import logging
import math
import statistics
from collections import namedtuple

import numpy as np

from .chains import MCSamples, rank0_draw

FORMAT = "%(levelname)s:%(filename)s.%(funcName)s():%(lineno)-8s %(message)s"
logger = logging.getLogger("mcevidence_amd")

__all__ = ["MCEvidence", "HipBackend", "evidence_many"]


class HipBackend(object):
    """The MI355X hot path.  ``knn_dotp`` returns (dotp[kmax], dist or None) where dist
    holds the distances that entered the sum (reference DkNN columns k0..kmax-1)."""

    name = "hip"

    def __init__(self, devices=None, verify=True, recheck_rows=None):
        self.devices = devices
        self.verify = verify        # multi-rank runs: compare a fingerprint of the inputs across the ranks on every call
        # run-time certificate of the search (mce_options.verify): after every single-process search this many query rows,
        # spread over the set, are re-checked by an exact fp64 scan of all reference rows that shares nothing with the
        # search kernels; a disagreement raises RuntimeError.  None (the default): the library's choice -- 256 rows behind
        # the fp16 filter (MCE_VERIFY=n / MCE_VERIFY=0 in the environment), nothing behind the fp64 kernels; 0 = off;
        # ~0.5 ms per 256 rows at 1 M x 27.
        self.recheck_rows = None if recheck_rows is None else int(recheck_rows)

    def _scoped(self):
        """the per-call options of this backend's library calls (thread-scoped: mce_options_push / _pop)"""
        import contextlib
        from . import _capi
        return _capi.options(verify=self.recheck_rows) if self.recheck_rows is not None else contextlib.nullcontext()

    def evidence_feed(self, S1, S2, ndim, cov_mode, kmax, weight, fs):
        """feeders on the device too (get_covariance + diagonalise_chain + the hot path, one upload);
        returns (dotp, J) or None when this route does not apply (multi-process / multi-device runs)."""
        from . import parallel
        if ndim > 127:            # (device feeders: d <= 127 -- the fp16 filter up to 63, the fp64 sweep's wide form beyond)
            return None
        from . import _capi
        if parallel.is_distributed():
            # one process per GPU: every rank uploads the (replicated) chain once, whitens it on ITS device and searches
            # its share; ONE all-reduce of kmax (+ 8: the input fingerprints) doubles completes the sums
            # (mce_evidence_feed_part_f64; reference MCEvidence.py:1034-1131)
            import torch
            import torch.distributed as dist
            if self.devices not in (None, [0], (0,)) and len(self.devices) != 1:
                return None
            group = parallel.current_group()
            dev = self.devices[0] if self.devices else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
            # this rank's device is the process's current one (what the torch-side routes below work on).  The routes differ in their
            # COLLECTIVES, so none of them is chosen by a rank for itself: each is entered by every rank and agreed inside, in one
            # all-reduce(MIN) of the ranks' own answers (the variable set on one rank only, a rank on another device: everybody
            # takes the fallback together -- ADVICE round 5)
            here = torch.cuda.is_available() and dev == torch.cuda.current_device()
            multi = dist.get_world_size(group) >= 2
            if S2 is None and multi:
                if parallel.pairs_once_route(np.asarray(S1).shape[0], ndim, kmax, group, local_ok=(parallel.pairs_once_enabled() and here)):
                    # MCE_PAIRS_ONCE=1 on EVERY rank: every pair of rows multiplied once per node
                    got = parallel.pairs_once_feed(S1, ndim, kmax, weight, fs, group, verify=self.verify)
                    if got is not None:
                        return got
                # a search that takes the pruned walk (C5's shape): the k-d preparation is distributed over the ranks -- whitening on
                # the device, this rank's part of the sorts, one gather of the permutation, the search (None: not this shape)
                got = parallel.pruned_part_feed(S1, ndim, kmax, weight, fs, group, verify=self.verify, local_ok=here)
                if got is not None:
                    return got
            part, jac, csum, failed = np.zeros(kmax), float("nan"), None, None
            # ONE upload of the chain per node (round 6): every rank uploads 1/W of the rows, an all_gather over RCCL hands
            # everybody the whole set; the ranks' inputs are compared through fingerprints of their HOST copies, computed on a
            # thread beside the GPU work (the gathered set is identical everywhere by construction).
            gathered, hostsum = None, None
            if multi or parallel._forced():
                want_node = here and parallel.node_upload_enabled(group)
                hostsum = parallel._HostFingerprint(S1, S2, ndim, weight, fs) if (self.verify and want_node) else None
                gathered = parallel.gather_chain_on_device(S1, S2, ndim, weight, fs, group, local_ok=want_node)      # (collective whatever this rank wants)
            try:
                if gathered is not None:
                    Sg, wg, fg = gathered
                    n1 = int(np.asarray(S1).shape[0])
                    n2 = 0 if S2 is None else int(np.asarray(S2).shape[0])
                    part, jac, _, _ = _capi.evidence_feed_part_dev(Sg.data_ptr(), n1, ndim, Sg[n1:].data_ptr() if n2 else 0, n2, ndim, ndim, cov_mode, kmax,
                                                                   wg.data_ptr(), fg.data_ptr(), dist.get_rank(group), dist.get_world_size(group),
                                                                   device=dev, want_checksum=False)
                    csum = hostsum.value() if hostsum is not None else None
                else:
                    part, jac, _, csum = _capi.evidence_feed_part(S1, S2, ndim, cov_mode, kmax, weight, fs, dist.get_rank(group),
                                                                  dist.get_world_size(group), device=dev, want_checksum=self.verify)
            except Exception as exc:        # still join the collective (with a failure flag): the other ranks must not hang in it
                failed = exc
            return parallel.feed_part_reduce(part, csum, group, failed=failed), jac
        if self.devices not in (None, [0], (0,)):
            return None
        with self._scoped():
            dotp, jac, _ = _capi.evidence_feed(S1, S2, ndim, cov_mode, kmax, weight, fs)
        return dotp, jac

    def evidence_feed_batch(self, problems):
        """many problems (tuples with evidence_feed's arguments) in one library call; under torchrun
        the problems are farmed over the ranks instead (parallel.farm_evidence_feed).  Returns a list
        of (dotp, J) or None when this route does not apply."""
        from . import parallel
        if any(p[2] > 127 for p in problems):
            return None
        if parallel.is_distributed():
            return parallel.farm_evidence_feed(problems)
        from . import _capi
        with self._scoped():
            return [(dotp, jac) for dotp, jac, _ in _capi.evidence_feed_batch(problems, devices=self.devices)]

    def knn_dotp(self, X, Y, weight, fs, kmax, k0, want_dist=False):
        from . import parallel
        if parallel.is_distributed():
            return parallel.sharded_knn_dotp(X, Y, weight, fs, kmax, k0, want_dist=want_dist)
        from . import _capi
        with self._scoped():
            out = _capi.knn_dotp(X, Y, weight, fs, kmax, k0, return_dist=want_dist, devices=self.devices)
        return out if want_dist else (out, None)


class MCEvidence(object):
    def __init__(self, method, ischain=True, isfunc=None,
                 thinlen=0.0, burnlen=0.0,
                 split=False, s1frac=0.5, shuffle=True,
                 ndim=None, kmax=5,
                 priorvolume=1, debug=False,
                 nsample=None, covtype="single",
                 nbatch=1,
                 brange=None,
                 bscale="",
                 verbose=1, args={},
                 **gdkwargs):
        """Evidence estimation from MCMC chains (Heavens et al. 2017, arXiv:1704.03472).

        Parameters have the reference's meaning.  One keyword is added and consumed here
        (it is not forwarded to the chain reader): ``backend`` -- an object with a
        ``knn_dotp`` method; default ``HipBackend()``.

        method      chain root name / file name(s), or list/tuple/dict of chain arrays
                    (columns: weight, -lnL, parameters...)
        kmax        k-th nearest neighbours used: k = 1 .. kmax-1 (kmax >= 2 is enforced)
        ndim        number of leading parameter columns to use (default all)
        split       cross-evidence: neighbours of s1 points are searched in s2
        priorvolume prior volume; burnlen/thinlen as in the reference (files only)
        nbatch, brange, bscale   batched runs; only bscale='logpower' is supported
        """
        self.backend = gdkwargs.pop("backend", None) or HipBackend()
        self.logger = logger
        self.verbose = verbose
        self.debug = bool(debug or verbose > 1)
        level = logging.DEBUG if self.debug else (logging.INFO if verbose == 1 else logging.WARNING)
        if not logging.getLogger().handlers:
            logging.basicConfig(format=FORMAT)
        self.logger.setLevel(level)

        self.info = {}
        self.split = split
        self.covtype = covtype
        self.nbatch = nbatch
        self.brange = brange
        self.bscale = bscale if not isinstance(brange, int) else "constant"
        self.snames = ["s1", "s2"] if split else ["s1"]
        self.idbatch = np.arange(self.nbatch, dtype=int)
        self.powers = np.zeros((self.nbatch, len(self.snames)))
        self.bsize = np.zeros((self.nbatch, len(self.snames)), dtype=int)
        self.nchain = np.zeros((self.nbatch, len(self.snames)), dtype=int)
        self.kmax = max(2, kmax)                       # reference :694
        self.priorvolume = priorvolume
        self.ischain = ischain
        self.fname = method if isinstance(method, str) else None
        if not ischain:
            raise NotImplementedError("ischain=False (sampling from a model class) is not supported; "
                                      "pass the chain (the reference's own path is broken: undefined self.nsamples)")

        gdkwargs.setdefault("thinlen", thinlen)
        gdkwargs.setdefault("burnlen", burnlen)
        csplit = namedtuple("split_var", "split frac shuffle")(split=split, frac=s1frac, shuffle=shuffle)
        self.gd = MCSamples(method, csplit=csplit, debug=self.debug, **gdkwargs)

        if isfunc:
            self.gd.importance_sample(isfunc, name="s1")
            if self.split:
                self.gd.importance_sample(isfunc, name="s2")

        self.info["NparamsMC"] = self.gd.nparamMC
        self.info["Nsamples_read"] = self.gd.get_shape()[0]
        self.info["Nparams_read"] = self.gd.get_shape()[1]
        self.nsample = [self.gd.get_shape(name=s)[0] for s in self.snames]
        # the reference cuts s[:, 0:ndim] wherever it uses ndim (:893, :857): a slice, so an ndim beyond the
        # parameter columns silently means "all of them".  Clamped ONCE here, so the host route and the
        # device-feeder route (which passes ndim to the library) see the same number.
        self.ndim = self.gd.nparamMC if ndim is None else int(ndim)
        if self.ndim < 1:
            raise ValueError("ndim must be >= 1 (got %r)" % (ndim,))
        if self.ndim > self.gd.nparamMC:
            self.logger.warning("ndim=%s exceeds the %s parameter columns of the chain; using all of them" % (ndim, self.gd.nparamMC))
            self.ndim = self.gd.nparamMC
        self.info["NparamsCosmo"] = self.ndim
        self.info["Nsamples"] = ", ".join(str(x) for x in self.nsample)
        self.logger.info("chain array dimensions: %s x %s =" % (self.nsample, self.ndim))
        self.set_batch()

    def set_split(self, s1_rows, s2_rows):
        """Cross evidence of a CALLER-CHOSEN pair instead of the reference's random split (:221-226), e.g. two
        independent chains stacked in one array: s1 = rows ``s1_rows`` (the samples whose evidence sum is taken),
        s2 = rows ``s2_rows`` (the set the nearest neighbours are sought in).  Extension of the reference's API
        (SURVEY.md 8d, config C4); everything downstream is the reference's split=True path."""
        self.split = True
        self.snames = ["s1", "s2"]
        self.gd.set_split(s1_rows, s2_rows)
        self.nsample = [self.gd.get_shape(name=s)[0] for s in self.snames]
        self.info["Nsamples"] = ", ".join(str(x) for x in self.nsample)
        self.powers = np.zeros((self.nbatch, 2))
        self.bsize = np.zeros((self.nbatch, 2), dtype=int)
        self.nchain = np.zeros((self.nbatch, 2), dtype=int)
        self.set_batch()
        return self

    # ------------------------------------------------------------------ batching
    def summary(self):
        print()
        for k in ("ndim", "nsample", "kmax", "brange", "bsize", "powers", "nchain"):
            print("%s=%s" % (k, getattr(self, k)))
        print()

    def get_batch_range(self):
        if self.brange is None:
            return None, None
        lo, hi = float(np.min(self.brange)), float(np.max(self.brange))
        if lo == hi and self.nbatch > 1:
            raise ValueError("nbatch>1 but batch range is set to zero.")
        return lo, hi

    def set_batch(self, bscale=None):
        """brange None: one batch with every sample.  bscale='logpower': batch sizes
        10**linspace(min,max,nbatch) (reference :808-840; its 'linear' and 'constant'
        branches raise NameError / ValueError, here they raise ValueError)."""
        if bscale is None:
            bscale = self.bscale
        else:
            self.bscale = bscale
        if self.brange is None:
            self.bsize = self.brange
            for ix, nn in enumerate(self.nsample):
                self.nchain[0, ix] = nn
                self.powers[0, ix] = np.log10(nn)
        elif bscale == "logpower":
            lo, hi = self.get_batch_range()
            for ix, _ in enumerate(self.nsample):
                self.powers[:, ix] = np.linspace(lo, hi, self.nbatch)
                self.bsize[:, ix] = np.array([int(pow(10.0, x)) for x in self.powers[:, ix]])
            self.nchain = self.bsize
        else:
            raise ValueError("batching supports bscale='logpower' only (got %r)" % (bscale,))

    # ------------------------------------------------------------------ whitening
    def diagonalise_chain(self, s, eigenVec, eigenVal):
        """Rotate onto the covariance eigenvectors and scale to unit variance (:842-849)."""
        if (np.asarray(eigenVal)[: s.shape[1]] < 0).any():
            raise ValueError("math domain error: negative covariance eigenvalue (use fewer parameters, ndim)")
        return np.dot(s, eigenVec) / np.sqrt(np.asarray(eigenVal)[: s.shape[1]])[None, :]

    def get_covariance(self, s=None):
        """UNWEIGHTED sample covariance, its eigen-system and J = sqrt(det) (:851-882).
        With a negative eigenvalue: J = 1, posdef False."""
        if s is None:
            self.logger.info("Estimating covariance matrix using all chains")
            s, _, _ = self.gd.all_sample_arrays()
            s = s[:, 0:self.ndim]
        self.logger.info("covariance matrix estimated using nsample=%s" % len(s))
        cov = np.atleast_2d(np.cov(s.T))
        eigenVal, eigenVec = np.linalg.eig(cov)
        if (eigenVal < 0).any():
            self.logger.warning("Some of the eigenvalues of the covariance matrix are negative and/or complex: %s" % (eigenVal,))
            return {"cov": cov, "posdef": False, "J": 1, "eVec": eigenVec, "eVal": eigenVal}
        return {"cov": cov, "posdef": True, "J": math.sqrt(np.linalg.det(cov)), "eVec": eigenVec, "eVal": eigenVal}

    def get_samples(self, nsamples, istart=0, rand=False, name="s1", prewhiten=True):
        """Rows [istart, istart+nsamples) of a partition (all rows if nsamples == 0),
        cut to the first ndim parameters (:884-947)."""
        ntot = self.gd.get_shape(name)[0]
        s, lnp, w = self.gd.arrays(name)
        s = s[:, 0:self.ndim]
        if nsamples > 0:
            if rand and self.brange is not None:
                if nsamples > ntot:
                    raise ValueError("partition %s nsamples=%s, ntotal_chain=%s" % (name, nsamples, ntot))
                idx = rank0_draw(lambda: np.random.randint(0, high=ntot, size=nsamples))
            else:
                idx = np.arange(istart, nsamples + istart)
            s, lnp, w = s[idx, :], lnp[idx], w[idx]
        else:
            nsamples = ntot
        self.logger.info("getting samples for partition %s: nsamples=%s" % (name, nsamples))
        stat = {"J": 1, "eVec": None, "eVal": None}
        if prewhiten:
            cs = self.get_covariance(s=s)
            stat = {"J": cs["J"], "eVec": cs["eVec"], "eVal": cs["eVal"]}
            if cs["posdef"]:
                s = self.diagonalise_chain(s, cs["eVec"], cs["eVal"])
        return s, lnp, w, stat

    def _feed_problem(self, covtype, pos_lnp):
        """Inputs of the device-feeder route for this object: the problem tuple of
        ``_capi.evidence_feed`` and what ``_feed_finish`` needs (reference :1034-1064)."""
        s1, lnp, weight = self.gd.arrays("s1")
        s2 = self.gd.arrays("s2")[0] if self.split else None
        logL = -lnp if pos_lnp else lnp
        logLmax = np.amax(logL)
        fs = logL - logLmax
        problem = (s1, s2, self.ndim, 0 if covtype == "all" else 1, self.kmax,
                   np.asarray(weight, dtype=np.float64), np.asarray(fs, dtype=np.float64))
        return problem, (s1.shape[0], logLmax)

    def _feed_finish(self, ctx, dotp, Jacobian, logPriorVolume):
        """ln E_k from the reduced sums (reference :1120-1131); returns MLE[kmax]."""
        S, logLmax = ctx
        kmax = self.kmax
        k0 = 0 if self.split else 1
        SumW = np.sum(self.gd.data["s1"].adjusted_weights)
        mle = np.zeros(kmax)
        for k in range(k0, kmax):
            k_nn = k if k0 == 1 else k + 1
            mle[k] = math.log(SumW * (dotp[k] / (S * k_nn + 1.0)) * Jacobian) + logLmax - logPriorVolume
        return mle

    def _feed_route_applies(self, verbose, covtype):
        # cross evidence with covtype 'single' whitens s1 and s2 with DIFFERENT eigen-systems, so the
        # distances depend on the eigenvector order/sign conventions of the solver: that case keeps the
        # reference's own np.linalg.eig (host route)
        if self.split and covtype == "single":
            return False
        return self.brange is None and verbose <= 1 and covtype in ("all", "single")

    def _evidence_device_feeders(self, covtype, pos_lnp, logPriorVolume):
        """evidence() with get_covariance/diagonalise_chain done by the library too.  Same quantities as
        the host route below (reference :1034-1131); returns MLE[kmax] or None if the backend declines."""
        problem, ctx = self._feed_problem(covtype, pos_lnp)
        got = self.backend.evidence_feed(*problem)
        if got is None:
            return None
        return self._feed_finish(ctx, got[0], got[1], logPriorVolume)

    def _report(self, MLE, verbose, info):
        if verbose > 0:
            for k in range(1, self.kmax):
                self.logger.info("   ln(B)[k={}] = {}".format(k, MLE[k - 1] if self.brange is None else MLE[:, k - 1]))
        return (MLE, self.info) if info else MLE

    # ------------------------------------------------------------------ the estimator
    def evidence(self, verbose=None, rand=False, info=False, covtype="all",
                 profile=False, pvolume=None, pos_lnp=False,
                 nproc=-1, prewhiten=True):
        """ln-evidence for k = 1..kmax-1 nearest neighbours (auto), or k = 2..kmax (cross,
        ``split=True``).  Returns the array ``MLE[1:]`` like the reference (and the info
        dict if ``info=True``).  ``nproc``/``profile``/``prewhiten`` are accepted for
        signature compatibility (the reference ignores the last two as well)."""
        if verbose is None:
            verbose = self.verbose
        logPriorVolume = math.log(self.priorvolume if pvolume is None else pvolume)
        kmax, ndim = self.kmax, self.ndim
        MLE = np.zeros((self.nbatch, kmax))
        if covtype is None:
            covtype = self.covtype
        # ---- device-feeder route: covariance, whitening AND the hot path in one library call ----
        # (single batch, no per-neighbour debug output, a backend that offers it)
        if self._feed_route_applies(verbose, covtype) and hasattr(self.backend, "evidence_feed"):
            out = self._evidence_device_feeders(covtype, pos_lnp, logPriorVolume)
            if out is not None:
                return self._report(out[1:], verbose, info)

        if covtype == "all":
            covstat = self.get_covariance()
            Jacobian = covstat["J"]

        for ipow, nsample in zip(self.idbatch, self.nchain):
            S = int(nsample[0])
            samples, logL, weight, _ = self.get_samples(S, istart=0, rand=rand, prewhiten=False, name="s1")
            if covtype == "single":
                covstat = self.get_covariance(s=samples)
                Jacobian = covstat["J"]
            samples = self.diagonalise_chain(samples, covstat["eVec"], covstat["eVal"])
            if pos_lnp:
                logL = -logL
            logLmax = np.amax(logL)                         # renormalise against underflow (:1062-1064)
            fs = logL - logLmax

            if self.split:
                samples2, _, _, _ = self.get_samples(0, istart=0, rand=rand, prewhiten=False, name="s2")
                if covtype == "single":
                    covstat = self.get_covariance(s=samples2)    # s2's own eigen-system, J stays s1's (:1080-1086)
                samples2 = self.diagonalise_chain(samples2, covstat["eVec"], covstat["eVal"])
                self.logger.info("using XMCEvidence. NN distance is estimated using nsamples=(%s, %s)" % (S, samples2.shape[0]))
                k0 = 0
                refset = samples2
            else:
                k0 = 1                                      # the nearest "neighbour" is the point itself (:1099)
                refset = None

            # ---- the hot path: kNN search + volume/weight reduction on the GPU -------
            want_dist = verbose > 1
            dotp, dist = self.backend.knn_dotp(np.ascontiguousarray(samples), refset, np.asarray(weight, dtype=np.float64),
                                               np.asarray(fs, dtype=np.float64), kmax, k0, want_dist=want_dist)

            SumW = np.sum(self.gd.data["s1"].adjusted_weights)   # ALL of s1, also when batching (:1126)
            for k in range(k0, kmax):
                k_nn = k if k0 == 1 else k + 1
                amax = dotp[k] / (S * k_nn + 1.0)
                MLE[ipow, k] = math.log(SumW * amax * Jacobian) + logLmax - logPriorVolume
                if verbose > 1:
                    # the reference logs statistics.median(volume[:, k]) (:1143-1145); same quantity from the distances
                    medvol = float("nan")
                    if dist is not None:
                        lnc = 0.5 * ndim * math.log(math.pi) - math.lgamma(1.0 + 0.5 * ndim)
                        rmed = statistics.median(dist[:, k - k0])
                        medvol = math.exp(lnc + ndim * math.log(rmed)) if rmed > 0 else 0.0
                    self.logger.debug("k={},nsample={}, dotp={}, median_volume={}, a_max={}, MLE={}".format(
                        k, S, dotp[k], medvol, amax, MLE[ipow, k]))

        MLE = MLE[0, 1:] if self.brange is None else MLE[:, 1:]
        return self._report(MLE, verbose, info)


def evidence_many(mces, verbose=None, info=False, covtype="all", pvolume=None, pos_lnp=False, **kwargs):
    """``[m.evidence(...) for m in mces]`` with the device work of all objects in ONE batched library
    call (``mce_evidence_feed_batch_f64``).

    This is the reference's Planck driver pattern -- one ``MCEvidence(...).evidence(info=True)``
    per (data set, model, chain), farmed over MPI ranks (planck_mcevidence.py:306-348) -- where a
    single chain (6k-100k rows, D = 6-8) fills a fraction of the GPU.  Same arguments and the same
    per-object results as ``evidence()``; ``pvolume`` may be a sequence (one per object).  Objects
    the batched route does not cover (batching ranges, ``verbose > 1``, ``covtype`` other than
    'all'/'single', a backend without ``evidence_feed_batch``) are evaluated one by one.  Under
    ``torchrun`` the problems are farmed over the ranks and every rank returns all results."""
    mces = list(mces)
    pvols = list(pvolume) if isinstance(pvolume, (list, tuple, np.ndarray)) else [pvolume] * len(mces)
    if len(pvols) != len(mces):
        raise ValueError("pvolume: expected %d entries, got %d" % (len(mces), len(pvols)))
    results = [None] * len(mces)
    groups = {}                                   # backend -> [(index, ctx, problem)]
    for i, m in enumerate(mces):
        v = m.verbose if verbose is None else verbose
        ct = m.covtype if covtype is None else covtype
        if m._feed_route_applies(v, ct) and hasattr(m.backend, "evidence_feed_batch"):
            problem, ctx = m._feed_problem(ct, pos_lnp)
            groups.setdefault(id(m.backend), (m.backend, []))[1].append((i, ctx, problem))
    for backend, items in groups.values():
        got = backend.evidence_feed_batch([p for _, _, p in items])
        if got is None:
            continue
        for (i, ctx, _), (dotp, jac) in zip(items, got):
            m = mces[i]
            lpv = math.log(m.priorvolume if pvols[i] is None else pvols[i])
            v = m.verbose if verbose is None else verbose
            results[i] = m._report(m._feed_finish(ctx, dotp, jac, lpv)[1:], v, info)
    for i, m in enumerate(mces):
        if results[i] is None:
            results[i] = m.evidence(verbose=verbose, info=info, covtype=covtype, pvolume=pvols[i], pos_lnp=pos_lnp, **kwargs)
    return results
