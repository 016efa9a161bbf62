"""Command line: ``python MCEvidence.py <root> [flags]`` -- same flags as the reference
CLI (``/root/reference/MCEvidence.py:1342-1473``)."""
from __future__ import annotations

import logging
import sys
from argparse import ArgumentParser

from . import prior
from .evidence import MCEvidence

desc = "Planck Chains MCEvidence. Returns the log Bayesian Evidence computed using the kth NN"
cite = """
**
When using this code in published work, please cite the following paper: **
Heavens et. al. (2017)
Marginal Likelihoods from Monte Carlo Markov Chains
https://arxiv.org/abs/1704.03472
"""


def build_parser(prog=None):
    p = ArgumentParser(prog=prog, add_help=True, description=desc, epilog=cite)
    p.add_argument("root_name", help="Root filename for MCMC chains")
    p.add_argument("-k", "--kmax", dest="kmax", default=2, type=int, help="maximum k of the k-th nearest neighbour")
    p.add_argument("-ic", "--idchain", dest="idchain", default=0, type=int,
                   help="Which chain to use - e.g. 1 means read only *_1.txt (default: all available)")
    p.add_argument("-np", "--ndim", dest="ndim", default=None, type=int, help="How many parameters to use (default: all)")
    p.add_argument("--paramsfile", dest="paramsfile", default="", type=str,
                   help="text file with additional parameter names to consider cosmological")
    p.add_argument("--burn", "--burnlen", dest="burnlen", default=0, type=float,
                   help="Burn-in length or fraction; burnlen<1 is a fraction, e.g. 0.3 = 30%%")
    p.add_argument("--thin", "--thinlen", dest="thinlen", default=0, type=float,
                   help="Thinning: 0<thinlen<1 Poisson-resampled weights; thinlen>1 weighted thinning")
    p.add_argument("-vb", "--verbose", dest="verbose", default=1, type=int, help="0: WARNINGS, 1: INFO, 2: DEBUG")
    p.add_argument("-pv", "--pvolume", dest="priorvolume", default=None, type=float,
                   help="prior volume to use; if *.ranges exists the volume estimated from it is used")
    p.add_argument("--allparams", action="store_true", help="use all parameters, not only the cosmological ones")
    p.add_argument("--cross", action="store_true",
                   help="split the chain(s) in two and estimate the cross evidence (otherwise auto evidence)")
    return p


def main(argv=None):
    args = build_parser(prog="MCEvidence.py").parse_args(argv)
    if args.paramsfile:
        with open(args.paramsfile) as fh:
            new = [ln.strip() for ln in fh if ln.strip() and "#" not in ln]
        print("adding the following names to the cosmological parameter list:", new)
        for n in new:
            if n not in prior.cosmo_params_list:
                prior.cosmo_params_list.append(n)
    prior_volume = prior.get_prior_volume(args, cosmo=not args.allparams)
    logging.getLogger("mcevidence_amd").setLevel(
        logging.DEBUG if args.verbose > 1 else (logging.INFO if args.verbose == 1 else logging.WARNING))
    print()
    print("Using file: ", args.root_name)
    mce = MCEvidence(args.root_name, split=args.cross, ndim=args.ndim, priorvolume=prior_volume,
                     idchain=args.idchain, kmax=args.kmax, verbose=args.verbose, burnlen=args.burnlen,
                     thinlen=args.thinlen)
    out = mce.evidence()
    print("* ln(B)[k] is the natural logarithm of the Baysian evidence estimated using the kth Nearest Neighbour.")
    print("")
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
