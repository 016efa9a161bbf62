"""Prior-volume bookkeeping from CosmoMC ``<root>.ranges`` or MontePython
``<dir>/log.param`` files (host side; mirrors ``params_info`` / ``get_prior_volume``
of ``/root/reference/MCEvidence.py:1195-1339``)."""
from __future__ import annotations

import glob
import logging
import re

import numpy as np

logger = logging.getLogger("mcevidence_amd")

#: parameter names treated as cosmological (the rest are nuisance); reference :84-86
cosmo_params_list = ["omegabh2", "omegach2", "theta", "tau", "omegak", "mnu", "meffsterile", "w", "wa",
                     "nnu", "yhe", "alpha1", "deltazrei", "Alens", "Alensf", "fdm", "logA", "ns", "nrun",
                     "nrunrun", "r", "nt", "ntrun", "Aphiphi"]


def iscosmo_param(p, cosmo_params=None):
    if cosmo_params is not None:
        cosmo_params_list.extend(cosmo_params)
    return p in cosmo_params_list


def _bound(tok, sign):
    """A .ranges bound: a float, or 'N' for unbounded (-> -inf / +inf)."""
    return sign * np.inf if tok == "N" else float(tok)


def params_info(fname, cosmo=False, volumes={}):
    """Names, bounds and prior volume of the sampled parameters.

    CosmoMC: every line of ``fname.ranges`` is ``name min max``; parameters with
    max == min are fixed and skipped; with ``cosmo=True`` only names in
    ``cosmo_params_list`` count.  MontePython: ``data.parameters['x'] = [mean, min, max,
    sigma, scale, role]`` lines of ``fname/log.param``; 'derived' entries are skipped and
    with ``cosmo=True`` only role 'cosmo' counts; unbounded (None) priors raise.
    Returns dict(name, min, max, range, str, ndim, nr_of_params, volume)."""
    par = {"name": [], "min": [], "max": [], "range": []}
    if glob.glob("{}*.ranges".format(fname)):
        logger.info("getting params info from COSMOMC file %s.ranges" % fname)
        with open(fname + ".ranges") as fh:
            for line in fh:
                tok = line.split()
                if len(tok) < 3 or tok[0].startswith("#"):
                    continue
                name, lo, hi = tok[0], _bound(tok[1], -1.0), _bound(tok[2], +1.0)
                if np.isclose(hi, lo) or (cosmo and not iscosmo_param(name)):
                    continue
                par["name"].append(name)
                par["min"].append(lo)
                par["max"].append(hi)
                par["range"].append(abs(hi - lo))
    elif glob.glob("{}/log.param".format(fname)):
        logger.info("getting params info from montepython log.params file")
        pat = re.compile(r"data\.parameters\[\s*['\"]([^'\"]+)['\"]\s*\]\s*=\s*\[(.*)\]")
        with open("{}/log.param".format(fname)) as fh:
            for line in fh:
                if "#" in line:
                    continue
                m = pat.search(line)
                if not m:
                    continue
                name = m.group(1)
                arr = [e.strip().strip("'\"") for e in m.group(2).split(",")]
                role = arr[5] if len(arr) > 5 else ""
                if role == "derived" or (cosmo and role != "cosmo"):
                    continue
                if arr[1] == "None" or arr[2] == "None":
                    raise Exception("Unbounded priors are not supported - please specify priors")
                lo, hi = float(arr[1]), float(arr[2])
                par["name"].append(name)
                par["min"].append(lo)
                par["max"].append(hi)
                par["range"].append(hi - lo)
    else:
        raise Exception("Could not read parameter volume from COSMOMC .ranges file or montepython log.param file")
    par["str"] = ",".join(par["name"])
    par["ndim"] = len(par["name"])
    par["nr_of_params"] = len(par["name"])
    par["volume"] = np.array(par["range"]).prod()
    return par


def get_prior_volume(args, **kwargs):
    """Prior volume from the chain's range files; also sets ``args.ndim`` (reference
    :1312-1339 -- whose interactive fallback is unreachable, so failures propagate)."""
    par = params_info(args.root_name, **kwargs)
    if getattr(args, "verbose", 0) > 1:
        print(par)
    args.ndim = par["ndim"]
    logger.info("getting prior volume using cosmomc *.ranges or montepython log.param outputs")
    logger.info("prior_volume=%s" % par["volume"])
    logger.info("Number of params to use: ndim=%s" % par["ndim"])
    return par["volume"]
