"""Seeded synthetic MCMC chains used by the parity tests, the golden-vector
generator and bench.py.

Column convention is the reference's (``/root/reference/MCEvidence.py:126-128``,
``:233-240``): column 0 = weight, column 1 = -ln L, columns 2.. = parameters.
The recipes follow SURVEY.md section 8(d); the ``gaussian`` recipe is the one the
reference's own demo uses (a normalised Gaussian, so the true evidence is 1 and
ln E = 0 with prior volume 1; ``/root/reference/examples.py:267-342``).

Everything is derived from ``numpy.random.default_rng(seed)`` in a fixed draw
order, so a fixture only has to store (recipe, seed, sizes).
"""
from __future__ import annotations

import math

import numpy as np

__all__ = ["gaussian_chain", "planck_like_chains", "write_cosmomc_chains", "CONFIGS", "config_chain"]


def gaussian_chain(seed, n, d, *, weights="unit", cov="unit", nextra=0, dtype=np.float64):
    """Chain of ``n`` draws from a normalised ``d``-dimensional Gaussian.

    Draw order (fixed): z[n,d] -> A[d,d] (only if cov='corr') -> integer weights
    (only if weights='int') -> nuisance columns (only if nextra>0).

    weights : 'unit' (all 1) or 'int' (uniform integers 1..5)
    cov     : 'unit' (theta = z) or 'corr' (theta = z @ A, A = I + 0.5*G)
    nextra  : extra nuisance parameter columns appended after the d parameters
              (the likelihood does not depend on them)
    Returns the [n, 2+d+nextra] chain array.
    """
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, d))
    lnl = -0.5 * np.einsum("ij,ij->i", z, z) - 0.5 * d * math.log(2.0 * math.pi)
    if cov == "corr":
        a = np.eye(d) + 0.5 * rng.standard_normal((d, d))
        theta = z @ a
        lnl = lnl - math.log(abs(np.linalg.det(a)))
    elif cov == "unit":
        theta = z
    else:
        raise ValueError("cov must be 'unit' or 'corr'")
    if weights == "int":
        w = rng.integers(1, 6, size=n).astype(np.float64)
    elif weights == "unit":
        w = np.ones(n)
    else:
        raise ValueError("weights must be 'unit' or 'int'")
    cols = [w, -lnl, theta]
    if nextra:
        cols.append(0.3 * rng.standard_normal((n, nextra)) + 1.0)
    return np.column_stack(cols).astype(dtype, copy=False)


#: rows per chain of the Planck base_plikHM_TT_lowTEB run
#: (``/root/reference/planck_fullgrid_R2/SingleChains/csv/mce_plikHM_TT_lowTEB.csv:2``)
PLANCK_ROWS = (6778, 6704, 6669, 6711)

#: (name, mean, sigma, prior_min, prior_max) of the 6 LCDM parameters of the stand-in
PLANCK_PARAMS = (
    ("omegabh2", 0.02222, 0.00023, 0.005, 0.1),
    ("omegach2", 0.1197, 0.0022, 0.001, 0.99),
    ("theta", 1.04085, 0.00047, 0.5, 10.0),
    ("tau", 0.078, 0.019, 0.01, 0.8),
    ("logA", 3.089, 0.036, 2.0, 4.0),
    ("ns", 0.9655, 0.0062, 0.8, 1.2),
)


def planck_like_chains(seed=1, rows=PLANCK_ROWS, nnuis=15):
    """Four CosmoMC-format chains standing in for config C1 (the real Planck
    chains are not shipped with the reference).  6 cosmological + ``nnuis``
    nuisance columns, integer weights 1+Poisson(3), mildly correlated.
    Returns (list_of_arrays, param_names, ranges) with ranges = [(name,lo,hi)].
    """
    rng = np.random.default_rng(seed)
    d = len(PLANCK_PARAMS)
    mix = np.eye(d) + 0.3 * rng.standard_normal((d, d))
    mu = np.array([p[1] for p in PLANCK_PARAMS])
    sg = np.array([p[2] for p in PLANCK_PARAMS])
    chains = []
    for n in rows:
        z = rng.standard_normal((n, d))
        theta = mu + (z @ mix) * sg
        nuis = 1.0 + 0.1 * rng.standard_normal((n, nnuis))
        neglnl = 0.5 * np.einsum("ij,ij->i", z, z) + 0.5 * np.einsum("ij,ij->i", nuis - 1.0, nuis - 1.0) / 0.01 + 5668.0
        w = 1.0 + rng.poisson(3.0, size=n)
        chains.append(np.column_stack([w, neglnl, theta, nuis]))
    names = [p[0] for p in PLANCK_PARAMS] + ["nuis%02d" % i for i in range(nnuis)]
    ranges = [(p[0], p[3], p[4]) for p in PLANCK_PARAMS] + [("nuis%02d" % i, 0.0, 2.0) for i in range(nnuis)]
    return chains, names, ranges


def write_cosmomc_chains(root, chains, ranges=None, fmt="%.10e"):
    """Write ``root_1.txt .. root_n.txt`` (+ ``root.ranges``) in CosmoMC layout,
    the on-disk format the reference reader globs for
    (``/root/reference/MCEvidence.py:590-596``, ``:1213-1230``)."""
    paths = []
    for i, c in enumerate(chains):
        p = "%s_%d.txt" % (root, i + 1)
        np.savetxt(p, c, fmt=fmt)
        paths.append(p)
    if ranges is not None:
        with open(root + ".ranges", "w") as fh:
            for name, lo, hi in ranges:
                hi_s = "N" if hi is None else repr(float(hi))
                fh.write("%-12s %r %s\n" % (name, float(lo), hi_s))
    return paths


#: BASELINE.json configs restated as recipes
CONFIGS = {
    "C1": dict(kind="planck_like", seed=1, n=sum(PLANCK_ROWS), d=6, kmax=2),
    "C2": dict(kind="gaussian", seed=2, n=100_000, d=6, kmax=4, cov="corr"),
    "C3": dict(kind="gaussian", seed=3, n=1_000_000, d=27, kmax=10, cov="corr"),
    "C4": dict(kind="gaussian_pair", seeds=(4, 5), n=1_000_000, d=15, kmax=4, cov="unit"),
    "C5": dict(kind="gaussian", seed=6, n=10_000_000, d=6, kmax=10, cov="corr"),
}


def config_chain(name, n=None):
    """The chain array of a BASELINE.json GPU config and its (s1_rows, s2_rows) split (None, None for auto
    evidence).  C4 is two independent chains stacked: s1 = the first, s2 = the second; ``n``
    overrides the rows per chain (reduced-size variants for tests)."""
    c = dict(CONFIGS[name])
    n = int(n or c["n"])
    if c["kind"] == "gaussian":
        return gaussian_chain(seed=c["seed"], n=n, d=c["d"], cov=c["cov"]), (None, None)
    if c["kind"] == "gaussian_pair":
        a, b = (gaussian_chain(seed=s, n=n, d=c["d"], cov=c["cov"]) for s in c["seeds"])
        return np.concatenate((a, b)), (np.arange(n), np.arange(n, 2 * n))
    raise ValueError("config_chain: %s is not an in-memory Gaussian config" % name)
