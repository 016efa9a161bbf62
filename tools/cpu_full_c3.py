#!/usr/bin/env python
"""GPU box, host cores only: the FULL config-C3 pass of the reference's CPU path -- sklearn NearestNeighbors exactly as
MCEvidence.py:1093-1104 calls it on all 1M whitened rows + the volume/weight sum (:1107-1131) -- timed once and cached
(profiles/cpu_full_c3.json), so that bench.py can report |dlnE| of the GPU path against a same-node, full-size CPU run
without spending minutes of every bench run on it.   usage: python tools/cpu_full_c3.py -> gpurun_out/cpu_full_c3.json"""
import json, math, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import host_info
from mcevidence_amd.synth import gaussian_chain
from oracle import oracle_np as orc            # the CPU restatement (checker / baseline)
from sklearn.neighbors import NearestNeighbors

n, d, kmax = 1_000_000, 27, 10
chain = gaussian_chain(seed=3, n=n, d=d, cov="corr")
t_all = time.perf_counter()
ev = orc.covariance_eig(chain[:, 2:])
X = np.ascontiguousarray(orc.whiten(chain[:, 2:], ev["eVec"], ev["eVal"]))
weight, logL = chain[:, 0], -chain[:, 1]
logLmax = float(np.amax(logL)); fs = logL - logLmax
t0 = time.perf_counter()
nb = NearestNeighbors(n_neighbors=kmax + 1, metric="euclidean", leaf_size=20, algorithm="auto", n_jobs=-1).fit(X)
DkNN, _ = nb.kneighbors(X)
t_knn = time.perf_counter() - t0
dotp = orc.dotp_literal(DkNN, weight, fs, d, 1, kmax)
t_hot = time.perf_counter() - t0
lnE = orc.mle_from_dotp(dotp, n, 1, kmax, float(np.sum(weight)), ev["J"], logLmax, 0.0)
gold = [c for c in json.load(open(os.path.join(REPO, "tests", "golden", "evidence_big.json"))) if c["name"] == "auto_n1000000_d27_k10_C3"][0]
out = dict(config="C3", n=n, d=d, kmax=kmax, seconds=round(t_hot, 2), knn_seconds=round(t_knn, 2), queries_per_s=round(n / t_hot, 1),
           fit_method=str(nb._fit_method), lnE=[float(x) for x in lnE], max_abs_dlnE_vs_reference_golden=float(np.max(np.abs(np.array(lnE) - np.array(gold["lnE"])))),
           host=host_info(), what="sklearn NearestNeighbors(algorithm='auto', n_jobs=-1).fit(X).kneighbors(X) on all rows + NumPy volume/weight sum (MCEvidence.py:1093-1131)")
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "cpu_full_c3.json"), "w"), indent=1)
print(json.dumps(out))
