#!/usr/bin/env python3
"""Randomised shapes through the C ABI against the CPU oracle (development tool; run on a GPU box).

    python tools/fuzz_shapes.py --seconds 600 --seed 1 [--max-rows 60000] [--min-rows 1] [--out gpurun_out/fuzz.json]

The parity tests pin chosen shapes; this draws them: sizes from one row to --max-rows, d from 1 to 140, K from 1 to 40, separate
and shared sets, with and without the self row, every process-wide search / prune / symmetric mode, and data that is meant to hurt
(exact duplicates, lattices full of ties, a far-away mean, tiny and huge scales, strongly anisotropic columns, clusters).  Every
draw goes through `mce_knn_f64` (distances + row numbers) and every third one through `mce_knn_dotp_f64` / the partitioned entry
point as well, and is compared with the oracle's exact direct-difference search:

 * row numbers equal, or -- where the oracle's own distances tie exactly -- the returned row at that exact distance;
 * distances within 1e-10 relative (tests/helpers.py DIST_RTOL), ascending;
 * sums within 1e-11 relative of the oracle's literal reduction, parts adding up to the whole.

A mismatch is printed with everything needed to replay it (`--replay '<json>'`) and counted; the summary names every kernel
variant the draws reached.  Exit code = number of mismatching draws (0: clean).  The oracle is the checker here, as in tests/."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from oracle import oracle_np as orc          # noqa: E402   (checker only)
from mcevidence_amd import _capi             # noqa: E402

DIST_RTOL = 1e-10
KINDS = ("gauss", "gauss", "aniso", "dups", "lattice", "offset", "tiny", "huge", "clusters", "line")


def make_rows(rng, kind, n, d):
    if kind == "gauss":
        return rng.standard_normal((n, d))
    if kind == "aniso":
        return rng.standard_normal((n, d)) * np.exp(rng.uniform(-6, 6, d))
    if kind == "dups":                      # every row several times over
        base = rng.standard_normal((max(1, n // 5), d))
        return base[rng.integers(0, base.shape[0], n)]
    if kind == "lattice":                   # integer lattice: distances tie massively
        return rng.integers(0, 4, (n, d)).astype(np.float64)
    if kind == "offset":                    # the mean far from the origin (GEMM-form keys lose digits there)
        return rng.standard_normal((n, d)) + 1.0e6 * rng.standard_normal(d)
    if kind == "tiny":
        return rng.standard_normal((n, d)) * 1e-150
    if kind == "huge":
        return rng.standard_normal((n, d)) * 1e140
    if kind == "clusters":
        c = rng.standard_normal((8, d)) * 50.0
        return c[rng.integers(0, 8, n)] + rng.standard_normal((n, d)) * np.exp(rng.uniform(-8, 0))
    if kind == "line":                      # all rows on one line: k-d boxes degenerate
        return rng.standard_normal((n, 1)) * rng.standard_normal((1, d))
    raise ValueError(kind)


def log_int(rng, lo, hi):
    return int(round(np.exp(rng.uniform(np.log(lo), np.log(hi + 1))))) if hi > lo else lo


def draw(rng, max_rows, min_rows=1, d_range=None):
    dsel = rng.integers(0, 10)
    d = int(rng.integers(1, 9)) if dsel < 4 else int(rng.integers(9, 33)) if dsel < 6 else int(rng.integers(33, 64)) if dsel < 7 else \
        int(rng.integers(64, 128)) if dsel < 9 else int(rng.integers(128, 141))
    if d_range:
        d = int(rng.integers(d_range[0], d_range[1] + 1))
    K = int(rng.integers(1, 17)) if rng.random() < 0.7 else int(rng.integers(17, 33)) if rng.random() < 0.8 else int(rng.integers(33, 41))
    budget = max_rows if d < 64 else max(2000, max_rows // 3)
    same = rng.random() < 0.55
    lo = max(1, min(min_rows, budget))
    nr = max(log_int(rng, lo, budget), K + 1)
    if rng.random() < 0.15 and min_rows <= 1:
        nr = max(K + 1, min(nr, K + int(rng.integers(0, 3))))            # barely enough reference rows
    nq = nr if same else log_int(rng, lo, budget)
    self_mode = (1 if rng.random() < 0.3 else 2) if same else 0
    if self_mode == 2 and nr <= K:
        nr = nq = K + 1
    return dict(d=d, K=K, nq=int(nq), nr=int(nr), same=bool(same), self_mode=int(self_mode), kind=str(KINDS[rng.integers(0, len(KINDS))]),
                search_mode=int(rng.choice([0, 0, 0, 1, 2])), prune_mode=int(rng.choice([0, 0, 1, 2])), sym_mode=int(rng.choice([0, 0, 1, 2])),
                verify=int(rng.choice([-1, 0, 64])), seed=int(rng.integers(0, 2**31)))


def run_case(c, with_sums):
    rng = np.random.default_rng(c["seed"])
    Y = np.ascontiguousarray(make_rows(rng, c["kind"], c["nr"], c["d"]))
    X = Y if c["same"] else np.ascontiguousarray(make_rows(rng, c["kind"], c["nq"], c["d"]))
    K, sm = c["K"], c["self_mode"]
    problems = []
    opt = _capi.Options(search_mode=c["search_mode"], prune_mode=c["prune_mode"], sym_mode=c["sym_mode"], verify=c["verify"])
    try:
        dist, idx = _capi.knn(X, Y, K, self_mode=sm, options=opt)
    except Exception as exc:
        return ["%s: %s" % (type(exc).__name__, exc)], None
    kern = _capi.last_kernel()
    od, oi = orc.knn_brute(X, Y, K, self_mode=2 if sm == 2 else 0)
    if sm == 1:
        # (self row included: the search returns it first, at distance 0 exactly; among exact duplicates the oracle may name another)
        if not np.all(dist[:, 0] == 0.0):
            problems.append("self column not exactly 0")
    if not np.all(np.diff(dist, axis=1) >= 0):
        problems.append("distances not ascending")
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(dist - od) / np.where(od != 0, np.abs(od), 1.0)
    rel = np.where(od == 0, np.abs(dist), rel)
    # The fp64 sweeps (search mode 1; lists longer than 16 at d >= 64; d >= 128) SELECT on GEMM-form keys |x|^2 + |y|^2 - 2 x.y, good to
    # ~eps R^2 absolute (R: the set's extent about its mean), and report exact distances of what they selected: where neighbours lie
    # closer together than that -- d = 1 with clusters 1e6 spacings apart -- the K-th may be swapped for an almost equally near row
    # (docs/design/sweep_f64.md; the default filter path has no such limit).  Such draws are counted, not failed.
    gemm = ("knn_mfma_kernel" in kern) or ("knn_long_kernel" in kern)
    allow = 0.0
    if gemm:
        mu = Y.mean(axis=0)
        allow = 512.0 * np.finfo(np.float64).eps * max(float(np.max(np.einsum("ij,ij->i", Y - mu, Y - mu))), float(np.max(np.einsum("ij,ij->i", X - mu, X - mu))))
    key_res = False
    if not np.all(rel <= DIST_RTOL):
        off = ~(rel <= DIST_RTOL)
        if gemm and np.all(np.abs(dist[off] ** 2 - od[off] ** 2) <= allow):
            key_res = True
        else:
            r, k = np.unravel_index(int(np.nanargmax(np.where(np.isnan(rel), np.inf, rel))), rel.shape)
            problems.append("distance off: row %d col %d got %r want %r" % (r, k, dist[r, k], od[r, k]))
    bad = idx != oi
    if np.any(bad):
        # a different row is fine only at an exactly tied distance: the row named must lie at the oracle's distance for that column
        rows, cols = np.nonzero(bad)
        diff = X[rows] - Y[idx[rows, cols]]
        dd = np.sqrt(np.einsum("ij,ij->i", diff, diff))
        ok = np.abs(dd - od[rows, cols]) <= 4e-16 * np.maximum(od[rows, cols], 1e-300) + 0.0
        if gemm and not np.all(ok):
            near = np.abs(dd ** 2 - od[rows, cols] ** 2) <= allow
            key_res = key_res or bool(np.any(near & ~ok))
            ok |= near
        if sm == 2:
            ok &= idx[rows, cols] != rows
        if not np.all(ok):
            j = int(np.nonzero(~ok)[0][0])
            problems.append("row number off: query %d col %d got %d (d=%r) want %d (d=%r)" % (rows[j], cols[j], idx[rows[j], cols[j]], dd[j], oi[rows[j], cols[j]],
                                                                                                od[rows[j], cols[j]]))
        # ... and a row may not be named twice for one query
        srt = np.sort(idx, axis=1)
        if np.any(srt[:, 1:] == srt[:, :-1]):
            problems.append("a reference row named twice for one query")
    if with_sums and sm != 1 and c["kind"] not in ("tiny", "huge") and c["d"] <= 150:          # (beyond: the oracle's literal r^d / Gamma overflows)
        k0 = 1 if sm == 2 else 0
        kmax = K + k0
        w = rng.integers(1, 6, c["nq"]).astype(np.float64)
        fs = -rng.random(c["nq"]) * 30.0
        want = orc.dotp_literal(np.concatenate([np.zeros((c["nq"], k0)), od], axis=1) if k0 else od, w, fs, c["d"], k0, kmax)
        with _capi.options(search_mode=c["search_mode"], prune_mode=c["prune_mode"], sym_mode=c["sym_mode"], verify=c["verify"]):
            try:
                got = _capi.knn_dotp(X, None if c["same"] else Y, w, fs, kmax, k0)
                sel = slice(k0, kmax)
                fin = np.isfinite(want[sel]) & (np.abs(want[sel]) > 1e-290)          # (a sum in the denormal range has no 1e-11 to compare)
                if not np.allclose(got[sel][fin], want[sel][fin], rtol=1e-11, atol=0.0):
                    problems.append("sums off: got %r want %r" % (got[sel].tolist(), want[sel].tolist()))
                if sm == 2 and c["nq"] >= 2:
                    nparts = int(rng.integers(2, 5))
                    tot = sum(_capi.knn_dotp_part(Y, w, fs, kmax, p, nparts) for p in range(nparts))
                    if not np.allclose(tot[sel][fin], want[sel][fin], rtol=1e-11, atol=0.0):
                        problems.append("parts (%d) do not add up: got %r want %r" % (nparts, tot[sel].tolist(), want[sel].tolist()))
                    kern += " | part: " + _capi.last_kernel()
            except Exception as exc:
                problems.append("sums: %s: %s" % (type(exc).__name__, exc))
    return problems, kern + (" | key-resolution" if key_res else "")


def kernel_family(k):
    if not k:
        return "?"
    name = k.split(" grid=")[0]
    extra = [w for w in ("pruned", "symmetric", "panel-kernel", "two passes", "wide", "tail", "chains", "pairs-once") if w in k]
    return name.split(" ")[0] + ((" [" + ",".join(extra) + "]") if extra else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-rows", type=int, default=60000)
    ap.add_argument("--min-rows", type=int, default=1, help="large draws only: several query blocks, reference splits, the trailing partial round")
    ap.add_argument("--dims", default=None, help="lo,hi: every draw's d from this range (e.g. 128,600: the long-row sweep)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--replay", default=None, help="one case as JSON (printed by a failing run)")
    a = ap.parse_args()
    if a.replay:
        c = json.loads(a.replay)
        problems, kern = run_case(c, True)
        print(json.dumps(dict(case=c, kernel=kern, problems=problems)))
        return 1 if problems else 0
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    n, failures, families, refused, keyres = 0, [], {}, {}, []
    while time.time() - t0 < a.seconds:
        c = draw(rng, a.max_rows, a.min_rows, [int(v) for v in a.dims.split(",")] if a.dims else None)
        problems, kern = run_case(c, n % 3 == 0)
        n += 1
        if kern is None:
            # a refusal with the library's own message (K beyond MCE_MAX_K, K > usable rows ...) is an answer, not a mismatch --
            # unless the oracle could do it and the header says the library can
            msg = problems[0]
            legal = c["K"] <= 32 and c["nr"] - (1 if c["self_mode"] == 2 else 0) >= c["K"]
            if legal:
                failures.append(dict(case=c, problems=problems))
                print("MISMATCH " + json.dumps(dict(case=c, problems=problems)), flush=True)
            else:
                refused[msg.split(":")[0]] = refused.get(msg.split(":")[0], 0) + 1
            continue
        if kern.endswith(" | key-resolution"):
            kern = kern[:-len(" | key-resolution")]
            keyres.append(dict(d=c["d"], K=c["K"], kind=c["kind"], kernel=kernel_family(kern.split(" | part: ")[0])))
        fam = kernel_family(kern.split(" | part: ")[0])
        families[fam] = families.get(fam, 0) + 1
        if " | part: " in kern:
            pf = "part -> " + kernel_family(kern.split(" | part: ")[1])
            families[pf] = families.get(pf, 0) + 1
        if problems:
            failures.append(dict(case=c, kernel=kern, problems=problems))
            print("MISMATCH " + json.dumps(dict(case=c, kernel=kern, problems=problems)), flush=True)
    summary = dict(draws=n, seconds=round(time.time() - t0, 1), seed=a.seed, max_rows=a.max_rows, min_rows=a.min_rows, dims=a.dims, mismatches=len(failures), refused_out_of_range=refused,
                   gemm_form_draws_with_a_neighbour_swapped_within_the_keys_resolution=keyres,
                   kernels_reached=dict(sorted(families.items(), key=lambda kv: -kv[1])), failures=failures[:50], library_source_hash=_capi.source_hash())
    print(json.dumps(summary), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(summary, open(a.out, "w"), indent=1)
    return min(len(failures), 100)


if __name__ == "__main__":
    sys.exit(main())
