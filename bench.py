#!/usr/bin/env python
"""bench.py -- headline benchmark of the kNN evidence hot path on MI355X.

Metric (BASELINE.json): kNN queries/s (+ |dlnE|) at N = 1M, D = 27, kmax = 10 (config C3:
seeded synthetic Gaussian chain, `mcevidence_amd.synth.CONFIGS['C3']`), inputs resident in
HBM when the timed region starts.  One "step" = one full pass of the hot path (pack ->
kNN search -> merge -> volume/weight reduction -> dotp[kmax]) over this rank's query
shard; with --gpus N the 1M queries are sharded over N ranks (reference set replicated),
one RCCL all-reduce of kmax doubles per step: strong scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 1000000] [--d 27] [--kmax 10] [--mode 0|1]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel, its launch duration
measured with HIP events on the launch stream.  Default (--mode 0): knn_f16_kernel, the fp16-MFMA
all-pairs filter + exact fp64 refine -- algorithmic flops per launch = nq * nr * 2 * 16*KST (the
augmented product |y^|^2 - 2 x^.y^ the MFMA evaluates) against the 2.5 PFLOP/s dense
fp16 peak.  At N = 1 (queries and references are one resident buffer) the library takes the symmetric
sweep: each pair of rows is multiplied once and gated for both of its sides, so the
kernel executes about half of those flops; `roofline.note` states the executed figure.  --mode 1: knn_mfma_kernel, the pure fp64 MFMA sweep -- nq * nr * 2 * 4*KS flops against
78.6 TFLOP/s (tools/mfma_f64_peak.hip reaches 72-74 on this part).  `traffic` = HBM bytes per launch
from the committed rocprofv3 PMC passes (profiles/).
`cpu_baseline` = the reference's own CPU path (scikit-learn NearestNeighbors with its
default algorithm + NumPy reduction, via the oracle) on a bounded query sample, rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 matrix = vector peak
F16_PEAK_TFLOPS = 2500.0     # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)


def hbm_traffic_from_profile(kernel_desc):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/<round>/pmc_summary.csv: FETCH_SIZE/WRITE_SIZE in KB, separate passes; FETCH_SIZE
    doubled per MI355X_MICROARCH.md's gfx950 note).  None when no profile of this kernel exists."""
    import glob
    import re
    m = re.match(r"(\w+)<\w+=(\d+),\w+=(\d+)>", kernel_desc)
    if not m:
        return None
    name, p1, p2 = m.groups()
    if name == "knn_f16_kernel":
        # <KST, KCAP, PRUNE, LOWER, SYM>: the symmetric sweep is the SYM = 2 instantiation (round-1 profiles: no SYM parameter)
        sy = "2" if " symmetric" in kernel_desc else "0"
        wants = ("%sILi%sELi%sELb0ELb0ELi%sE" % (name, p1, p2, sy),) + (("%sILi%sELi%sELb0ELb0EE" % (name, p1, p2),) if sy == "0" else ())
    else:
        wants = ("%s<%s, %s>" % (name, p1, p2), "%sILi%sELi%sE" % (name, p1, p2))     # demangled / mangled spelling
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*", "pmc_summary.csv")), reverse=True):
        fetch = write = None
        for ln in open(f):
            c = ln.strip().split(",")
            if len(c) >= 4 and any(w in ln for w in wants):
                name, val, nd = c[-3], float(c[-2]), int(c[-1])
                if name == "FETCH_SIZE":
                    fetch = val / nd
                if name == "WRITE_SIZE":
                    write = val / nd
        if fetch is not None and write is not None:
            return dict(bytes=(2.0 * fetch + write) * 1024.0, source=os.path.relpath(f, REPO),
                        note="per launch; (2*FETCH_SIZE + WRITE_SIZE) KB, separate --pmc passes")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=27)
    ap.add_argument("--kmax", type=int, default=10)
    ap.add_argument("--cpu-sample", type=int, default=20000, help="queries timed on the CPU baseline (0 = skip)")
    ap.add_argument("--mode", type=int, default=0, help="0 auto (fp16 filter + fp64 refine), 1 fp64 MFMA sweep")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 or world > 1:
        assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
        # (MCE_BENCH_BACKEND=gloo MCE_BENCH_ONE_DEVICE=1: functional check of the N>1 path on a 1-GPU box)
        if os.environ.get("MCE_BENCH_ONE_DEVICE") == "1":
            local = 0
        torch.cuda.set_device(local)
        backend = os.environ.get("MCE_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from mcevidence_amd import _capi, parallel
    _capi.set_search_mode(a.mode)
    from mcevidence_amd.synth import gaussian_chain
    import mcevidence_amd as pkg

    # ---- synthetic chain (config C3 recipe) + host-side feeders (whitening etc.) ----
    n, d, kmax = a.n, a.d, a.kmax
    chain = gaussian_chain(seed=3, n=n, d=d, cov="corr")
    mce = pkg.MCEvidence([chain], kmax=kmax, verbose=0)
    cov = mce.get_covariance()
    s1, logL, weight, _ = mce.get_samples(n, prewhiten=False)
    Xh = np.ascontiguousarray(mce.diagonalise_chain(s1, cov["eVec"], cov["eVal"]))
    logLmax = float(np.amax(logL))
    fsh = logL - logLmax
    SumW = float(np.sum(weight))

    lo, hi = parallel.shard_bounds(n, world, rank)
    nq = hi - lo
    K = kmax - 1
    X = torch.from_numpy(Xh).to(dev)                       # reference set, replicated
    Xq = X[lo:hi]                                          # this rank's query shard (a view)
    w = torch.from_numpy(np.ascontiguousarray(weight[lo:hi])).to(dev)
    fs = torch.from_numpy(np.ascontiguousarray(fsh[lo:hi])).to(dev)
    wsb = _capi.knn_workspace_bytes(nq, n, d, K) + _capi.dotp_workspace_bytes(nq, kmax)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dotp = torch.zeros(kmax, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream()

    w_all = torch.from_numpy(np.ascontiguousarray(weight)).to(dev) if world > 1 else w
    fs_all = torch.from_numpy(np.ascontiguousarray(fsh)).to(dev) if world > 1 else fs
    if world > 1:
        wsb = _capi.knn_workspace_bytes(n, n, d, K) + _capi.dotp_workspace_bytes(n, kmax)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def step():
        if world > 1:
            # this rank's part of the queries (the library picks the partition: rows [lo, hi) for the sweep)
            _capi.knn_dotp_part_dev(X.data_ptr(), n, d, kmax, rank, world, w_all.data_ptr(), fs_all.data_ptr(),
                                    dotp.data_ptr(), ws.data_ptr(), wsb, stream.cuda_stream)
        else:
            _capi.knn_dotp_dev(Xq.data_ptr(), nq, X.data_ptr(), n, d, kmax, 1, lo, w.data_ptr(), fs.data_ptr(),
                               dotp.data_ptr(), 0, ws.data_ptr(), wsb, stream.cuda_stream)
        if world > 1:
            dist.all_reduce(dotp, op=dist.ReduceOp.SUM)    # the single collective of the path

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    _capi.set_profiling(True)          # hipEvent brackets around each knn_mfma_kernel launch, on its stream
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = _capi.last_kernel_ms()   # mean over the K timed launches (read after the timed region)
    _capi.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_step = elapsed / a.steps * 1e3
    # ---- ln E from the device result, and parity against the golden / CPU sample ----
    dp = dotp.cpu().numpy()
    lnE = np.array([math.log(SumW * dp[k] / (n * k + 1.0) * cov["J"]) + logLmax - math.log(1.0) for k in range(1, kmax)])

    out = None
    if rank == 0:
        kdesc = _capi.last_kernel()
        if kdesc.startswith("knn_f16"):
            # fp16-MFMA filter sweep: algorithmic flops = the padded augmented dot product it evaluates
            # for every (query, reference) pair, 2 * 16*KST per pair; peak = dense fp16 MFMA.
            kst = (d + 3 + 15) // 16
            flops = float(nq) * n * 2.0 * 16 * kst
            peak, note = F16_PEAK_TFLOPS, "fp16 MFMA pre-filter over all pairs (2*16*KST flop/pair) + exact fp64 refine of the survivors"
            if " symmetric" in kdesc:
                # d(i,j) = d(j,i): block a (512 rows) multiplies only the tiles of blocks 0..a, each tile gated for both
                # of its sides.  `achieved` stays the ALGORITHMIC all-pairs figure over the sweep kernel's duration (the
                # contract's definition); what the matrix cores execute is about half of it:
                nb = (n + 511) // 512
                executed = float(512) * 512 * nb * (nb + 1) / 2.0 * 2.0 * 16 * kst
                note = ("symmetric sweep: every pair of rows multiplied once (blocks 0..a per block a of 512 rows) and gated for both sides; "
                        "algorithmic flops = all pairs (2*16*KST flop/pair), executed MFMA flops in the sweep kernel = %.3g "
                        "(%.1f TFLOP/s); the prepass, repair and merge kernels of the same search are in ms_per_step, not in kernel_ms" %
                        (executed, executed / (kern_ms * 1e-3) / 1e12))
        else:
            KS = (d + 1 + 3) // 4
            flops = float(nq) * n * 2.0 * 4 * KS
            peak, note = FP64_PEAK_TFLOPS, "fp64 MFMA sweep (2*4*KS flop/pair), fp64 MFMA-bound (SURVEY 8d)"
        achieved = flops / (kern_ms * 1e-3) / 1e12
        traffic = hbm_traffic_from_profile(kdesc)
        roof = dict(bound="mfma", achieved=round(achieved, 3), peak=peak, unit="TFLOP/s",
                    frac=round(achieved / peak, 4), traffic=(traffic or {}).get("bytes"), traffic_source=(traffic or {}).get("source"),
                    traffic_note=(traffic or {}).get("note"),
                    hbm_gbps=(round(traffic["bytes"] / (kern_ms * 1e-3) / 1e9, 1) if traffic else None),
                    hbm_frac_of_8TBps=(round(traffic["bytes"] / (kern_ms * 1e-3) / 8e12, 4) if traffic else None),
                    kernel_ms=round(kern_ms, 3), kernel=kdesc, algorithmic_flops_per_launch=flops, note=note,
                    fp64_equivalent_tflops=round(float(nq) * n * 2.0 * 4 * ((d + 4) // 4) / (kern_ms * 1e-3) / 1e12, 2))
        cpu = None
        dlnE = None
        if a.cpu_sample > 0 and world == 1:         # CPU baseline: rank 0, N=1 only
            from oracle import oracle_np as orc                 # checker / baseline only
            rng = np.random.default_rng(0)
            rows = np.sort(rng.choice(n, size=min(a.cpu_sample, n), replace=False))
            t1 = time.perf_counter()
            dsk, _ = orc.knn_sklearn(Xh[rows], Xh, kmax + 1)  # the reference's exact call (MCEvidence.py:1093-1104)
            full = orc.dotp_literal(dsk, weight[rows], fsh[rows], d, 1, kmax)
            t_cpu = time.perf_counter() - t1
            cpu = dict(value=round(len(rows) / t_cpu, 1), unit="queries/s", cores=len(os.sched_getaffinity(0)), kind="port",
                       sample="%d random query rows of the same chain against the full %d-row reference set; sklearn NearestNeighbors(algorithm='auto', n_jobs=-1) + NumPy volume/weight sum" % (len(rows), n))
            # row-level parity on the sample: distances of the sampled rows, GPU vs CPU
            dg, _ = _capi.knn(Xh[rows], Xh, kmax + 1, self_mode=_capi.SELF_NONE)
            rel = float(np.max(np.abs(dg[:, 1:kmax] - dsk[:, 1:kmax]) / dsk[:, 1:kmax]))
            cpu["max_rel_dist_err_vs_gpu"] = rel
        # whole MCEvidence(...).evidence() call from host arrays (device feeders + H2D + hot path): the
        # PCIe-inclusive figure, reported next to `value` (which is HBM-resident), N=1 only
        e2e = None
        if world == 1:
            mce.evidence()
            t2 = time.perf_counter()
            lnE_e2e = mce.evidence()
            e2e = dict(seconds=round(time.perf_counter() - t2, 4), queries_per_s=round(n / (time.perf_counter() - t2), 1),
                       max_abs_dlnE_vs_resident_path=float(np.max(np.abs(lnE_e2e - lnE))))
        gold = os.path.join(REPO, "tests", "golden", "evidence_big.json")
        if (n, d, kmax) == (1_000_000, 27, 10) and os.path.exists(gold):
            for c in json.load(open(gold)):
                if c["name"] == "auto_n1000000_d27_k10_C3":
                    dlnE = float(np.max(np.abs(lnE - np.array(c["lnE"]))))
        out = dict(metric="knn_queries_per_sec", value=round(n / (ms_step * 1e-3), 1), unit="queries/s", n_gpus=world,
                   steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True,
                   scaling="strong", vs_baseline=None, dtype="f64" if a.mode == 1 else "f16 filter + f64 refine (exact f64 results)", data="synthetic",
                   config=dict(workload="C3: auto-evidence, seeded Gaussian chain N=%d D=%d kmax=%d (K=%d true neighbours/query), query-sharded over %d GPU(s)" % (n, d, kmax, K, world),
                               N=n, D=d, kmax=kmax, queries_per_rank=nq),
                   max_abs_dlnE_vs_reference=dlnE, lnE=[round(float(x), 10) for x in lnE],
                   roofline=roof, cpu_baseline=cpu, evidence_call_from_host=e2e)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
