#!/usr/bin/env python
"""bench.py -- headline benchmark of the kNN evidence hot path on MI355X.

Metric (BASELINE.json): kNN queries/s (+ |dlnE|) at N = 1M, D = 27, kmax = 10 (config C3: seeded synthetic Gaussian
chain, `mcevidence_amd.synth.CONFIGS['C3']`), inputs resident in HBM when the timed region starts.  One "step" = one full
pass of the hot path (pack -> kNN search -> merge -> volume/weight reduction -> dotp[kmax]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 1000000] [--d 27] [--kmax 10] [--mode 0|1] [--no-extras]

`--gpus N` with N > 1 works both ways: started by a launcher (`python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N ...`: WORLD_SIZE is set, this process is one rank) or plainly (`python bench.py --gpus N`: the parent
starts the N ranks itself as child processes BEFORE anything touches the GPU -- their output, rank 0's JSON line
included, goes to the parent's stdout -- and exits with the job's code).  One process per GPU, the set replicated on every rank, and per step ONE all-reduce of
kmax doubles (RCCL): strong scaling.  What a rank computes is the library's choice: auto evidence -- the
symmetric partition (a contiguous range of the sorted query blocks) or every N-th wave of the pruned walk's dispatch order
(`mce_knn_dotp_part_f64_dev`); cross evidence -- contiguous query rows of s1 against the replicated s2.  Under N > 1 the
line carries C3 (headline) AND, unless --no-extras, `configs.C2 / C4 / C5` timed the same way (barrier + synchronize
on both sides, max over ranks), `ranks_seen`, and per rank its device ordinal and its own HIP-event search time.

Prints ONE JSON line on rank 0.

`roofline` is for the dominant kernel, its launch duration measured with HIP events on the launch stream.  `achieved` =
the MFMA flops that kernel EXECUTED per launch (mce_last_search_stats: the symmetric sweep multiplies every pair of rows
once -- about half of the all-pairs count; 2*16*KST flop per multiplied pair) / that duration; `peak` = 2.5 PFLOP/s dense
fp16 (MI355X_MICROARCH.md); `frac` = achieved / peak.  It can be recomputed from profiles/: kernel_stats.csv (average
duration) and pmc_summary.csv (SQ_INSTS_MFMA x 32 768 flop per v_mfma_f32_32x32x16_f16).  The all-pairs figure -- what a
kernel WITHOUT the symmetry would have to execute for the same result -- is reported separately (`all_pairs_flops`,
`algorithmic_speedup`), never as a fraction of peak.  `search` covers every launch of the search (prepass, sweep, repair,
bucket merge).  `traffic`, `mfma_busy_frac`, `valu_per_mfma`, `wait_frac` come from the committed rocprofv3 passes of
this kernel (profiles/<round>/) and are only reported when that profile was taken from the SAME kernel sources as the
loaded library (profiles/<round>/meta.json: source_hash == mce_source_hash()); otherwise they are null and
`traffic_stale` is true.  --mode 1: knn_mfma_kernel, the pure fp64 MFMA sweep, against 78.6 TFLOP/s.

`cpu_baseline` = the reference's own CPU path -- scikit-learn NearestNeighbors exactly as MCEvidence.py:1093-1104 calls it
+ the NumPy volume/weight sum -- on a bounded query sample, rank 0, with the host it ran on.

`configs` (unless --no-extras): the other BASELINE.json GPU configs C2, C4, C5 through the same entry points (resident
data), each with ln E against the reference's own golden output and (N = 1) a CPU baseline on a query sample;
`fp64_mode` (N = 1): C3 through the fp64 sweep -- the reference-precision arithmetic end to end.
"""
import argparse
import glob
import hashlib
import json
import math
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 matrix = vector peak
F16_PEAK_TFLOPS = 2500.0     # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
N_SIMD = 1024                # 256 CUs x 4
PEAK_CLOCK_HZ = 2.4e9


def source_hash():
    """SHA-256 over the kernel sources (csrc/*.hpp, csrc/*.hip, sorted by name) -- the same digest csrc/Makefile bakes
    into the library (mce_source_hash()) and tools/profile_bench.sh stores next to a profile (meta.json)."""
    h = hashlib.sha256()
    src = os.path.join(REPO, "mcevidence_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hpp")) + glob.glob(os.path.join(src, "*.hip")), key=os.path.basename):
        h.update(open(f, "rb").read())
    return h.hexdigest()


def profile_counters(kernel_desc, library_hash=None, tag=""):
    """Counters of the dominant kernel from the newest committed rocprofv3 summary that has it
    (profiles/<round>/pmc_summary.csv: separate --pmc passes; FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 note; kernel_stats.csv: average duration).  {} when no profile of this kernel exists.
    `stale` = the profile's meta.json names other kernel sources than `library_hash` (or names none).  `tag`: "_C2" / "_C4" /
    "_C5" read that config's own passes (pmc_summary_C5.csv, kernel_stats_C5.csv: SQ counters only, no traffic)."""
    m = re.match(r"(\w+)<\w+=(\d+),\w+=(\d+)>", kernel_desc)
    if not m:
        return {}
    name, p1, p2 = m.groups()
    if name == "knn_f16_kernel" and "pruned" in kernel_desc:
        wants = ("%sILi%sELi%sELb1E" % (name, p1, p2),)
    elif name == "knn_f16_kernel" and "panel-kernel" in kernel_desc:
        wants = ("knn_panel_kernelILi%sELi%sE" % (p1, p2), "knn_panel_kernel<%s, %s," % (p1, p2), "knn_panel_kernel<%s, %s>" % (p1, p2))
    elif name == "knn_f16_kernel":
        sy = "2" if " symmetric" in kernel_desc else "0"
        wants = ("%sILi%sELi%sELb0ELb0ELi%sE" % (name, p1, p2, sy),) + (("%sILi%sELi%sELb0ELb0EE" % (name, p1, p2),) if sy == "0" else ())
    else:
        wants = ("%s<%s, %s>" % (name, p1, p2), "%sILi%sELi%sE" % (name, p1, p2))
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*", "pmc_summary%s.csv" % tag)), reverse=True):
        c = {}
        for ln in open(f):
            col = ln.strip().split(",")
            if len(col) >= 4 and any(w in ln for w in wants):
                try:
                    c[col[-3]] = float(col[-2]) / int(col[-1])          # per dispatch
                except ValueError:
                    pass
        if tag:
            if "SQ_INSTS_VALU" not in c:
                continue
            out = dict(source=os.path.relpath(f, REPO), traffic=None)
        elif "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        else:
            out = dict(source=os.path.relpath(f, REPO), traffic=(2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
                       traffic_note="per launch; (2*FETCH_SIZE + WRITE_SIZE) KB, separate --pmc passes")
        meta = os.path.join(os.path.dirname(f), "meta.json")
        out["profile_source_hash"] = json.load(open(meta)).get("source_hash") if os.path.exists(meta) else None
        out["stale"] = library_hash is not None and out["profile_source_hash"] != library_hash
        ks = os.path.join(os.path.dirname(f), "kernel_stats%s.csv" % tag)
        avg_ns = None
        if os.path.exists(ks):
            for ln in open(ks):
                if any(w in ln for w in wants):
                    try:
                        avg_ns = float(ln.split('",')[-1].split(",")[2]) if ln.startswith('"') else None
                    except (ValueError, IndexError):
                        avg_ns = None
                    break
        if avg_ns:
            out["profile_kernel_ms"] = round(avg_ns / 1e6, 3)
        if c.get("SQ_INSTS_MFMA"):
            out["mfma_insts"] = c["SQ_INSTS_MFMA"]
            out["valu_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"], 2)
            if avg_ns:      # SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per 8-pass MFMA, summed over the SIMDs
                out["mfma_busy_frac"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (N_SIMD * avg_ns * 1e-9 * PEAK_CLOCK_HZ), 3)
                out["profile_tflops_from_SQ_INSTS_MFMA"] = round(c["SQ_INSTS_MFMA"] * 32768.0 / (avg_ns * 1e-9) / 1e12, 1)
        if c.get("SQ_WAVE_CYCLES"):
            out["wait_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
        if c.get("SQ_INSTS_VALU"):
            out["valu_insts"] = c["SQ_INSTS_VALU"]
        out["counters"] = c
        return out
    return {}


def valu_issue_table():
    """Measured issue cost of a wave64 vector instruction by class (tools/valu_issue_clock.hip -> profiles/<round>/valu_issue_clock.json,
    round 6): SIMD cycles per instruction at the nominal 2.4 GHz with the SIMD saturated (best of 1 - 4 waves issuing independent
    instructions of the class).  MI355X_MICROARCH.md's "a wave64 fp32 op issues over 2 cycles" is not reached by any class: v_mov 2.6,
    v_fma_f32 2.8, integer 3.6, compares / min / max / selects / fp64 / conversions 4.2 - 4.5."""
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*", "valu_issue_clock.json")), reverse=True):
        try:
            t = json.load(open(f))
            return {k: v["simd_cycles_per_inst_saturated"] for k, v in t["classes"].items() if v.get("simd_cycles_per_inst_saturated")}, os.path.relpath(f, REPO)
        except (ValueError, KeyError, OSError):
            continue
    return {}, None


# which measured class prices which SQ_INSTS_VALU_* counter; what no class counter counts (compares, selects, min / max, moves, lane
# reads, logic: SQ_INSTS_VALU - MFMA - the classes) is priced as the compare + select pair -- the CHEAPEST of those, so the peak is not
# understated
VALU_CLASS_OF = {"ADD_F32": "v_fma_f32", "MUL_F32": "v_fma_f32", "FMA_F32": "v_fma_f32", "ADD_F64": "v_add_f64", "MUL_F64": "v_fma_f64", "FMA_F64": "v_fma_f64",
                 "INT32": "v_add_u32/v_lshlrev/v_and", "INT64": "v_add_u32/v_lshlrev/v_and", "CVT": "v_cvt_f32_f64/v_cvt_f64_f32",
                 "TRANS_F32": "v_cvt_f32_f64/v_cvt_f64_f32", "TRANS_F64": "v_cvt_f32_f64/v_cvt_f64_f32"}
VALU_OTHER_CLASS = "v_cmp_lt_f32+v_cndmask_b32"


def valu_issue_floor(counters):
    """Issue time floor of a kernel from its per-launch instruction mix: sum over classes of count x measured SIMD cycles per instruction,
    spread over the 1024 SIMDs at 2.4 GHz.  -> (seconds, by_class) or (None, None) when the mix or the table is missing."""
    table, src = valu_issue_table()
    total = counters.get("SQ_INSTS_VALU")
    if not table or not total or "SQ_INSTS_VALU_FMA_F32" not in counters:
        return None, None
    by, classified, cycles = {}, 0.0, 0.0
    for cname, cls in VALU_CLASS_OF.items():
        n = counters.get("SQ_INSTS_VALU_" + cname, 0.0)
        if n <= 0 or cls not in table:
            continue
        by[cname] = dict(insts=n, cycles_per_inst=table[cls], priced_as=cls)
        classified += n
        cycles += n * table[cls]
    other = max(total - counters.get("SQ_INSTS_MFMA", 0.0) - classified, 0.0)
    by["other (compare, select, min / max, move, lane read, logic)"] = dict(insts=other, cycles_per_inst=table.get(VALU_OTHER_CLASS), priced_as=VALU_OTHER_CLASS)
    cycles += other * table.get(VALU_OTHER_CLASS, 4.2)
    for v in by.values():
        v["share_of_issue_cycles"] = round(v["insts"] * v["cycles_per_inst"] / cycles, 4)
    return cycles / N_SIMD / PEAK_CLOCK_HZ, dict(classes=by, table=src, issue_cycles_total=cycles,
                                                 salu_insts=counters.get("SQ_INSTS_SALU"), lds_insts=counters.get("SQ_INSTS_LDS"), branch_insts=counters.get("SQ_INSTS_BRANCH"))


def config_roofline(kdesc, stats, nq, nr, d, lib_hash, tag="", prune_stats=None, profiled_shape=True):
    """`roofline` object of one timed config: what bounds its dominant kernel, what the kernel achieved against that bound over
    its HIP-event duration, and the same from the committed rocprofv3 passes of that config when they are of this build.
    Sweeps (fp16 filter, fp64): matrix-core bound -- executed MFMA flops (mce_last_search_stats) / duration against the dense
    peak.  Pruned walk: its binding resource is vector-instruction ISSUE (a tree walk: box tests, reach tests, list upkeep --
    ~180 k VALU against ~1 k MFMA per wave).  Round 6: the peak is no longer "one instruction per 4 cycles whatever it is" but the
    launch's own instruction MIX (SQ_INSTS_VALU_* class counters of the committed pass of this config, source-hash guarded) priced
    with the issue cost MEASURED per class (tools/valu_issue_clock.hip): frac = issue-time floor of the mix / kernel duration.
    `profiled_shape`: the run has the shape the committed counters were taken at (counters of the full-size pass
    divided by a scaled-down run's time mean nothing) -- otherwise the counter-derived fields are None."""
    kms = stats["kernel_ms"]
    if kms <= 0:
        return None
    is64 = kdesc.startswith("knn_mfma")
    peak = FP64_PEAK_TFLOPS if is64 else F16_PEAK_TFLOPS
    executed = stats["flops_main"]
    if "pruned" in kdesc and prune_stats:
        mk = re.search(r"KST=(\d+)", kdesc)
        kst = int(mk.group(1)) if mk else 1
        executed = prune_stats[1] * math.ceil(nq / 32.0) * math.ceil(nr / 32.0) * 32768.0 * kst      # tile fraction x tile pairs x flop per tile
    tf = executed / (kms * 1e-3) / 1e12 if executed > 0 else None
    prof = profile_counters(kdesc, lib_hash, tag) if tag else {}
    stale = bool(prof.get("stale", True)) or not profiled_shape
    live = (lambda k: None if stale else prof.get(k))
    if "pruned" in kdesc:
        vi = live("valu_insts")
        floor_s, mix = valu_issue_floor(prof.get("counters", {})) if not stale else (None, None)
        ach = vi / (kms * 1e-3) / 1e9 if vi else None
        peak_g = vi / floor_s / 1e9 if (vi and floor_s) else None
        roof = dict(bound="valu_issue", achieved=(round(ach, 1) if ach else None), peak=(round(peak_g, 1) if peak_g else None), unit="Ginst/s",
                    frac=(round(floor_s / (kms * 1e-3), 4) if floor_s else None), traffic=None,
                    what="k-d pruned walk: bound by vector-instruction issue (box / reach tests, list upkeep), not by the matrix cores; achieved = "
                         "SQ_INSTS_VALU per launch (committed rocprofv3 pass of this config) / kernel_ms; peak = the same instructions at the issue cost "
                         "measured per class (issue_cycles_by_class: SQ_INSTS_VALU_* mix x tools/valu_issue_clock.hip, 1024 SIMDs at 2.4 GHz)",
                    valu_insts_per_launch=vi, issue_cycles_by_class=mix, issue_floor_ms=(round(floor_s * 1e3, 3) if floor_s else None),
                    mfma_tflops=(round(tf, 1) if tf else None), mfma_frac=(round(tf / peak, 4) if tf else None))
    else:
        roof = dict(bound="mfma", achieved=(round(tf, 2) if tf else None), peak=peak, unit="TFLOP/s", frac=(round(tf / peak, 4) if tf else None),
                    traffic=None, executed_flops_per_launch=executed)
    roof.update(kernel_ms=round(kms, 3), kernel=kdesc, counters_stale=stale, counters_source=prof.get("source"),
                profile_kernel_ms=live("profile_kernel_ms"), valu_per_mfma=live("valu_per_mfma"), mfma_busy_frac=live("mfma_busy_frac"),
                wait_frac=live("wait_frac"), profile_tflops_from_SQ_INSTS_MFMA=live("profile_tflops_from_SQ_INSTS_MFMA"))
    return roof


def host_info():
    info = dict(os_cpu_count=os.cpu_count(), affinity=len(os.sched_getaffinity(0)))
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                info["cpu_model"] = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        import sklearn
        info["sklearn"] = sklearn.__version__
    except Exception:
        pass
    info["numpy"] = np.__version__
    try:
        from threadpoolctl import threadpool_info
        info["threadpools"] = [dict(api=t.get("user_api"), lib=t.get("internal_api"), threads=t.get("num_threads"), version=t.get("version")) for t in threadpool_info()]
    except Exception:
        pass
    return info


# ---------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher
# ---------------------------------------------------------------------------------------------------------------------
RANK_TIMEOUT_S = float(os.environ.get("MCE_BENCH_RANK_TIMEOUT", "900"))


def progress(msg):
    """One line on stderr per phase of a rank of a multi-rank run: what the self-launching parent's watchdog listens for."""
    if "WORLD_SIZE" in os.environ:
        sys.stderr.write("[bench rank %s/%s] %s\n" % (os.environ.get("RANK", "0"), os.environ["WORLD_SIZE"], msg))
        sys.stderr.flush()


def self_launch(argv, gpus, timeout_s=None, grace_s=15.0, reap_s=30.0):
    """Start the N ranks as children -- one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment,
    exactly what a launcher would hand them -- and wait.  Runs before torch is imported: this process never touches the GPU,
    and nothing is exec'ed from a process that has.  (Not through `python -m torch.distributed.run`: its argument parser
    rejects script options that abbreviate its own, e.g. `--n`.)  Returns the first non-zero exit code of a rank; the other
    ranks are then stopped by PID.

    Watchdog: the ranks' stdout and stderr come through pipes and are relayed line by line; every rank reports its phases on
    stderr (`progress`).  A rank that has been SILENT for `timeout_s` seconds (MCE_BENCH_RANK_TIMEOUT, default 900: a hung
    collective, a wedged device) is terminated -- as a child, by PID; nothing is re-exec'ed -- together with the others, the
    tail of its stderr is printed, and the parent exits with 124.  Ranks that do not leave on SIGTERM within `grace_s` are
    killed (SIGKILL, by PID); one that cannot be reaped `reap_s` after that is abandoned -- the parent never waits forever."""
    import collections
    import threading
    timeout_s = RANK_TIMEOUT_S if timeout_s is None else timeout_s
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or gpus) // gpus)))
    base.update(WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs, last, tails, threads = [], [], [], []

    def relay(pipe, sink, r, keep):
        for ln in iter(pipe.readline, ""):
            last[r] = time.monotonic()
            if keep is not None:
                keep.append(ln)
            sink.write(ln)
            sink.flush()
        pipe.close()

    for r in range(gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        last.append(time.monotonic())
        tails.append(collections.deque(maxlen=40))
        for pipe, sink, keep in ((p.stdout, sys.stdout, None), (p.stderr, sys.stderr, tails[r])):
            if pipe is not None:
                t = threading.Thread(target=relay, args=(pipe, sink, r, keep), daemon=True)
                t.start()
                threads.append(t)
    rc = 0
    live = list(range(gpus))
    stop_at = None                      # when the survivors were told to stop (SIGTERM); GRACE_S later they are killed (SIGKILL)
    GRACE_S, REAP_S = grace_s, reap_s

    def stop_all():
        nonlocal stop_at
        if stop_at is None:
            stop_at = time.monotonic()
            for q in live:
                if procs[q].poll() is None:
                    procs[q].terminate()

    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                if rc == 0 and time.monotonic() - last[r] > timeout_s:
                    rc = 124
                    sys.stderr.write("bench.py: rank %d silent for %.0f s -- stopping all ranks.  Its last stderr lines:\n%s\n"
                                     % (r, timeout_s, "".join(tails[r]) or "(none)"))
                    sys.stderr.flush()
                    stop_all()
                continue
            live.remove(r)
            if code != 0 and rc == 0:
                rc = code
                stop_all()               # a rank failed: the others would wait for it in the next collective
        if live and stop_at is not None:
            # a rank wedged in a driver call, or one that handles SIGTERM, does not leave by itself: SIGKILL after the grace
            # period -- children only, by PID -- and a bounded wait for the reaping; a child that cannot be reaped even then
            # (uninterruptible sleep in the driver) is abandoned, and the parent still exits non-zero
            waited = time.monotonic() - stop_at
            if waited > GRACE_S:
                for q in live:
                    if procs[q].poll() is None:
                        procs[q].kill()
            if waited > GRACE_S + REAP_S:
                sys.stderr.write("bench.py: rank(s) %s could not be reaped %.0f s after SIGKILL -- giving up on them\n" % (live, REAP_S))
                rc = rc or 124
                break
        if live:
            time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# the BASELINE.json configs as hot-path inputs (host side: chain -> whitened sets, weights, fs, the scalars of ln E)
# ---------------------------------------------------------------------------------------------------------------------
GOLDENS = {"C2": ("evidence_medium.json", "auto_n100000_d6_k4_C2"), "C3": ("evidence_big.json", "auto_n1000000_d27_k10_C3"),
           "C4": ("evidence_c4.json", "cross_n1000000_d15_k4_C4"), "C5": ("evidence_c5.json", "auto_n10000000_d6_k10_C5")}


def whiten_all(theta):
    """covariance eigen-system of ALL rows (the reference's covtype='all', MCEvidence.py:851-882, :842-849)"""
    cov = np.cov(theta.T)
    ev, U = np.linalg.eigh(cov)
    return np.ascontiguousarray((theta @ U) / np.sqrt(ev)), math.sqrt(float(np.prod(ev)))


def prep_config(name, scale=1.0):
    """X (queries = s1), Y (None: auto evidence), weight, fs, and the scalars MCEvidence.py:1120-1131 needs.
    `scale` < 1: the same recipe with fewer rows (functional checks of the N > 1 path on small boxes)."""
    from mcevidence_amd.synth import CONFIGS, config_chain
    chain, (r1, r2) = config_chain(name, n=max(2048, int(CONFIGS[name]["n"] * scale)))
    W, J = whiten_all(chain[:, 2:])
    if r1 is None:
        X, Y, c1 = W, None, chain
    else:
        X, Y, c1 = np.ascontiguousarray(W[r1]), np.ascontiguousarray(W[r2]), chain[r1]
    logL = -c1[:, 1]                                   # column 1 = -ln L (MCEvidence.py:399)
    logLmax = float(np.amax(logL))
    return dict(name=name, X=X, Y=Y, weight=np.ascontiguousarray(c1[:, 0]), fs=np.ascontiguousarray(logL - logLmax),
                kmax=CONFIGS[name]["kmax"], k0=1 if Y is None else 0, S=len(X), SumW=float(np.sum(c1[:, 0])), J=J, logLmax=logLmax,
                chain=chain if r1 is not None else None, split=(r1, r2), full_size=(scale == 1.0))


def lnE_from_dotp(dotp, cfg, lnPV=0.0):
    """MCEvidence.py:1120-1131 + the slice of :1157: k_nn = k (auto) / k + 1 (cross); returned MLE[1:]"""
    k0 = cfg["k0"]
    out = []
    for k in range(max(k0, 1), cfg["kmax"]):
        k_nn = k if k0 == 1 else k + 1
        out.append(math.log(cfg["SumW"] * dotp[k] / (cfg["S"] * k_nn + 1.0) * cfg["J"]) + cfg["logLmax"] - lnPV)
    return np.array(out)


def golden_lnE(name, cfg):
    f, case = GOLDENS.get(name, (None, None))
    p = os.path.join(REPO, "tests", "golden", f) if f else None
    if not p or not os.path.exists(p):
        return None
    for c in json.load(open(p)):
        if c["name"] == case and c["S"] == cfg["S"] and c["kmax"] == cfg["kmax"]:
            return c
    return None


class Ctx(object):
    """process-wide bits every timed config needs"""

    def __init__(self, torch, dist, _capi, world, rank, dev, dist_on=None):
        self.torch, self.dist, self.capi, self.world, self.rank, self.dev = torch, dist, _capi, world, rank, dev
        # the multi-rank code path (library partition + all-reduce); MCE_BENCH_FORCE_DIST=1 takes it with ONE rank too
        self.dist_on = (world > 1) if dist_on is None else dist_on

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist_on:
            self.dist.barrier()
        self.torch.cuda.synchronize()


def time_config(ctx, cfg, steps, warmup, mode=0, nsample=0, orc=None):
    """One config through the device-pointer entry points on resident data.  world = 1: mce_knn_dotp_f64_dev over the whole
    set.  world > 1: this rank's share + ONE all-reduce of kmax doubles per step -- auto evidence through
    mce_knn_dotp_part_f64_dev (the library's partition), cross evidence on the contiguous query rows [S r / W, S (r+1) / W)
    of s1 against the replicated s2.  EXACTLY `steps` timed steps between barrier + synchronize; ms = max over ranks."""
    torch, dist, _capi, world, rank, dev = ctx.torch, ctx.dist, ctx.capi, ctx.world, ctx.rank, ctx.dev
    dist_on = ctx.dist_on
    progress("config %s: %d steps + %d warm-up" % (cfg["name"], steps, warmup))
    X, Y, kmax, k0 = cfg["X"], cfg["Y"], cfg["kmax"], cfg["k0"]
    S, d = X.shape
    K = kmax - k0
    auto = Y is None
    lo, hi = (0, S) if (world == 1 or auto) else ((S * rank) // world, (S * (rank + 1)) // world)
    Xd = torch.from_numpy(X[lo:hi]).to(dev)
    Yd = Xd if auto else torch.from_numpy(Y).to(dev)
    nq, nr = hi - lo, (S if auto else Y.shape[0])
    w = torch.from_numpy(cfg["weight"][lo:hi]).to(dev)
    fs = torch.from_numpy(cfg["fs"][lo:hi]).to(dev)
    _capi.set_search_mode(mode)
    wsb = _capi.knn_workspace_bytes(nq, nr, d, K) + _capi.dotp_workspace_bytes(nq, kmax)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    out = torch.zeros(kmax, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    # a pruned search over several ranks (C5): the k-d preparation is DISTRIBUTED (round 6) -- this rank's part of the sorts, one all-reduce
    # of the permutation, the search on the shared order; agreed once, outside the timed region (MCE_BENCH_DIST_PREP=0: every rank repeats
    # the whole preparation, as before)
    dist_prep = False
    if dist_on and auto and world >= 2 and os.environ.get("MCE_BENCH_DIST_PREP", "1") != "0":
        from mcevidence_amd import parallel
        dist_prep = parallel.agree_all(_capi.prune_part_applies(nr, d, kmax, world))

    def step(dist_out=0):
        if dist_on and auto and dist_prep:
            off, cnt, lo_, hi_ = _capi.prune_part_prepare_dev(Xd.data_ptr(), nr, d, kmax, rank, world, ws.data_ptr(), wsb, st, want_range=True)
            if cnt <= 0:
                raise RuntimeError("bench.py: the distributed k-d preparation was agreed on but does not apply on rank %d" % rank)
            parallel.gather_permutation(ws, off, cnt, seg=(lo_, hi_))
            _capi.knn_dotp_part_prepared_dev(Xd.data_ptr(), nr, d, kmax, rank, world, w.data_ptr(), fs.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, st)
        elif dist_on and auto:
            _capi.knn_dotp_part_dev(Xd.data_ptr(), nr, d, kmax, rank, world, w.data_ptr(), fs.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, st)
        else:
            _capi.knn_dotp_dev(Xd.data_ptr(), nq, Yd.data_ptr(), nr, d, kmax, k0, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(),
                               dist_out, ws.data_ptr(), wsb, st)
        if dist_on:
            dist.all_reduce(out, op=dist.ReduceOp.SUM)          # the single collective of the path

    for _ in range(warmup):
        step()
    ctx.barrier()
    _capi.set_profiling(True)          # hipEvent brackets around the dominant kernel and around the whole search, on the launch stream
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    stats = _capi.last_search_stats()  # means over the timed steps (read after the timed region)
    _capi.set_profiling(False)
    kdesc = _capi.last_kernel()
    per_rank = None
    if dist_on:
        mine = torch.tensor([elapsed, stats["search_ms"], stats["kernel_ms"], float(torch.cuda.current_device()), float(nq)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = [t.cpu().numpy() for t in allr]
        elapsed = max(float(t[0]) for t in allr)
        per_rank = [dict(rank=i, device=int(t[3]), search_ms=round(float(t[1]), 3), kernel_ms=round(float(t[2]), 3), query_rows=int(t[4]),
                         ms_per_step=round(float(t[0]) / steps * 1e3, 3)) for i, t in enumerate(allr)]
    ms = elapsed / steps * 1e3
    dotp = out.cpu().numpy().copy()
    res = dict(nq=S, nr=nr, d=d, kmax=kmax, k0=k0, steps=steps, warmup=warmup, ms_per_step=round(ms, 3), kernel_ms=round(stats["kernel_ms"], 3),
               search_ms=round(stats["search_ms"], 3), queries_per_s=round(S / (ms * 1e-3), 1), kernel=kdesc, ranks=world)
    if per_rank:
        res["per_rank"] = per_rank
        res["partition"] = (("library partition of the auto-evidence search, k-d preparation distributed over the ranks (mce_prune_part_prepare_dev, all-reduce of "
                             "the permutation, mce_knn_dotp_part_prepared_f64_dev), one all-reduce of the sums" if dist_prep else
                             "library partition of the auto-evidence search (mce_knn_dotp_part_f64_dev), one all-reduce") if auto
                            else "contiguous query rows of s1 against the replicated s2, one all-reduce")
        res["distributed_kd_preparation"] = bool(dist_prep)
    if stats["flops_main"] > 0 and stats["kernel_ms"] > 0:
        res["executed_tflops"] = round(stats["flops_main"] / (stats["kernel_ms"] * 1e-3) / 1e12, 1)
    lnE = lnE_from_dotp(dotp, cfg)
    res["lnE"] = [float(x) for x in lnE]
    g = golden_lnE(cfg["name"], cfg)
    if g is not None:
        res["max_abs_dlnE_vs_reference"] = float(np.max(np.abs(lnE - np.array(g["lnE"]))))
        res["reference_wall_s"] = round(g.get("ref_wall_s", 0.0), 1)
    if nsample and not dist_on and orc is not None:
        dd = torch.zeros((nq, K), dtype=torch.float64, device=dev)
        step(dd.data_ptr())
        torch.cuda.synchronize()
        rng = np.random.default_rng(1)
        rows = np.sort(rng.choice(nq, min(nsample, nq), replace=False))
        od, _ = orc.knn_brute(X[rows], X if auto else Y, K + k0)
        od = od[:, k0:] if k0 == 1 else od
        got = dd[torch.from_numpy(rows).to(dev)].cpu().numpy()
        res["sampled_rows"] = len(rows)
        res["max_rel_dist_err_sampled_rows_vs_exact_cpu_search"] = float(np.max(np.abs(got - od) / od))
        del dd
    if not dist_on and K <= 32 and d <= 128:
        # what the default run-time certificate costs a host-pointer call of this shape (round 6: it is ON by default behind the fp16
        # filter): the step once more WITH the distances written out (what the check reads) + the re-check of 256 rows by an exact fp64
        # scan (mce_verify_knn_f64_dev), against the plain step -- resident data, after the timed region, best of 3
        try:
            dd = torch.zeros((nq, K), dtype=torch.float64, device=dev)
            vws = _capi.load().mce_verify_workspace_bytes(256, K)
            vbuf = torch.empty(max(int(vws), 1), dtype=torch.uint8, device=dev)
            vres = torch.zeros(2, dtype=torch.int32, device=dev)
            best = []
            for with_check in (False, True):
                ts = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    step(dd.data_ptr() if with_check else 0)
                    if with_check:
                        _capi.check(_capi.load().mce_verify_knn_f64_dev(Xd.data_ptr(), nq, Yd.data_ptr(), nr, d, K, 2 if auto else 0, 0, dd.data_ptr(), K, 256,
                                                                         12345, vres.data_ptr(), vbuf.data_ptr(), int(vws), st))
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1) * 1e3)
                best.append(min(ts))
            failed_rows = int(vres.cpu()[1])
            res["certificate"] = dict(rows=256, overhead_ms=round(best[1] - best[0], 3), step_without_ms=round(best[0], 3), rows_failed=failed_rows,
                                      note="default of the host-pointer entry points behind the fp16 filter (mce_options.verify = -1); MCE_VERIFY=0 turns it off")
            del dd, vbuf, vres
        except Exception as exc:          # (never lose the line over a diagnostic)
            res["certificate"] = dict(error="%s: %s" % (type(exc).__name__, exc))
    pstats = None
    if "pruned" in kdesc:            # (device counters in the workspace: read before it is freed)
        try:
            pstats = _capi.last_prune_stats()
            res["pruned_walk"] = dict(chunk_fraction=round(pstats[0], 5), tile_fraction=round(pstats[1], 5))
        except Exception:
            pass
    if cfg["name"] != "C3":          # (the headline's own, fuller object is built in main())
        res["roofline"] = config_roofline(kdesc, stats, nq, nr, d, _capi.source_hash(), "_" + cfg["name"], pstats, profiled_shape=cfg.get("full_size", True))
    _capi.set_search_mode(0)
    del Xd, Yd, ws, w, fs
    torch.cuda.empty_cache()
    return res, stats, dotp


def sklearn_baseline(cfg, algs, orc, seed=0, nrows_all=20000):
    """The reference's CPU call (MCEvidence.py:1093-1104) on a bounded sample of query rows against the full set."""
    from sklearn.neighbors import NearestNeighbors
    X, Y = cfg["X"], cfg["X"] if cfg["Y"] is None else cfg["Y"]
    rng = np.random.default_rng(seed)
    rows = np.sort(rng.choice(len(X), min(nrows_all, len(X)), replace=False))
    out = {}
    for alg, nrows in algs:
        sub = rows[:: max(1, len(rows) // nrows)]
        t0 = time.perf_counter()
        nb = NearestNeighbors(n_neighbors=cfg["kmax"] + 1, metric="euclidean", leaf_size=20, algorithm=alg, n_jobs=-1).fit(Y)
        t_fit = time.perf_counter() - t0
        dsk, _ = nb.kneighbors(X[sub])
        orc.dotp_literal(dsk, cfg["weight"][sub], cfg["fs"][sub], X.shape[1], cfg["k0"], cfg["kmax"])
        t_all = time.perf_counter() - t0
        out[alg] = dict(fit_method=str(nb._fit_method), query_rows=len(sub), fit_s=round(t_fit, 2), total_s=round(t_all, 2),
                        queries_per_s=round(len(sub) / (t_all - t_fit), 1), queries_per_s_incl_fit=round(len(sub) / t_all, 1))
    return dict(sample="random query rows against the full reference set; kind = the reference's own sklearn call, imported",
                cores=len(os.sched_getaffinity(0)), algorithms=out)


def beyond_baseline_shapes(ctx):
    """The kernels that serve rows longer than the BASELINE configs' (round 6): 100 k x 100 k auto evidence at d = 64 / 100 / 127 (the deep
    fp16 filter; kmax 25: its two passes), 128 / 256 (the long-row fp64 sweep) -- resident data, fused search + reduction, best of 3,
    with the library's own event bracket around the dominant kernel.  Not a BASELINE metric: a section of the line."""
    torch, _capi, dev = ctx.torch, ctx.capi, ctx.dev
    out = []
    for n, d, kmax in ((100000, 64, 10), (100000, 100, 10), (100000, 127, 10), (100000, 100, 25), (100000, 128, 10), (100000, 256, 10)):
        try:
            g = torch.Generator(device="cpu").manual_seed(n + d)
            X = torch.randn((n, d), dtype=torch.float64, generator=g).to(dev)
            w = torch.ones(n, dtype=torch.float64, device=dev)
            fs = torch.zeros(n, dtype=torch.float64, device=dev)
            wsb = _capi.knn_workspace_bytes(n, n, d, kmax - 1) + _capi.dotp_workspace_bytes(n, kmax)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            res = torch.zeros(kmax, dtype=torch.float64, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            best, kms = 1e30, None
            for it in range(4):
                _capi.set_profiling(it > 0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _capi.knn_dotp_dev(X.data_ptr(), n, X.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), res.data_ptr(), 0, ws.data_ptr(), wsb, st)
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) * 1e3
                if it > 0 and t < best:
                    best, stats = t, _capi.last_search_stats()
                _capi.set_profiling(False)
            kdesc = _capi.last_kernel()
            fp64 = "knn_long_kernel" in kdesc or "knn_mfma_kernel" in kdesc
            peak = FP64_PEAK_TFLOPS if fp64 else F16_PEAK_TFLOPS
            tf = stats["flops_main"] / (stats["kernel_ms"] * 1e-3) / 1e12 if stats["kernel_ms"] > 0 else None
            out.append(dict(n=n, d=d, kmax=kmax, ms=round(best, 3), kernel_ms=round(stats["kernel_ms"], 3), queries_per_s=round(n / (best * 1e-3), 1), kernel=kdesc,
                            roofline=dict(bound="mfma", achieved=None if tf is None else round(tf, 2), peak=peak, unit="TFLOP/s", frac=None if tf is None else round(tf / peak, 4),
                                          executed_flops_per_launch=stats["flops_main"])))
            del X, ws, w, fs
            torch.cuda.empty_cache()
        except Exception as exc:
            out.append(dict(n=n, d=d, kmax=kmax, error="%s: %s" % (type(exc).__name__, exc)))
    return out


def extra_configs(ctx, orc, pkg, scale=1.0):
    """C2, C4, C5 of BASELINE.json (C1 is the CPU plumbing config): resident data, the same timing as the headline."""
    torch, _capi, world = ctx.torch, ctx.capi, ctx.world
    out = {}
    plan = (("C2", 20, 3, 1000, (("auto", 20000),)),
            ("C4", 4, 1, 1000, (("auto", 1000), ("brute", 20000))),      # the reference's call picks kd_tree at d = 15: ~140 queries/s
            ("C5", 3, 1, 300, (("auto", 20000),)))
    for name, steps, warmup, nsample, algs in plan:
        if scale != 1.0:
            nsample, algs = min(nsample, 100), tuple((alg, min(nr, 500)) for alg, nr in algs)
        progress("preparing config %s" % name)
        cfg = prep_config(name, scale)
        try:
            res, _, _ = time_config(ctx, cfg, steps, warmup, nsample=0 if ctx.dist_on else nsample, orc=orc)
        except Exception as exc:      # (a section of the line, never the line: the headline above is already measured)
            progress("config %s failed: %s: %s" % (name, type(exc).__name__, exc))
            out[name] = dict(error="%s: %s" % (type(exc).__name__, exc))
            del cfg
            continue
        if ctx.rank == 0 and not ctx.dist_on:
            if name == "C4" and "max_abs_dlnE_vs_reference" in res:
                # ... and ln E through the class (device feeders, from host arrays) for this pair
                mce = pkg.MCEvidence([cfg["chain"]], kmax=cfg["kmax"], verbose=0).set_split(*cfg["split"])
                t0 = time.perf_counter()
                lnE = mce.evidence()
                res["evidence_call_from_host_s"] = round(time.perf_counter() - t0, 4)
                res["max_abs_dlnE_class_vs_resident_path"] = float(np.max(np.abs(lnE - np.array(res["lnE"]))))
            res["cpu_baseline"] = sklearn_baseline(cfg, algs, orc)
        out[name] = res
        del cfg
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=27)
    ap.add_argument("--kmax", type=int, default=10)
    ap.add_argument("--cpu-sample", type=int, default=100000, help="queries timed on the CPU baseline (0 = skip): ~12 s of the reference's sklearn call on a 256-thread host")
    ap.add_argument("--mode", type=int, default=0, help="0 auto (fp16 filter + fp64 refine), 1 fp64 MFMA sweep")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip the C2/C4/C5 and fp64-mode sections")
    ap.add_argument("--extras-scale", type=float, default=1.0, help="rows of the C2/C4/C5 sections x this factor (functional checks; 1 = BASELINE.json sizes)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one (before torch is imported -- this process never initialises the GPU)
        sys.exit(self_launch(sys.argv[1:], a.gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or without a launcher)" % (a.gpus, world, a.gpus))
    # MCE_BENCH_FORCE_DIST=1: `--gpus 1` takes the multi-rank code path -- a process group of ONE rank (RCCL by default), the
    # library's partition entry point, the all-reduce per step, the class's part feed -- so that a 1-GPU box exercises every
    # line an 8-GPU run executes (tests/test_gpu_nccl.py)
    dist_on = world > 1 or os.environ.get("MCE_BENCH_FORCE_DIST") == "1"
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["MCE_FORCE_DIST"] = "1"              # mcevidence_amd.parallel: a group of one rank counts as distributed
    if dist_on:
        # (MCE_BENCH_BACKEND=gloo MCE_BENCH_ONE_DEVICE=1: functional check of the N>1 path on a 1-GPU box)
        if os.environ.get("MCE_BENCH_ONE_DEVICE") == "1":
            local = 0
        elif local >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d wants GPU %d of %d" % (rank, local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        backend = os.environ.get("MCE_BENCH_BACKEND", "nccl")
        progress("init_process_group(%s) on device %d" % (backend, local))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        progress("process group up: %d rank(s)" % dist.get_world_size())
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from mcevidence_amd import _capi
    _capi.set_search_mode(a.mode)
    from mcevidence_amd.synth import gaussian_chain
    import mcevidence_amd as pkg
    ctx = Ctx(torch, dist, _capi, world, rank, dev, dist_on)

    # ---- synthetic chain (config C3 recipe) + host-side feeders (whitening etc.) ----
    n, d, kmax = a.n, a.d, a.kmax
    chain = gaussian_chain(seed=3, n=n, d=d, cov="corr")
    mce = pkg.MCEvidence([chain], kmax=kmax, verbose=0)
    cov = mce.get_covariance()
    s1, logL, weight, _ = mce.get_samples(n, prewhiten=False)
    Xh = np.ascontiguousarray(mce.diagonalise_chain(s1, cov["eVec"], cov["eVal"]))
    logLmax = float(np.amax(logL))
    fsh = np.ascontiguousarray(logL - logLmax)
    weight = np.ascontiguousarray(weight, dtype=np.float64)
    c3 = dict(name="C3", X=Xh, Y=None, weight=weight, fs=fsh, kmax=kmax, k0=1, S=n, SumW=float(np.sum(weight)), J=cov["J"], logLmax=logLmax)
    K = kmax - 1

    head, stats, dp = time_config(ctx, c3, a.steps, a.warmup, mode=a.mode)
    ms_step, kern_ms = head["ms_per_step"], stats["kernel_ms"]
    lnE = np.array(head["lnE"])

    e2e_ranks = None
    if dist_on:
        progress("evidence() from host arrays under the process group")
        # the whole MCEvidence(...).evidence() call from host arrays under the process group: each rank uploads the chain once,
        # whitens on its device, searches its share (mce_evidence_feed_part_f64), ONE all-reduce -- the PCIe-inclusive figure
        # (guarded like the sections below: the class's routes carry failure flags through their collectives, so the ranks fail
        # TOGETHER; a failure becomes {"error": ...} in the line, never a lost headline)
        try:
            mce.evidence()
            ctx.barrier()
            t2 = time.perf_counter()
            lnE_e2e = mce.evidence()
            mine = torch.tensor([time.perf_counter() - t2, float(np.max(np.abs(lnE_e2e - lnE)))], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            e2e_ranks = [dict(rank=i, seconds=round(float(t[0]), 4), max_abs_dlnE_vs_resident_path=float(t[1])) for i, t in enumerate(allr)]
        except Exception as exc:
            e2e_ranks = dict(error="%s: %s" % (type(exc).__name__, exc))
            progress("evidence() under the process group failed: %s" % exc)

    # Two ranks or more: the same workload through the ALL-PAIRS-ONCE partition as well, by default (round 6; MCE_BENCH_PAIRS_ONCE=0
    # skips it) -- every rank's sweep, the exchange of the candidates and the all-reduces inside the timed region, data resident --
    # so that the first run on a multi-GPU node times both partitions.  `value` stays the default partition's.  The section is
    # guarded: whatever fails in it (on any rank: the partition's collectives carry failure flags, so the ranks fail TOGETHER
    # and nobody is left in a collective) becomes {"error": ...} in the line, never a lost headline.
    pairs_once, po_stats, lnE1 = None, None, None
    if dist_on and world >= 2 and os.environ.get("MCE_BENCH_PAIRS_ONCE", "1") != "0" and a.mode == 0:
        from mcevidence_amd import parallel
        impl = None
        try:
            try:
                impl = parallel._HipPairsOnce(Xh, weight, fsh, kmax, world)
                applicable = impl.blocks() >= world
            except Exception as exc:         # (this rank cannot even set up: the others must not start the collectives without it)
                applicable, impl = False, None
                progress("pairs-once partition: setup failed on this rank: %s" % exc)
            if parallel.agree_all(applicable):
                progress("pairs-once partition: %d steps + %d warm-up" % (a.steps, a.warmup))
                st = {}
                for _ in range(max(a.warmup, 1)):
                    dp1 = parallel.pairs_once_knn_dotp(Xh, weight, fsh, kmax, stats=st, impl=impl)
                ctx.barrier()
                _capi.set_profiling(True)      # (the same event brackets as the default partition's steps: this rank's sweep kernel)
                t3 = time.perf_counter()
                for _ in range(a.steps):
                    dp1 = parallel.pairs_once_knn_dotp(Xh, weight, fsh, kmax, stats=st, impl=impl)
                mine_s = time.perf_counter() - t3
                ctx.barrier()
                t_all = time.perf_counter() - t3
                po_stats = _capi.last_search_stats()
                _capi.set_profiling(False)
                mine = torch.tensor([t_all, float(st["sent"]), float(st["received"]), mine_s, float(torch.cuda.current_device()),
                                     po_stats["kernel_ms"], po_stats["search_ms"]], dtype=torch.float64, device=dev)
                allr = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allr, mine)
                allr = [t.cpu().numpy() for t in allr]
                ms1 = max(float(t[0]) for t in allr) / a.steps * 1e3
                lnE1 = lnE_from_dotp(dp1, c3)
                pairs_once = dict(ms_per_step=round(ms1, 3), queries_per_s=round(n / (ms1 * 1e-3), 1), kernel=_capi.last_kernel(),
                                  per_rank=[dict(rank=i, device=int(t[4]), ms_per_step=round(float(t[3]) / a.steps * 1e3, 3), candidates_sent=int(t[1]),
                                                 candidates_received=int(t[2]), kernel_ms=round(float(t[5]), 3), search_ms=round(float(t[6]), 3))
                                            for i, t in enumerate(allr)],
                                  candidates_sent=[int(t[1]) for t in allr], candidates_received=[int(t[2]) for t in allr],
                                  max_abs_dlnE_vs_default_partition=float(np.max(np.abs(lnE1 - lnE))),
                                  vs_default_partition=round(ms_step / ms1, 3),
                                  collectives="all_reduce(MIN) of the rows' bounds, all_gather of the counts, all_reduce(MAX) of the flags, all_to_all of the candidates, all_reduce(SUM)")
            else:
                pairs_once = dict(skipped="the partition does not apply to this shape on %d ranks (or a rank could not set it up)" % world)
        except Exception as exc:
            pairs_once = dict(error="%s: %s" % (type(exc).__name__, exc))
            progress("pairs-once partition failed: %s" % exc)
        del impl
        torch.cuda.empty_cache()

    orc = None
    if rank == 0:
        from oracle import oracle_np as orc                 # checker / CPU baseline only; never inside a timed region
    extras = None
    want_extras = not a.no_extras and a.mode == 0 and ((n, d, kmax) == (1_000_000, 27, 10) or a.extras_scale != 1.0)
    if want_extras and dist_on:
        extras = extra_configs(ctx, orc, pkg, a.extras_scale)   # every rank takes part

    # Two ranks or more: the line's `value` is the FASTER of the library's two partitions of this workload, both timed above over
    # the same K steps between the same barriers (round 6: the all-pairs-once partition is predicted ahead at 2, 4 and 8 ranks --
    # profiles/r06_final/pairs_once_emulated.json -- but had never met a second GPU; the run decides, and says so:
    # `config.partition` names the one `value` is of, `partitions` carries both).  It must have reproduced the default partition's
    # ln E to 1e-9 to be eligible.  MCE_BENCH_HEADLINE=default keeps the default partition whatever the times.
    partition = "default"
    partitions = None
    head_per_rank = head.get("per_rank")
    dlnE_ref = head.get("max_abs_dlnE_vs_reference")
    if pairs_once is not None and "ms_per_step" in pairs_once:
        partitions = dict(default=dict(ms_per_step=round(ms_step, 3), queries_per_s=round(n / (ms_step * 1e-3), 1), kernel=head["kernel"],
                                       what="mce_knn_dotp_part_f64_dev: every rank takes its share of the symmetric sweep's units, one all-reduce"),
                          pairs_once=dict(ms_per_step=pairs_once["ms_per_step"], queries_per_s=pairs_once["queries_per_s"], kernel=pairs_once["kernel"],
                                          what="every pair of rows multiplied once per NODE: " + pairs_once["collectives"]))
        if (pairs_once["ms_per_step"] < ms_step and pairs_once["max_abs_dlnE_vs_default_partition"] < 1e-9 and po_stats is not None
                and po_stats["kernel_ms"] > 0 and os.environ.get("MCE_BENCH_HEADLINE", "") != "default"):
            partition = "pairs_once"
            ms_step, kern_ms, stats, lnE = pairs_once["ms_per_step"], po_stats["kernel_ms"], po_stats, lnE1
            head = dict(head, kernel=pairs_once["kernel"])
            head_per_rank = pairs_once["per_rank"]
            g = golden_lnE("C3", c3)
            dlnE_ref = float(np.max(np.abs(lnE1 - np.array(g["lnE"])))) if g is not None else None
        partitions["value_is_of"] = partition

    out = None
    if rank == 0:
        kdesc = head["kernel"]
        is_f16 = kdesc.startswith("knn_f16")
        mk = re.search(r"KST=(\d+)", kdesc)
        kst = int(mk.group(1)) if mk else (d + 16) // 16     # capi.hip: f16_ksteps(D) = (D + 16) / 16
        KS = (d + 1 + 3) // 4
        if is_f16:
            peak, flop_pair = F16_PEAK_TFLOPS, 2.0 * 16 * kst
            what = "fp16 MFMA filter (2*16*KST flop per multiplied pair) + exact fp64 refine of the survivors"
        else:
            peak, flop_pair = FP64_PEAK_TFLOPS, 2.0 * 4 * KS
            what = "fp64 MFMA sweep (2*4*KS flop/pair), fp64 MFMA-bound (SURVEY 8d)"
        executed = stats["flops_main"]
        achieved = executed / (kern_ms * 1e-3) / 1e12
        all_pairs = float(n) * n * flop_pair / world                 # what a sweep without the symmetry would execute on this rank
        lib_hash = _capi.source_hash()
        prof = profile_counters(kdesc, lib_hash)
        # (the committed counters are one launch of the profiled shape on ONE GPU: a rank's share of it, or another size, is another launch)
        stale = bool(prof.get("stale", True)) or world > 1 or (n, d, kmax) != (1_000_000, 27, 10)
        live = (lambda k: None if stale else prof.get(k))           # counters of other sources are not this kernel's
        roof = dict(bound="mfma", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4),
                    traffic=live("traffic"), kernel_ms=round(kern_ms, 3), kernel=kdesc,
                    executed_flops_per_launch=executed, flop_per_multiplied_pair=flop_pair, what=what,
                    all_pairs_flops=all_pairs, algorithmic_speedup=round(all_pairs / executed, 3) if executed > 0 else None,
                    all_pairs_equivalent_tflops=round(all_pairs / (kern_ms * 1e-3) / 1e12, 1),
                    search=dict(ms=round(stats["search_ms"], 3), executed_flops=stats["flops_all"],
                                tflops=round(stats["flops_all"] / (stats["search_ms"] * 1e-3) / 1e12, 2),
                                frac=round(stats["flops_all"] / (stats["search_ms"] * 1e-3) / 1e12 / peak, 4),
                                note="every launch of the search: packing, prepass, sweep, repair, bucket merge"),
                    traffic_stale=stale, traffic_source=prof.get("source"), traffic_note=prof.get("traffic_note"),
                    library_source_hash=lib_hash, profile_source_hash=prof.get("profile_source_hash"),
                    source_hash_matches_tree=(lib_hash == source_hash()),
                    hbm_gbps=(round(prof["traffic"] / (kern_ms * 1e-3) / 1e9, 1) if live("traffic") else None),
                    hbm_frac_of_8TBps=(round(prof["traffic"] / (kern_ms * 1e-3) / 8e12, 4) if live("traffic") else None),
                    mfma_busy_frac=live("mfma_busy_frac"), valu_per_mfma=live("valu_per_mfma"), wait_frac=live("wait_frac"),
                    profile_kernel_ms=live("profile_kernel_ms"), profile_tflops_from_SQ_INSTS_MFMA=live("profile_tflops_from_SQ_INSTS_MFMA"),
                    fp64_equivalent_tflops=round(float(n) * n / world * 2.0 * 4 * KS / (kern_ms * 1e-3) / 1e12, 2))
        cpu = None
        fp64_mode = None
        e2e = None
        shapes = None
        if not dist_on:
            if a.cpu_sample > 0:
                rng = np.random.default_rng(0)
                rows = np.sort(rng.choice(n, size=min(a.cpu_sample, n), replace=False))
                from sklearn.neighbors import NearestNeighbors
                t1 = time.perf_counter()
                nb = NearestNeighbors(n_neighbors=kmax + 1, metric="euclidean", leaf_size=20, algorithm="auto", n_jobs=-1).fit(Xh)   # MCEvidence.py:1100-1101
                dsk, _ = nb.kneighbors(Xh[rows])                                                                                  # :1104
                orc.dotp_literal(dsk, weight[rows], fsh[rows], d, 1, kmax)                                                        # :1107-1117
                t_cpu = time.perf_counter() - t1
                cpu = dict(value=round(len(rows) / t_cpu, 1), unit="queries/s", cores=len(os.sched_getaffinity(0)), kind="reference",
                           what="the reference's own CPU path: sklearn.neighbors.NearestNeighbors(n_neighbors=kmax+1, metric='euclidean', leaf_size=20, algorithm='auto', n_jobs=-1) "
                                "exactly as MCEvidence.py:1093-1104 calls it + its NumPy volume/weight sum (third-party library code, imported; nothing of the reference is compiled or copied)",
                           sample="%d random query rows of the same chain against the full %d-row reference set (%.1f s)" % (len(rows), n, t_cpu),
                           fit_method=str(nb._fit_method), host=host_info())
                # row-level parity on the sample: distances of the sampled rows, GPU vs CPU
                dg, _ = _capi.knn(Xh[rows], Xh, kmax + 1, self_mode=_capi.SELF_NONE)
                cpu["max_rel_dist_err_vs_gpu"] = float(np.max(np.abs(dg[:, 1:kmax] - dsk[:, 1:kmax]) / dsk[:, 1:kmax]))
                full = os.path.join(REPO, "profiles", "cpu_full_c3.json")          # the full 1M-query CPU pass, run once on a GPU box (tools/cpu_full_c3.py)
                if os.path.exists(full) and (n, d, kmax) == (1_000_000, 27, 10):
                    fc = json.load(open(full))
                    cpu["full_run_cached"] = dict(seconds=fc["seconds"], queries_per_s=fc["queries_per_s"], cores=fc["host"]["affinity"], cpu_model=fc["host"].get("cpu_model"),
                                                  max_abs_dlnE_gpu_vs_this_cpu_run=float(np.max(np.abs(lnE - np.array(fc["lnE"])))), source="profiles/cpu_full_c3.json")
            # whole MCEvidence(...).evidence() call from host arrays (device feeders + H2D + hot path): the PCIe-inclusive
            # figure, reported next to `value` (which is HBM-resident)
            mce.evidence()
            t2 = time.perf_counter()
            lnE_e2e = mce.evidence()
            e2e = dict(seconds=round(time.perf_counter() - t2, 4), queries_per_s=round(n / (time.perf_counter() - t2), 1),
                       max_abs_dlnE_vs_resident_path=float(np.max(np.abs(lnE_e2e - lnE))))
            if want_extras:
                fp64_mode, _, _ = time_config(ctx, c3, 3, 1, mode=1)
                fl = float(n) * n * 2.0 * 4 * KS / (fp64_mode["kernel_ms"] * 1e-3) / 1e12
                fp64_mode["roofline"] = dict(bound="mfma", achieved=round(fl, 2), peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=round(fl / FP64_PEAK_TFLOPS, 4),
                                             note="pure fp64 arithmetic: v_mfma_f64_16x16x4_f64 over all pairs (2*4*KS flop/pair, unpadded rows) + fp64 refine")
                fp64_mode["max_abs_dlnE_vs_default_mode"] = float(np.max(np.abs(np.array(fp64_mode["lnE"]) - lnE)))
                extras = extra_configs(ctx, orc, pkg, a.extras_scale)
                if a.extras_scale == 1.0:
                    shapes = beyond_baseline_shapes(ctx)
        if isinstance(e2e_ranks, dict):
            e2e = e2e_ranks
        elif e2e_ranks:
            slow = max(r["seconds"] for r in e2e_ranks)
            e2e = dict(seconds=slow, queries_per_s=round(n / slow, 1), per_rank=e2e_ranks,
                       max_abs_dlnE_vs_resident_path=max(r["max_abs_dlnE_vs_resident_path"] for r in e2e_ranks),
                       note="every rank: one upload of the chain, device covariance + whitening, its share of the search, one all-reduce")
        dlnE = dlnE_ref if (n, d, kmax) == (1_000_000, 27, 10) else None
        out = dict(metric="knn_queries_per_sec", value=round(n / (ms_step * 1e-3), 1), unit="queries/s", n_gpus=world,
                   steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True,
                   scaling="strong", vs_baseline=None, dtype="f64" if a.mode == 1 else "f16 filter + f64 refine (exact f64 results)", data="synthetic",
                   config=dict(workload="C3: auto-evidence, seeded Gaussian chain N=%d D=%d kmax=%d (K=%d true neighbours/query), %s" %
                               (n, d, kmax, K, "one GPU" if world == 1 else
                                ("the all-pairs-once partition over %d GPUs (DESIGN.md 5): bounds all-reduce, candidates all_to_all, one all-reduce of kmax doubles" % world
                                 if partition == "pairs_once" else "the library's partition over %d GPUs (DESIGN.md 5), one all-reduce of kmax doubles" % world)),
                               N=n, D=d, kmax=kmax, ranks=world, partition=(None if world == 1 else partition)),
                   ranks_seen=(dist.get_world_size() if dist_on else 1), backend=(dist.get_backend() if dist_on else None),
                   per_rank=head_per_rank, partitions=partitions,
                   max_abs_dlnE_vs_reference=dlnE, lnE=[round(float(x), 10) for x in lnE],
                   roofline=roof, cpu_baseline=cpu, evidence_call_from_host=e2e, certificate=head.get("certificate"), configs=extras, fp64_mode=fp64_mode,
                   beyond_baseline_shapes=shapes)
        if pairs_once is not None:
            out["pairs_once"] = pairs_once
        print(json.dumps(out), flush=True)
    if dist_on:
        progress("done")
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
