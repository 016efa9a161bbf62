"""Multi-GPU query sharding: one process per GPU, ``torch.distributed`` over RCCL.

The kNN evidence sum shards naturally: queries are
independent given a replicated reference set.  Rank r takes the contiguous
query rows [N*r/W, N*(r+1)/W) (auto mode: its self-exclusion offset is its first
global row), computes its partial ``dotp[k]`` with the fused HIP entry point,
and ONE all-reduce(sum) of ``kmax`` doubles (<= 256 B, latency-bound; xGMI
bandwidth is irrelevant) completes the job.  The reference has no equivalent:
its only parallelism is joblib threads inside scikit-learn
(``/root/reference/MCEvidence.py:952,1094,1101``) and an mpi4py task farm over
datasets (``planck_mcevidence.py:149-160``).

torch is plumbing here (process group, device tensors); the compute is the C ABI.
"""
from __future__ import annotations

import numpy as np


def is_distributed():
    try:
        import torch.distributed as dist
    except Exception:
        return False
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_bounds(n, world, rank):
    """Contiguous, balanced row range of ``rank`` (same split rule as the C library's
    multi-device path, ``mce_knn_dotp_f64``)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def _local_hip(Xs, Y, ws, fss, kmax, k0, self_offset, want_dist):
    """This rank's shard on this rank's GPU (torch.cuda.current_device())."""
    import torch
    from . import _capi
    dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
    out = _capi.knn_dotp(Xs, Y, ws, fss, kmax, k0, self_offset=self_offset, return_dist=want_dist, devices=[dev])
    return out if want_dist else (out, None)


def sharded_knn_dotp(X, Y, weight, fs, kmax, k0, want_dist=False, group=None, local_fn=None):
    """Query-sharded fused kNN + reduction.  Every rank passes the FULL arrays (they are
    replicated host-side, as the reference set must be anyway) and gets the full
    ``dotp`` back.  ``local_fn`` lets the CPU tests substitute the per-shard compute."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = X.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    ref = X if Y is None else Y
    fn = local_fn or _local_hip
    if hi > lo:
        part, dpart = fn(np.ascontiguousarray(X[lo:hi]), ref, weight[lo:hi], fs[lo:hi], kmax, k0, lo if k0 == 1 else 0, want_dist)
    else:
        part, dpart = np.zeros(kmax), (np.zeros((0, kmax - k0)) if want_dist else None)
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.as_tensor(np.asarray(part, dtype=np.float64), device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)      # the single RCCL collective of the path
    dotp = t.cpu().numpy()
    dist_full = None
    if want_dist:
        # row-concatenation of the shards (debug/verbose path only)
        gathered = [None] * world
        dist.all_gather_object(gathered, dpart, group=group)
        dist_full = np.concatenate(gathered, axis=0)
    return dotp, dist_full
