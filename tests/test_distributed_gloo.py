"""N>1 path on CPU: world_size-2 gloo process group, query sharding + one all-reduce.
The per-shard compute is the CPU oracle (tests only); what is under test is the
sharding rule, the self-exclusion offsets and the reduce."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import REPO, orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_oracle(Xs, Y, ws, fss, kmax, k0, self_offset, want_dist):
    K = kmax - k0
    d, _ = orc.knn_brute(Xs, Y, K, self_mode=2 if k0 == 1 else 0, self_offset=self_offset)
    full = np.zeros((Xs.shape[0], kmax))
    full[:, k0:] = d
    return orc.dotp_literal(full, ws, fss, Xs.shape[1], k0, kmax), (d if want_dist else None)


def _worker(rank, world, port, k0, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcevidence_amd import parallel
    rng = np.random.default_rng(42)
    n, d, kmax = 1501, 5, 4
    X = rng.standard_normal((n, d))
    Y = None if k0 == 1 else rng.standard_normal((1300, d))
    w = rng.integers(1, 4, n).astype(float)
    fs = -rng.random(n)
    assert parallel.is_distributed()
    dotp, dd = parallel.sharded_knn_dotp(X, Y, w, fs, kmax, k0, want_dist=True, local_fn=_local_oracle)
    # every rank can also drive the whole class through the same path
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    import logging
    logging.disable(logging.CRITICAL)

    class Be(object):
        def knn_dotp(self, X, Y, weight, fs, kmax, k0, want_dist=False):
            return parallel.sharded_knn_dotp(X, Y, weight, fs, kmax, k0, want_dist=want_dist, local_fn=_local_oracle)
    lnE = pkg.MCEvidence([gaussian_chain(3, 1200, 4)], kmax=4, verbose=0, backend=Be()).evidence()
    if rank == 0:
        q.put((dotp, dd, lnE))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k0", [1, 0])
def test_two_rank_sharding_matches_single_process(k0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, k0, q)) for r in range(2)]
    for p in procs:
        p.start()
    dotp, dd, lnE = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(42)
    n, d, kmax = 1501, 5, 4
    X = rng.standard_normal((n, d))
    Y = X if k0 == 1 else rng.standard_normal((1300, d))
    w = rng.integers(1, 4, n).astype(float)
    fs = -rng.random(n)
    ref, refd = _local_oracle(X, Y, w, fs, kmax, k0, 0, True)
    assert np.allclose(dotp, ref, rtol=1e-13)
    assert np.array_equal(dd, refd)
    # single-process class result with the same oracle backend
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    from helpers import OracleBackend
    one = pkg.MCEvidence([gaussian_chain(3, 1200, 4)], kmax=4, verbose=0, backend=OracleBackend()).evidence()
    assert np.allclose(lnE, one, atol=1e-12)


def test_shard_bounds_cover_everything():
    from mcevidence_amd.parallel import shard_bounds
    for n in (1, 7, 1000, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def _farm_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import logging
    logging.disable(logging.CRITICAL)
    import mcevidence_amd as pkg
    from mcevidence_amd import parallel
    from mcevidence_amd.synth import gaussian_chain
    from helpers import OracleFeedBackend
    seen = []

    class Farm(OracleFeedBackend):
        def evidence_feed_batch(self, problems):
            def local(ps):
                seen.append(len(ps))
                return [OracleFeedBackend.evidence_feed(self, *p) for p in ps]
            return parallel.farm_evidence_feed(problems, local_fn=local)
    ms = [pkg.MCEvidence([gaussian_chain(seed=30 + i, n=500 + 130 * i, d=3)], kmax=3 + (i % 2), verbose=0, backend=Farm()) for i in range(7)]
    be = ms[0].backend
    for m in ms:
        m.backend = be
    out = pkg.evidence_many(ms)
    failed = None
    try:                                            # a failing problem raises on EVERY rank
        def boom(ps):
            raise ValueError("math domain error")
        parallel.farm_evidence_feed([m._feed_problem("all", False)[0] for m in ms[:1]], local_fn=boom)
    except (ValueError, RuntimeError) as exc:
        failed = type(exc).__name__
    q.put((rank, out, seen, failed))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_problem_farm_matches_single_process():
    """evidence_many under a 2-rank group: problems are split over the ranks (each runs its share as
    one batch), one all-reduce returns every result to every rank, identical to a single process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_farm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    from mcevidence_amd.parallel import farm_assignment
    from helpers import OracleFeedBackend
    be = OracleFeedBackend()
    ref = [pkg.MCEvidence([gaussian_chain(seed=30 + i, n=500 + 130 * i, d=3)], kmax=3 + (i % 2), verbose=0, backend=be).evidence() for i in range(7)]
    for rank, out, seen, failed in got:
        assert len(out) == 7
        for a, b in zip(out, ref):
            assert np.array_equal(a, b)
        assert failed in ("ValueError", "RuntimeError")
    assert got[0][2][0] + got[1][2][0] == 7 and min(got[0][2][0], got[1][2][0]) >= 2      # both ranks worked
    owner = farm_assignment([float(n) ** 2 for n in (500 + 130 * i for i in range(7))], 2)
    assert sorted(set(owner)) == [0, 1]
    loads = [sum((500 + 130 * i) ** 2 for i in range(7) if owner[i] == r) for r in (0, 1)]
    assert max(loads) < 1.35 * min(loads)


def _split_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import logging
    import tempfile
    logging.disable(logging.CRITICAL)
    import mcevidence_amd as pkg
    from mcevidence_amd import parallel
    from mcevidence_amd.synth import gaussian_chain

    class Be(object):
        def knn_dotp(self, X, Y, weight, fs, kmax, k0, want_dist=False):
            return parallel.sharded_knn_dotp(X, Y, weight, fs, kmax, k0, want_dist=want_dist, local_fn=_local_oracle)
    # every rank has its OWN global RNG state, as separate torchrun processes have
    np.random.seed(1000 + 17 * rank)
    chain = gaussian_chain(5, 1400, 4, weights="int")
    m = pkg.MCEvidence([chain], kmax=4, split=True, s1frac=0.4, verbose=0, backend=Be())
    rows1 = np.asarray(m.gd.data["s1"].ichain)
    lnE = m.evidence(covtype="all")
    # Poisson thinning of a file-read chain draws from the global RNG too
    tmp = tempfile.mkdtemp()
    np.savetxt(os.path.join(tmp, "c_1.txt"), chain)
    mt = pkg.MCEvidence(os.path.join(tmp, "c"), kmax=3, thinlen=0.5, verbose=0, backend=Be())
    wthin = np.asarray(mt.gd.data["s1"].weights)
    lnEt = mt.evidence()
    # ranks that do hold different rows are refused, on every rank
    refused = None
    X = np.random.default_rng(rank).standard_normal((300, 3))
    try:
        parallel.sharded_knn_dotp(X, None, np.ones(300), np.zeros(300), 3, 1, local_fn=_local_oracle)
    except RuntimeError as exc:
        refused = "different samples" in str(exc)
    q.put((rank, rows1, lnE, wthin, lnEt, refused))
    dist.barrier()
    dist.destroy_process_group()


def test_random_split_and_thinning_agree_across_ranks():
    """split=True / 0<thinlen<1 under a process group: the draws come from rank 0 (each process has its own
    global RNG), so every rank reduces shards of the SAME partition; ranks with different inputs raise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, rows_a, lnE_a, w_a, lnEt_a, ref_a), (_, rows_b, lnE_b, w_b, lnEt_b, ref_b) = got
    assert np.array_equal(rows_a, rows_b) and np.array_equal(w_a, w_b)
    assert np.array_equal(lnE_a, lnE_b) and np.array_equal(lnEt_a, lnEt_b)
    assert ref_a is True and ref_b is True
    # and it is rank 0's split: the single-process class under rank 0's seed gives the same numbers
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    from helpers import OracleBackend
    np.random.seed(1000)
    one = pkg.MCEvidence([gaussian_chain(5, 1400, 4, weights="int")], kmax=4, split=True, s1frac=0.4, verbose=0, backend=OracleBackend())
    assert np.array_equal(np.asarray(one.gd.data["s1"].ichain), rows_a)
    assert np.allclose(one.evidence(covtype="all"), lnE_a, atol=1e-12)


# --------------------------------------------------------------------------- the symmetric partition (auto evidence)
def _part_oracle(Y, w, fs, kmax, rank, world):
    """CPU double of ``_capi.knn_dotp_part``: the library's partition restated on the host
    (``parallel.symmetric_partition_rows``), each owned row searched exactly among ALL rows."""
    from mcevidence_amd import parallel
    rows = parallel.symmetric_partition_rows(Y, world, rank)
    out = np.zeros(kmax)
    if len(rows):
        d, _ = orc.knn_brute(Y[rows], Y, kmax, self_mode=0)          # column 0: the row itself
        full = np.zeros((len(rows), kmax))
        full[:, 1:] = d[:, 1:kmax]
        out = orc.dotp_literal(full, w[rows], fs[rows], Y.shape[1], 1, kmax)
    return out


def _part_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcevidence_amd import parallel
    rng = np.random.default_rng(7)
    n, d, kmax = 2600, 6, 5                                             # 6 blocks of 512: ranks of a 4-rank job own 1 or 2
    Y = rng.standard_normal((n, d)) * (1.0 + rng.random((1, d)))
    w = rng.integers(1, 4, n).astype(float)
    fs = -rng.random(n)
    dotp, _ = parallel.sharded_knn_dotp(Y, None, w, fs, kmax, 1, part_fn=_part_oracle)
    owned = parallel.symmetric_partition_rows(Y, world, rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, owned.tolist())
    if rank == 0:
        q.put((dotp, gathered))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_symmetric_partition_over_ranks_equals_the_single_rank_sum(world):
    """world sizes 2 and 4 over gloo: every rank sums the evidence terms of ITS range of the sorted blocks, ONE
    all-reduce; the result equals the single-process sum to 1e-12 and the ranks' rows tile the set exactly once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_part_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    dotp, owned = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(7)
    n, d, kmax = 2600, 6, 5
    Y = rng.standard_normal((n, d)) * (1.0 + rng.random((1, d)))
    w = rng.integers(1, 4, n).astype(float)
    fs = -rng.random(n)
    one = _part_oracle(Y, w, fs, kmax, 0, 1)
    assert np.allclose(dotp[1:], one[1:], rtol=1e-12, atol=0)
    allrows = np.concatenate([np.asarray(o, dtype=np.int64) for o in owned])
    assert len(allrows) == n and np.array_equal(np.sort(allrows), np.arange(n))
    sizes = [len(o) for o in owned]
    assert max(sizes) - min(sizes) <= 512 + 511                          # whole blocks; the last one is partial


class _HostPairsOnce:
    """Host restatement of one rank's share of the all-pairs-once partition (include/mcevidence_hip.h: mce_pairs_once_*;
    csrc/sym_types.hpp: PanelGeom.blk_stride) for the CPU test of parallel.pairs_once_knn_dotp's collectives: rows sorted by
    their distance from the mean, blocks of B rows, rank r owns the blocks r, r + W, r + 2W, ...; it multiplies each of its
    blocks a against the blocks 0..a (itself included).  Every multiplied pair (i, j) is a candidate for i (kept here) and
    for j (kept here if j is this rank's, shipped to j's owner otherwise) -- entries {d2, caller row of the other, sorted row}."""
    ENTRY = np.dtype([("d2", "f8"), ("src", "i4"), ("row", "i4")])
    B = 64

    def __init__(self, Y, w, fs, kmax):
        self.Y, self.w, self.fs, self.kmax = Y, w, fs, kmax
        n = Y.shape[0]
        key = ((Y - Y.mean(axis=0)) ** 2).sum(axis=1).astype(np.float32)
        self.order = np.argsort(key, kind="stable")                       # sorted position -> caller's row
        self.nblk = (n + self.B - 1) // self.B

    def blocks(self):
        return self.nblk

    def _own(self, b):
        return b % self.world == self.rank

    def _pairs_of_blocks(self):
        """(query block a, tile block b) this rank multiplies; a == b: the block against itself"""
        return [(a, b) for a in range(self.rank, self.nblk, self.world) for b in range(a + 1)]

    def prepare(self, rank, world):
        """a bound on every own row's K-th squared distance (here simply the K-th distance within the row's own block), +inf elsewhere"""
        self.rank, self.world = rank, world
        n, B, K = self.Y.shape[0], self.B, self.kmax - 1
        Ys = self.Y[self.order]
        self.bounds = torch.full((self.nblk * B,), float("inf"), dtype=torch.float64)
        for b in range(rank, self.nblk, world):
            rows = np.arange(b * B, min((b + 1) * B, n))
            d2 = ((Ys[rows][:, None, :] - Ys[rows][None, :, :]) ** 2).sum(-1)
            if len(rows) > K:
                self.bounds[b * B:b * B + len(rows)] = torch.from_numpy(np.sort(d2, axis=1)[:, K])
        return self.bounds

    def sweep(self, rank, world, nblk):
        assert bool(torch.isfinite(self.bounds[:(self.Y.shape[0] // self.B) * self.B]).all())          # after the MIN: everybody's rows (whole blocks)
        n, B = self.Y.shape[0], self.B
        Ys = self.Y[self.order]
        self.mine = {}                     # sorted row -> list of (d2, caller row of the neighbour)
        ship = [[] for _ in range(world)]
        for a, b in self._pairs_of_blocks():
            ia = np.arange(a * B, min((a + 1) * B, n)); jb = np.arange(b * B, min((b + 1) * B, n))
            d2 = ((Ys[ia][:, None, :] - Ys[jb][None, :, :]) ** 2).sum(-1)
            for x, i in enumerate(ia):
                for y, j in enumerate(jb):
                    if i == j or (a == b and j > i):
                        continue                    # a block against itself: every pair once, from the larger row's side
                    self.mine.setdefault(int(i), []).append((d2[x, y], int(self.order[j])))
                    if self._own(b):
                        self.mine.setdefault(int(j), []).append((d2[x, y], int(self.order[i])))
                    else:
                        ship[b % world].append((d2[x, y], int(self.order[i]), int(j)))
        self.ship = ship
        counts = torch.tensor([len(v) for v in ship], dtype=torch.int64)
        return counts, torch.zeros(nblk, dtype=torch.int32)

    def export(self, total):
        ent = np.zeros(total, dtype=self.ENTRY)
        k = 0
        for dest in self.ship:
            for d2, src, row in dest:
                ent[k] = (d2, src, row); k += 1
        assert k == total
        return torch.from_numpy(ent.view(np.float64).reshape(-1, 2).copy())

    def finish(self, recv, flags):
        ent = recv.numpy().copy().reshape(-1).view(self.ENTRY)
        n, B, K = self.Y.shape[0], self.B, self.kmax - 1
        for e in ent:
            assert 0 <= e["row"] < n and self._own(int(e["row"]) // B)
            self.mine.setdefault(int(e["row"]), []).append((float(e["d2"]), int(e["src"])))
        rows = np.concatenate([np.arange(b * B, min((b + 1) * B, n)) for b in range(self.rank, self.nblk, self.world)] + [np.zeros(0, dtype=int)])
        full = np.zeros((len(rows), self.kmax))
        for x, i in enumerate(rows):
            c = sorted(self.mine.get(int(i), []))
            assert len(c) >= K and len({src for _, src in c}) == len(c)          # every pair once: no neighbour twice
            full[x, 1:] = np.sqrt([v for v, _ in c[:K]])
        cr = self.order[rows]
        return orc.dotp_literal(full, self.w[cr], self.fs[cr], self.Y.shape[1], 1, self.kmax) if len(rows) else np.zeros(self.kmax)


def _pairs_once_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcevidence_amd import parallel
    rng = np.random.default_rng(11)
    n, d, kmax = 700, 5, 4                                               # 11 blocks of 64
    Y = rng.standard_normal((n, d)) * (1.0 + rng.random((1, d)))
    w = rng.integers(1, 4, n).astype(float)
    fs = -rng.random(n)
    stats = {}
    dotp = parallel.pairs_once_knn_dotp(Y, w, fs, kmax, stats=stats, impl=_HostPairsOnce(Y, w, fs, kmax))
    gathered = [None] * world
    dist.all_gather_object(gathered, (stats["sent"], stats["received"]))
    if rank == 0:
        q.put((dotp, gathered))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_pairs_once_partition_collectives_over_gloo(world):
    """parallel.pairs_once_knn_dotp's control flow on CPU, the library's three calls replaced by a host restatement of the
    partition: all_gather of the counts, all_reduce(MAX) of the flags, all_to_all_single of the 16-byte candidates with the
    counts as split sizes, all_reduce(sum) of the sums.  Every row ends with each of its neighbours exactly once, and the
    result equals the single-process sum -- odd and even world sizes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pairs_once_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    dotp, traffic = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(11)
    n, d, kmax = 700, 5, 4
    Y = rng.standard_normal((n, d)) * (1.0 + rng.random((1, d)))
    w = rng.integers(1, 4, n).astype(float)
    fs = -rng.random(n)
    one = _part_oracle(Y, w, fs, kmax, 0, 1)
    assert np.allclose(dotp[1:], one[1:], rtol=1e-12, atol=0)
    assert sum(s for s, _ in traffic) == sum(r for _, r in traffic) > 0


class _FailingPairsOnce(_HostPairsOnce):
    """one of the three calls after the bounds raises on this rank (a HIP error, an out-of-memory send buffer, ...)"""
    def __init__(self, *a, stage=None):
        super().__init__(*a)
        self.stage = stage

    def sweep(self, rank, world, nblk):
        if self.stage == "sweep":
            raise MemoryError("pairs-once sweep: out of device memory (test)")
        counts, flags = super().sweep(rank, world, nblk)
        if self.stage == "self":
            counts[rank] = 3            # candidates addressed to itself: the rank-local check must not strand the others either
        return counts, flags

    def export(self, total):
        if self.stage == "export":
            raise MemoryError("pairs-once export: send buffer (test)")
        return super().export(total)

    def finish(self, recv, flags):
        if self.stage == "finish":
            raise MemoryError("pairs-once finish (test)")
        return super().finish(recv, flags)


def _pairs_once_failing_worker(rank, world, port, q, stage):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from mcevidence_amd import parallel
    rng = np.random.default_rng(11)
    n, d, kmax = 400, 4, 3
    Y = rng.standard_normal((n, d))
    w = np.ones(n)
    fs = -rng.random(n)
    impl = _FailingPairsOnce(Y, w, fs, kmax, stage=stage if rank == 1 else None)
    try:
        parallel.pairs_once_knn_dotp(Y, w, fs, kmax, impl=impl)
        out = ("ok", "")
    except (RuntimeError, MemoryError) as e:
        out = ("raised", type(e).__name__ + ": " + str(e))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stage", ["sweep", "self", "export", "finish"])
def test_pairs_once_rank_failing_after_the_bounds_strands_nobody(stage):
    """only the FIRST collective of pairs_once_knn_dotp was protected.  A rank whose sweep, export or finish raises
    (or whose counts fail the rank-local check) still joins the collectives that follow -- the failure flag rides behind the
    counts in the all_gather, or in the last all-reduce -- so every rank raises instead of sitting in a collective until its
    timeout: the failing rank its own exception, the others a RuntimeError that says a rank failed."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pairs_once_failing_worker, args=(r, 3, port, q, stage)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(got[r][0] == "raised" for r in range(3)), got
    own = "RuntimeError: pairs-once partition: a rank has candidates" if stage == "self" else "MemoryError"
    assert got[1][1].startswith(own), got
    for r in (0, 2):
        assert "1 of the 3 ranks" in got[r][1], got


def _route_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if rank == 1:
        os.environ["MCE_PAIRS_ONCE"] = "1"          # set on ONE rank only
    else:
        os.environ.pop("MCE_PAIRS_ONCE", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcevidence_amd import parallel, _capi
    _capi.pairs_once_blocks = lambda n, d, kmax: 100           # (no library call on the CPU box)
    a = parallel.pairs_once_route(100000, 27, 10)
    os.environ["MCE_PAIRS_ONCE"] = "1"
    b = parallel.pairs_once_route(100000, 27, 10)
    c = parallel.pairs_once_route(100000, 27, 10, local_ok=(rank != 0))
    q.put((rank, (a, b, c)))
    dist.barrier()
    dist.destroy_process_group()


def test_pairs_once_route_is_agreed_by_all_ranks():
    """The choice between the all-pairs-once partition and the part feed is made ONCE for the group (all-reduce MIN of every
    rank's own answer): the variable set on one rank only, or a rank whose device differs, sends everybody down the default
    route -- never into mismatched collectives."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_route_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == (False, True, False)


def _route2_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if rank == 1:
        os.environ["MCE_NODE_UPLOAD"] = "1"         # set on ONE rank only
    else:
        os.environ.pop("MCE_NODE_UPLOAD", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcevidence_amd import parallel, _capi
    _capi.prune_part_applies = lambda n, d, kmax, nparts: True        # (no library call on the CPU box)
    S = np.random.default_rng(0).standard_normal((64, 3))
    w = np.ones(64)
    a = parallel.pruned_part_feed(S, 3, 4, w, w, local_ok=(rank != 0))            # rank 0's device is not the current one
    b = parallel.gather_chain_on_device(S, None, 3, w, w, local_ok=(rank != 0))   # ... so nobody takes the node upload either
    c = parallel.gather_chain_on_device(S, None, 3, w, w, local_ok=parallel.node_upload_enabled(None))   # the variable on one rank
    q.put((rank, (a, b, c)))
    dist.barrier()
    dist.destroy_process_group()


def test_kd_route_and_node_upload_are_agreed_by_all_ranks():
    """The distributed k-d preparation and the one-upload-per-node gather are collectives too: a rank that cannot take them (its
    device is not the current one, the variable set elsewhere only, no GPU at all) enters them with its own 'no', and EVERY
    rank falls back to the part feed together -- none is left waiting in a gather the others never join."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_route2_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == (None, None, None)


def _perm_worker(rank, world, port, q, mode):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcevidence_amd import parallel
    cnt, off = 10 * 2048, 256
    bounds = [0, 3 * 2048, 7 * 2048, cnt] if world == 3 else [0, 5 * 2048, cnt]      # uneven ranges, in rank order
    full = torch.arange(cnt, dtype=torch.int32).flip(0) - 1                           # the "single-GPU permutation" (with a -1 in it)
    ws = torch.zeros(off + 4 * cnt + 64, dtype=torch.uint8)
    view = ws[off:off + 4 * cnt].view(torch.int32)
    lo, hi = bounds[rank], bounds[rank + 1]
    view[lo:hi] = full[lo:hi]
    try:
        if mode == "gather":
            parallel.gather_permutation(ws, off, cnt, seg=(lo, hi))
        elif mode == "reduce":
            parallel.gather_permutation(ws, off, cnt)
        else:                          # a rank that reports a range which does not tile the array: everybody raises
            parallel.gather_permutation(ws, off, cnt, seg=(lo, hi - (2048 if rank == 0 else 0)))
        out = ("ok", bool(torch.equal(view, full)) and int(ws[:off].sum()) == 0 and int(ws[off + 4 * cnt:].sum()) == 0)
    except RuntimeError as e:
        out = ("raised", str(e))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(3, "gather"), (2, "gather"), (3, "reduce"), (2, "gap")])
def test_the_ranks_ranges_of_the_kd_order_are_put_together(world, mode):
    """The one extra collective of the distributed k-d preparation (parallel.gather_permutation): every rank holds its own range of the
    permutation and zeros elsewhere; all_gather of the (uneven) ranges -- bounds first, then equal-length chunks -- or the all-reduce of the
    whole array leaves the SAME full array on every rank and touches nothing around it; ranges that do not tile the array raise everywhere."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_perm_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if mode == "gap":
        assert all(got[r][0] == "raised" and "do not tile" in got[r][1] for r in range(world)), got
    else:
        assert all(got[r] == ("ok", True) for r in range(world)), got


def test_replica_fingerprint_sees_every_row():
    from mcevidence_amd.parallel import replica_fingerprint
    rng = np.random.default_rng(3)
    A = rng.standard_normal((50000, 7))
    h = replica_fingerprint(A, None, A[:, 0])
    B = A.copy()
    B[12345, 3] = np.nextafter(B[12345, 3], 1.0)                         # one bit in one row that no sampling would hit
    assert replica_fingerprint(B, None, B[:, 0]) != h
    assert replica_fingerprint(A, None, A[:, 0]) == h and 0 <= h < 2 ** 63


# ---- the device-feeder route of the class under a process group (mce_evidence_feed_part_f64) -------------------------------
def _fake_feed_part(S1, S2, d, cov_mode, kmax, w, fs, part, nparts, device=0, want_checksum=True):
    """CPU stand-in for _capi.evidence_feed_part (tests only): whitening with the covariance of all rows, this rank's
    contiguous rows of s1 through the oracle, a byte fingerprint as the checksum."""
    import hashlib
    S1 = np.ascontiguousarray(np.asarray(S1)[:, :d])
    S2a = None if S2 is None else np.ascontiguousarray(np.asarray(S2)[:, :d])
    allrows = S1 if S2a is None else np.concatenate((S1, S2a))
    cov = np.atleast_2d(np.cov(allrows.T))
    ev, U = np.linalg.eigh(cov)
    X = (S1 @ U) / np.sqrt(ev)
    Y = X if S2a is None else (S2a @ U) / np.sqrt(ev)
    k0 = 1 if S2a is None else 0
    lo, hi = (len(X) * part) // nparts, (len(X) * (part + 1)) // nparts
    dotp, _ = _local_oracle(np.ascontiguousarray(X[lo:hi]), Y, w[lo:hi], fs[lo:hi], kmax, k0, lo if k0 == 1 else 0, False)
    h = hashlib.blake2b(digest_size=8)
    for a in (S1, S2a, w, fs):
        h.update(b"-" if a is None else np.ascontiguousarray(a).tobytes())
    return dotp, float(np.sqrt(np.prod(ev))), ev[::-1].copy(), (int.from_bytes(h.digest(), "little") if want_checksum else None)


def _failing_feed_part(*a, **k):
    raise MemoryError("mce_evidence_feed_part_f64: out of device memory (test)")


def _feed_part_worker(rank, world, port, q, split, poison):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import logging
    logging.disable(logging.CRITICAL)
    import mcevidence_amd as pkg
    from mcevidence_amd import _capi
    from mcevidence_amd.synth import gaussian_chain
    _capi.evidence_feed_part = _fake_feed_part
    chain = gaussian_chain(5, 1400, 4, weights="int", cov="corr")
    if poison == "fail" and rank == 1:
        _capi.evidence_feed_part = _failing_feed_part           # this rank's share raises inside the library call
    elif poison and rank == 1:
        chain = chain.copy()
        chain[777, 3] = np.nextafter(chain[777, 3], 9.0)         # one bit of one sample differs on one rank
    kw = dict(split=True, s1frac=0.4) if split else {}
    m = pkg.MCEvidence([chain], kmax=4, verbose=0, **kw)
    assert isinstance(m.backend, pkg.HipBackend)
    try:
        out = ("ok", m.evidence())
    except (RuntimeError, MemoryError) as e:
        out = ("raised", type(e).__name__ + ": " + str(e))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("split,poison", [(False, False), (True, False), (False, True), (False, "fail")])
def test_class_under_a_process_group_takes_the_part_feed_and_one_all_reduce(split, poison):
    """MCEvidence(...).evidence() with HipBackend under a 2-rank group: every rank hands the library the whole chain
    (evidence_feed_part: here a CPU stand-in), gets its share of the sums, ONE all-reduce carries sums and input
    fingerprints.  Same ln E as one process; ranks holding different samples raise on BOTH ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_feed_part_worker, args=(r, 2, port, q, split, poison)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if poison == "fail":
        # a rank whose share raises still joins the ONE collective (failure flag): nobody hangs, the failing rank re-raises its own
        # error, the other one learns that a rank failed
        assert got[0][0] == got[1][0] == "raised"
        assert "1 of the 2 ranks" in got[0][1] and got[1][1].startswith("MemoryError")
        return
    if poison:
        assert got[0][0] == got[1][0] == "raised" and "different samples" in got[0][1]
        return
    assert got[0][0] == got[1][0] == "ok" and np.array_equal(got[0][1], got[1][1])
    import mcevidence_amd as pkg
    from mcevidence_amd.synth import gaussian_chain
    from helpers import OracleBackend
    if split:
        return      # (the realised split is rank 0's draw from the global RNG: both ranks agree -- checked above; the value itself
                    #  is compared with a single process on the GPU box, tests/test_gpu_parity.py, with a caller-chosen split)
    one = pkg.MCEvidence([gaussian_chain(5, 1400, 4, weights="int", cov="corr")], kmax=4, verbose=0, backend=OracleBackend()).evidence()
    assert np.allclose(got[0][1], one, atol=1e-11)


def test_feed_part_reduce_checks_fingerprints_exactly():
    """the Cauchy-Schwarz test on four 16-bit pieces: exact in fp64 up to 1024 ranks"""
    from mcevidence_amd import parallel
    import unittest.mock as um
    for world, sums in ((1024, [0xFFFFFFFFFFFFFFFF] * 1024), (1024, [0xFFFFFFFFFFFFFFFF] * 1023 + [0xFFFFFFFFFFFFFFFE]), (3, [5, 5, 5]), (3, [5, 5, 1 << 63])):
        def fake_reduce(vec, group=None, _w=world, _s=sums):
            tot = np.zeros_like(vec)
            for c in _s:
                p = [float((c >> (16 * i)) & 0xFFFF) for i in range(4)]
                tot[-9:-5] += p
                tot[-5:-1] += [x * x for x in p]
            tot[:-9] = vec[:-9] * _w
            return tot
        with um.patch.object(parallel, "_reduce_partial", fake_reduce), um.patch("torch.distributed.get_world_size", lambda g=None: world):
            same = len(set(sums)) == 1
            if same:
                assert np.array_equal(parallel.feed_part_reduce(np.array([0.0, 2.0]), sums[0]), [0.0, 2.0 * world])
            else:
                with pytest.raises(RuntimeError):
                    parallel.feed_part_reduce(np.array([0.0, 2.0]), sums[0])
