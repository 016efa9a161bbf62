"""RCCL first contact on a 1-GPU box: an `nccl` process group of ONE rank drives every multi-rank code path of the package --
the library's partition entry points, `parallel.feed_part_reduce`, `parallel.sharded_knn_dotp`, the class's `evidence()`
under a group, the farm of `evidence_many`, and `bench.py`'s timed loop (`MCE_BENCH_FORCE_DIST=1`) -- and every result must
equal the non-distributed run's bit for bit (a sum over one rank is the rank's own value).  What this cannot show is the
exchange between devices; what it does show is that the `nccl` backend initialises with this package's environment, that the
all-reduce runs on the device tensors the code hands it, and that no line of the N > 1 path is first executed on the driver's
8-GPU clock.  Each case runs in its own process (a process group is process-wide state)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import LNE_TOL, REPO

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import json, os, socket, sys
import numpy as np
sys.path.insert(0, %(repo)r)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["MASTER_ADDR"] = "127.0.0.1"
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
import logging
logging.disable(logging.CRITICAL)
import mcevidence_amd as pkg
from mcevidence_amd import _capi, parallel
from mcevidence_amd.synth import gaussian_chain

out = dict(backend=dist.get_backend(), world=dist.get_world_size())
assert not parallel.is_distributed()                 # a group of one rank is not "distributed" ...
parallel.force_distributed(True)
assert parallel.is_distributed()                     # ... unless the bring-up hook says so

# 1. the ONE collective of the device-feeder route, on a CUDA tensor over RCCL
part = np.array([0.0, 1.25, 2.5, 1e-300, 3.0e200])
red = parallel.feed_part_reduce(part, 0xDEADBEEFCAFEF00D)
out["feed_part_reduce_identical"] = bool(np.array_equal(red, part))

# 2. sharded_knn_dotp: auto evidence through the library's partition (symmetric sweep size and a small set), cross evidence
#    through the row shard -- against the plain single-process call
rng = np.random.default_rng(7)
cases = {}
for name, n, d, kmax, cross in (("auto_small", 30000, 6, 4, False), ("auto_sym", 300000, 27, 10, False), ("auto_walk", 400000, 4, 5, False), ("cross", 60000, 15, 4, True)):
    X = rng.standard_normal((n, d))
    Y = rng.standard_normal((n + 1000, d)) if cross else None
    w = rng.integers(1, 4, n).astype(float)
    fs = -rng.random(n)
    k0 = 0 if cross else 1
    got, _ = parallel.sharded_knn_dotp(X, Y, w, fs, kmax, k0)
    kern = _capi.last_kernel()
    ref = _capi.knn_dotp(X, Y, w, fs, kmax, k0)
    cases[name] = dict(identical=bool(np.array_equal(got, ref)), kernel=kern, max_rel=float(np.max(np.abs(got[k0:] - ref[k0:]) / ref[k0:])))
out["sharded_knn_dotp"] = cases

# 3. the class under the group: part feed (mce_evidence_feed_part_f64, part 0 of 1) + all-reduce, vs no group involvement
chain = gaussian_chain(seed=11, n=200000, d=8, weights="int", cov="corr")
m = pkg.MCEvidence([chain], kmax=5, verbose=0)
lnE_dist = m.evidence()
two = gaussian_chain(seed=12, n=120000, d=15, cov="corr")
mx = pkg.MCEvidence([two], kmax=4, verbose=0).set_split(np.arange(0, 60000), np.arange(60000, 120000))
lnX_dist = mx.evidence()
many_dist = pkg.evidence_many([pkg.MCEvidence([gaussian_chain(seed=20 + i, n=9000 + 1000 * i, d=6)], kmax=3, verbose=0) for i in range(5)])
parallel.force_distributed(False)
assert not parallel.is_distributed()
lnE_one = pkg.MCEvidence([chain], kmax=5, verbose=0).evidence()
lnX_one = pkg.MCEvidence([two], kmax=4, verbose=0).set_split(np.arange(0, 60000), np.arange(60000, 120000)).evidence()
many_one = pkg.evidence_many([pkg.MCEvidence([gaussian_chain(seed=20 + i, n=9000 + 1000 * i, d=6)], kmax=3, verbose=0) for i in range(5)])
out["class_auto_identical"] = bool(np.array_equal(lnE_dist, lnE_one))
out["class_cross_identical"] = bool(np.array_equal(lnX_dist, lnX_one))
out["evidence_many_identical"] = bool(all(np.array_equal(a, b) for a, b in zip(many_dist, many_one)))
out["lnE"] = [float(x) for x in lnE_dist]

# 4. a rank whose share fails still joins the collective and raises its own error
parallel.force_distributed(True)
try:
    parallel.feed_part_reduce(np.zeros(3), None, failed=MemoryError("boom"))
    out["failed_rank_raises"] = False
except MemoryError:
    out["failed_rank_raises"] = True
# 5. the collectives of the all-pairs-once partition (parallel.pairs_once_knn_dotp) as it issues them, over RCCL on device
#    tensors: MIN over a float64 view into a uint8 workspace, all_gather of int64 counts, MAX over int32 flags, and the
#    all_to_all_single of 16-byte candidates with split sizes (one rank: everything comes back)
dev = torch.device("cuda", 0)
wsbuf = torch.zeros(4096 + 8 * 1000, dtype=torch.uint8, device=dev)
bounds = wsbuf[4096:4096 + 8 * 1000].view(torch.float64)
bounds.copy_(torch.arange(1000, dtype=torch.float64, device=dev))
dist.all_reduce(bounds, op=dist.ReduceOp.MIN)
counts = torch.tensor([777], dtype=torch.int64, device=dev)
table = [torch.zeros(1, dtype=torch.int64, device=dev)]
dist.all_gather(table, counts)
flags = torch.tensor([0, 1, 0, 1], dtype=torch.int32, device=dev)
dist.all_reduce(flags, op=dist.ReduceOp.MAX)
send = torch.arange(2 * 777, dtype=torch.float64, device=dev).reshape(777, 2)
recv = parallel._exchange_rows(send, [777], [int(table[0][0])], None)
empty = parallel._exchange_rows(send[:0], [0], [0], None)
out["pairs_once_collectives"] = bool(torch.equal(bounds, torch.arange(1000, dtype=torch.float64, device=dev)) and int(table[0][0]) == 777
                                     and flags.tolist() == [0, 1, 0, 1] and torch.equal(recv, send) and recv.is_cuda and empty.shape[0] == 0)
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out), flush=True)
"""


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MCE_FORCE_DIST")}
    env.update(extra)
    return env


def test_nccl_group_of_one_rank_drives_every_multi_rank_path():
    r = subprocess.run([sys.executable, "-c", _SCRIPT % dict(repo=REPO)], capture_output=True, text=True, env=_clean_env(), timeout=900, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["feed_part_reduce_identical"] and out["failed_rank_raises"]
    for name, c in out["sharded_knn_dotp"].items():
        assert c["identical"], (name, c)
    assert "symmetric" in out["sharded_knn_dotp"]["auto_sym"]["kernel"] and "pruned" in out["sharded_knn_dotp"]["auto_walk"]["kernel"]
    assert out["class_auto_identical"] and out["class_cross_identical"] and out["evidence_many_identical"]
    assert out["pairs_once_collectives"]


def _bench_line(args, env_extra, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=_clean_env(**env_extra), timeout=timeout, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def test_bench_timed_loop_over_nccl_with_one_rank():
    """`MCE_BENCH_FORCE_DIST=1 python bench.py --gpus 1`: the `world > 1` code path of bench.py -- nccl group, the library's
    partition entry point, one all-reduce per timed step, the per-rank gather, the class's evidence() under the group, the
    C2 / C4 / C5 sections timed the same way -- with ONE rank; every ln E equals the plain one-GPU run's to the last bit."""
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--n", "150000", "--d", "27", "--kmax", "10", "--cpu-sample", "0", "--extras-scale", "0.05"]
    forced, err = _bench_line(common, dict(MCE_BENCH_FORCE_DIST="1"))
    plain, _ = _bench_line(common, {})
    assert forced["backend"] == "nccl" and forced["ranks_seen"] == 1 and forced["n_gpus"] == 1 and plain["backend"] is None
    assert forced["per_rank"] and forced["per_rank"][0]["device"] == 0
    assert "[bench rank 0/1] process group up: 1 rank(s)" in err           # the phases the self-launch watchdog listens for
    assert forced["lnE"] == plain["lnE"]
    assert forced["evidence_call_from_host"]["max_abs_dlnE_vs_resident_path"] < LNE_TOL
    for name in ("C2", "C4", "C5"):
        a, b = forced["configs"][name], plain["configs"][name]
        assert a["lnE"] == b["lnE"], name
        assert a["roofline"]["kernel_ms"] > 0 and a["roofline"]["bound"] in ("mfma", "valu_issue")
    assert forced["configs"]["C5"]["roofline"]["bound"] == "valu_issue" and forced["configs"]["C5"]["roofline"]["mfma_frac"] is not None
