"""bench.py's own logic: the self-launch of `--gpus N`, ln E from the reduced sums, the staleness guard of the profile
counters (CPU), and the whole N = 2 path from a plain shell on the GPU box (both ranks on its one GPU, gloo)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import LNE_TOL, load_golden

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = load_golden()


def test_self_launch_starts_one_rank_per_gpu(monkeypatch):
    import bench
    started = []

    class FakeProc(object):
        stdout = stderr = None

        def __init__(self, cmd, env=None, **kw):
            self.cmd, self.env, self.code, self.terminated = cmd, env, (3 if env["RANK"] == "2" else 0), False
            started.append(self)

        def poll(self):
            return self.code

        def terminate(self):
            self.terminated = True

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    assert bench.self_launch(["--gpus", "4", "--n", "5000"], 4) == 3          # a failing rank's exit code is the parent's
    assert [p.env["RANK"] for p in started] == ["0", "1", "2", "3"] and [p.env["LOCAL_RANK"] for p in started] == ["0", "1", "2", "3"]
    for p in started:
        assert p.cmd == [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--n", "5000"]
        assert p.env["WORLD_SIZE"] == "4" and p.env["MASTER_ADDR"] == "127.0.0.1" and 0 < int(p.env["MASTER_PORT"]) < 65536
        assert p.env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert len({p.env["MASTER_PORT"] for p in started}) == 1


def test_self_launch_watchdog_stops_a_silent_rank(monkeypatch, capsys):
    """a rank that says nothing for the time limit (a hung collective, a wedged device) is terminated AS A CHILD together with
    the others, the tail of its stderr is shown and the parent returns 124 -- nothing is re-exec'ed"""
    import io
    import bench
    started = []

    class Hung(object):
        def __init__(self, cmd, env=None, **kw):
            self.rank, self.terminated = int(env["RANK"]), False
            self.stdout = io.StringIO("")
            self.stderr = io.StringIO("[bench rank %d/2] init_process_group(nccl) on device %d\n" % (self.rank, self.rank))
            started.append(self)

        def poll(self):
            return -15 if self.terminated else None

        def terminate(self):
            self.terminated = True

    monkeypatch.setattr(bench.subprocess, "Popen", Hung)
    assert bench.self_launch(["--gpus", "2"], 2, timeout_s=0.3) == 124
    assert all(p.terminated for p in started)
    err = capsys.readouterr().err
    assert "silent for" in err and "init_process_group(nccl)" in err


def test_self_launch_watchdog_kills_what_sigterm_does_not_stop(monkeypatch, capsys):
    """a rank wedged in a driver call (or one that handles SIGTERM) ignores terminate(); after the grace period it
    is killed by PID, and a rank that cannot even be reaped does not keep the parent: 124 after a bounded wait."""
    import io
    import bench

    def make(reapable):
        started = []

        class Wedged(object):
            def __init__(self, cmd, env=None, **kw):
                self.rank, self.terminated, self.killed = int(env["RANK"]), False, False
                self.stdout, self.stderr = io.StringIO(""), io.StringIO("")
                started.append(self)

            def poll(self):
                if self.rank == 0:
                    return -15 if self.terminated else None          # rank 0 leaves on SIGTERM
                return -9 if (self.killed and reapable) else None       # rank 1 only on SIGKILL -- or never

            def terminate(self):
                self.terminated = True

            def kill(self):
                self.killed = True
        return Wedged, started

    for reapable in (True, False):
        cls, started = make(reapable)
        monkeypatch.setattr(bench.subprocess, "Popen", cls)
        assert bench.self_launch(["--gpus", "2"], 2, timeout_s=0.2, grace_s=0.3, reap_s=0.3) == 124
        assert started[1].terminated and started[1].killed and not started[0].killed
    assert "could not be reaped" in capsys.readouterr().err


def test_parent_of_a_self_launch_never_imports_torch():
    """`python bench.py --gpus 2` without a launcher: the parent only spawns (nothing that could initialise the GPU is even
    imported); here the ranks are replaced by a stub."""
    code = ("import sys, runpy, subprocess\n"
            "class P(object):\n"
            "    stdout = stderr = None\n"
            "    def __init__(self, cmd, env=None, **kw): print('SPAWN', 'torch' in sys.modules, env['RANK'], env['WORLD_SIZE'])\n"
            "    def poll(self): return 0\n"
            "subprocess.Popen = P\n"
            "sys.argv = ['bench.py', '--gpus', '2']\n"
            "runpy.run_path(%r, run_name='__main__')\n" % os.path.join(REPO, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "SPAWN False 0 2" in r.stdout and "SPAWN False 1 2" in r.stdout


@pytest.mark.parametrize("name", ["auto_n1000000_d27_k10_C3", "cross_n1000000_d15_k4_C4", "auto_n100000_d6_k4_C2"])
def test_lnE_from_the_reduced_sums_is_the_references(name):
    """bench.lnE_from_dotp restates MCEvidence.py:1120-1131 + the slice of :1157 -- fed the reference's own dotp and scalars
    it must give the reference's ln E (auto: k_nn = 1..kmax-1; cross: 2..kmax)."""
    import bench
    c = G[name]
    cfg = dict(k0=c["k0"], kmax=c["kmax"], S=c["S"], SumW=c["SumW"], J=c["J"], logLmax=c["logLmax"])
    got = bench.lnE_from_dotp(np.array(c["dotp"]), cfg, c["lnPriorVolume"])
    assert got.shape == (c["kmax"] - 1,) and np.max(np.abs(got - np.array(c["lnE"]))) < 1e-12


def test_counters_of_a_profile_from_other_sources_are_flagged_stale():
    import bench
    desc = "knn_f16_kernel<KST=2,KCAP=12> symmetric panel-kernel grid=5964"
    c = bench.profile_counters(desc, library_hash="not-the-hash-of-any-profile")
    if not c:
        pytest.skip("no committed profile of the panel kernel")
    assert c["stale"] is True
    if c.get("profile_source_hash"):
        assert bench.profile_counters(desc, library_hash=c["profile_source_hash"])["stale"] is False


def test_library_carries_the_digest_of_the_sources_in_the_tree():
    import bench
    from mcevidence_amd import _capi
    assert _capi.source_hash() == bench.source_hash()


def _bench_line(args, env_extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_gpus_2_from_a_plain_shell():
    """`python bench.py --gpus 2` with no launcher around it (what the driver's SCALE run does): the parent starts two ranks,
    every config is timed under N = 2 with one all-reduce per step, and the sums equal the single-rank run's.  Both ranks
    share this box's one GPU (MCE_BENCH_ONE_DEVICE=1) over gloo; sizes scaled down."""
    common = ["--steps", "2", "--warmup", "1", "--n", "150000", "--d", "27", "--kmax", "10", "--cpu-sample", "0", "--extras-scale", "0.05"]
    two = _bench_line(["--gpus", "2"] + common, dict(MCE_BENCH_BACKEND="gloo", MCE_BENCH_ONE_DEVICE="1"))
    one = _bench_line(["--gpus", "1"] + common, {})
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["backend"] == "gloo" and one["n_gpus"] == 1
    assert [r["rank"] for r in two["per_rank"]] == [0, 1] and all(r["device"] == 0 for r in two["per_rank"])
    # both partitions in the line by default; at 150 k rows (293 blocks on 2 ranks) the all-pairs-once partition applies
    assert "error" not in two["pairs_once"] and two["pairs_once"]["max_abs_dlnE_vs_default_partition"] < LNE_TOL, two["pairs_once"]
    assert two["partitions"]["value_is_of"] == two["config"]["partition"] and one["partitions"] is None and one["config"]["partition"] is None
    assert two["ms_per_step"] == min(two["partitions"]["default"]["ms_per_step"], two["partitions"]["pairs_once"]["ms_per_step"])
    assert two["steps"] == 2 and two["warmup"] == 1 and two["scaling"] == "strong"
    e2e = two["evidence_call_from_host"]                  # the class under the process group: part feed + one all-reduce
    assert len(e2e["per_rank"]) == 2 and e2e["max_abs_dlnE_vs_resident_path"] < LNE_TOL
    assert np.max(np.abs(np.array(two["lnE"]) - np.array(one["lnE"]))) < LNE_TOL
    for name in ("C2", "C4", "C5"):
        a, b = two["configs"][name], one["configs"][name]
        assert a["ranks"] == 2 and len(a["per_rank"]) == 2 and a["ms_per_step"] > 0
        assert np.max(np.abs(np.array(a["lnE"]) - np.array(b["lnE"]))) < LNE_TOL, name
    assert sum(r["query_rows"] for r in two["configs"]["C4"]["per_rank"]) == two["configs"]["C4"]["nq"]      # cross: row shards of s1
    assert "pruned" in two["configs"]["C5"]["kernel"] and "pruned" in one["configs"]["C5"]["kernel"]


@pytest.mark.gpu
def test_bench_gpus_2_under_the_launcher():
    """the other way to start it -- `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 --steps K --warmup W`,
    as the driver's contract spells it: WORLD_SIZE is set, the process is one rank and must NOT spawn again.  Full-size C3,
    headline only, both ranks on this box's one GPU over gloo: ln E of the all-reduced sums equals the reference's."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.pop("MCE_BENCH_PAIRS_ONCE", None)                  # (round 6: the second partition is timed by DEFAULT under two ranks or more)
    env.pop("MCE_PAIRS_ONCE", None)
    env.update(MCE_BENCH_BACKEND="gloo", MCE_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "0", "--no-extras"], capture_output=True, text=True, env=env, timeout=900, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["config"]["N"] == 1_000_000 and line["config"]["D"] == 27 and line["max_abs_dlnE_vs_reference"] < LNE_TOL
    assert "symmetric" in line["roofline"]["kernel"] and len(line["per_rank"]) == 2
    # the same workload through the all-pairs-once partition, collectives and all, in the same line -- WITHOUT any opt-in in the
    # environment: both partitions with their per-rank figures, equal ln E
    po = line["pairs_once"]
    assert "error" not in po and "skipped" not in po, po
    assert "pairs-once" in po["kernel"] and po["max_abs_dlnE_vs_default_partition"] < LNE_TOL and po["ms_per_step"] > 0
    assert sum(po["candidates_sent"]) == sum(po["candidates_received"]) > 0
    assert [r["rank"] for r in po["per_rank"]] == [0, 1] and all(r["ms_per_step"] > 0 and r["kernel_ms"] > 0 for r in po["per_rank"])
    # `value` is the faster of the two (both timed over the same steps), and the line says which: ms_per_step, the kernel the roofline
    # is of and the per-rank figures all belong to that one
    pt = line["partitions"]
    fastest = "pairs_once" if pt["pairs_once"]["ms_per_step"] < pt["default"]["ms_per_step"] else "default"
    assert pt["value_is_of"] == line["config"]["partition"] == fastest
    assert line["ms_per_step"] == pt[fastest]["ms_per_step"] and line["roofline"]["kernel"] == pt[fastest]["kernel"]
    assert abs(line["value"] - 1_000_000 / (line["ms_per_step"] * 1e-3)) < 1.0
    assert ("pairs-once" in line["roofline"]["kernel"]) == (fastest == "pairs_once") and line["roofline"]["traffic"] is None
    assert ("candidates_sent" in line["per_rank"][0]) == (fastest == "pairs_once")
    e2e = line["evidence_call_from_host"]
    assert len(e2e["per_rank"]) == 2 and e2e["max_abs_dlnE_vs_resident_path"] < LNE_TOL


@pytest.mark.gpu
def test_beyond_baseline_shapes_section():
    """the default one-GPU line also times the kernels that serve rows longer than the BASELINE configs' (round 6): the deep fp16 filter
    at d = 64 / 100 / 127 (and its two passes at kmax = 25), the long-row fp64 sweep at d = 128 / 256 -- each with a roofline object"""
    import torch
    import torch.distributed as dist
    import bench
    from mcevidence_amd import _capi
    ctx = bench.Ctx(torch, dist, _capi, 1, 0, torch.device("cuda", 0))
    out = bench.beyond_baseline_shapes(ctx)
    assert [(o["d"], o["kmax"]) for o in out] == [(64, 10), (100, 10), (127, 10), (100, 25), (128, 10), (256, 10)]
    assert all("error" not in o for o in out), out
    for o in out:
        want = "knn_long_kernel" if o["d"] >= 128 else "knn_deep_kernel"
        assert want in o["kernel"] and ("two passes" in o["kernel"]) == (o["kmax"] == 25), o["kernel"]
        r = o["roofline"]
        assert r["peak"] == (78.6 if o["d"] >= 128 else 2500.0) and 0.05 < r["frac"] < 1.0 and 0 < o["kernel_ms"] <= o["ms"]
    assert out[0]["ms"] < 8.0 and out[4]["ms"] < 120.0          # (100 k x 100 k x 64 asked for <= 6 ms; d = 128 took 278 on the vector-FMA kernel)
