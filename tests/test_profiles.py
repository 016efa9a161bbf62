"""The committed measurement artefacts are consistent with each other: the roofline figure bench.py prints follows from the
committed rocprofv3 summaries (profiles/r04_final/) -- executed MFMA flops from SQ_INSTS_MFMA, the kernel's duration from the
kernel trace -- the profile names the kernel sources it was taken from, and per-config traces let every config's figure be
recomputed.  CPU only (reads files)."""
import csv
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
PROF = os.path.join(REPO, "profiles", "r04_final")


def _avg_ms(stats_file, needle):
    for r in csv.DictReader(open(os.path.join(PROF, stats_file))):
        if needle in r["Name"]:
            return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    raise AssertionError("%s: no kernel matching %r" % (stats_file, needle))


@pytest.mark.skipif(not os.path.exists(os.path.join(PROF, "bench.json")), reason="no round-4 profile committed")
def test_roofline_fraction_follows_from_the_profile():
    import bench
    line = json.load(open(os.path.join(PROF, "bench.json")))
    meta = json.load(open(os.path.join(PROF, "meta.json")))
    roof = line["roofline"]
    assert meta["source_hash"] == meta["library_source_hash"]          # the library that was profiled was built from the tree that was hashed
    c = bench.profile_counters(roof["kernel"], library_hash=meta["source_hash"])
    assert c["source"] == "profiles/r04_final/pmc_summary.csv" and c["stale"] is False
    # executed flops: one v_mfma_f32_32x32x16_f16 = 32 x 32 x 16 multiply-adds
    assert abs(c["mfma_insts"] * 32768.0 / roof["executed_flops_per_launch"] - 1.0) < 1e-3
    # duration: rocprofv3's average over the process's six searches vs the HIP-event bracket over its three timed ones
    assert abs(c["profile_kernel_ms"] / roof["kernel_ms"] - 1.0) < 0.04
    frac_from_profile = c["profile_tflops_from_SQ_INSTS_MFMA"] / roof["peak"]
    assert abs(frac_from_profile / roof["frac"] - 1.0) < 0.04
    assert roof["peak"] == 2500.0 and roof["bound"] == "mfma"
    # the all-pairs count is not what the fraction is made of
    assert roof["algorithmic_speedup"] > 1.9 and roof["all_pairs_flops"] > 1.9 * roof["executed_flops_per_launch"]
    assert 7.5 < c["valu_per_mfma"] < 9.0 and 0.3 < c["mfma_busy_frac"] < 0.4 and 0.4 < c["wait_frac"] < 0.55
    assert 1.5e10 < c["traffic"] < 1.9e10
    # a library built from other sources does not get these counters
    assert bench.profile_counters(roof["kernel"], library_hash="0" * 64)["stale"] is True


@pytest.mark.skipif(not os.path.exists(os.path.join(PROF, "bench_all_configs.json")), reason="no round-4 profile committed")
def test_every_config_can_be_recomputed_from_its_own_kernel_trace():
    """one kernel-trace file per config (round 3 merged C2's and C4's launches of knn_f16_kernel<1,4,..,0> into one row): the
    search kernel's average duration there agrees with what bench.py's configs object reports, and C4's and the fp64 sweep's
    fractions of peak follow from it"""
    allc = json.load(open(os.path.join(PROF, "bench_all_configs.json")))
    for name, needle in (("C2", "knn_f16_kernelILi1ELi4ELb0ELb0ELi0E"), ("C4", "knn_f16_kernelILi1ELi4ELb0ELb0ELi0E"), ("C5", "knn_f16_kernelILi1ELi12ELb1E")):
        ms, calls = _avg_ms("kernel_stats_%s.csv" % name, needle)
        assert calls <= 2
        assert abs(ms / allc["configs"][name]["kernel_ms"] - 1.0) < 0.08, (name, ms, allc["configs"][name]["kernel_ms"])       # different boxes of the pool: +- 4 %
    # C4: executed flops = (chunks + seed chunks) x 48 tiles x 16 query tiles x 32 768 flop per block -- the library's own count
    c4 = allc["configs"]["C4"]
    ms, _ = _avg_ms("kernel_stats_C4.csv", "knn_f16_kernelILi1ELi4ELb0ELb0ELi0E")
    assert 0.25 < c4["executed_tflops"] / 2500.0 < 0.40
    # fp64 sweep: 10^12 pairs x 2 x 4 x 7 flop
    ms64, _ = _avg_ms("kernel_stats_fp64.csv", "knn_mfma_kernel<7, 12>")
    frac64 = 1e12 * 56.0 / (ms64 * 1e-3) / 78.6e12
    assert abs(frac64 / allc["fp64_mode"]["roofline"]["frac"] - 1.0) < 0.04 and 0.7 < frac64 < 0.85
    # ln E of every config against the reference's own output, in the same line
    for name in ("C2", "C4", "C5"):
        assert allc["configs"][name]["max_abs_dlnE_vs_reference"] < 1e-9
    assert allc["max_abs_dlnE_vs_reference"] < 1e-9


def test_predicted_scaling_is_labelled_and_adds_up():
    p = os.path.join(PROF, "predicted_scaling.json")
    if not os.path.exists(p):
        pytest.skip("no predicted scaling committed")
    cfgs = json.load(open(p))
    assert {c["config"] for c in cfgs} == {"C3", "C4", "C5"}
    for cfg in cfgs:
        assert "PREDICTED" in cfg["label"]
        for w, r in cfg["worlds"].items():
            assert len(r["rank_ms"]) == int(w) and r["predicted_step_ms"] == max(r["rank_ms"])
            assert r["max_rel_dev_of_summed_dotp_vs_1gpu"] < 1e-12
        assert cfg["max_abs_dlnE_vs_reference"] < 1e-9
    c5 = [c for c in cfgs if c["config"] == "C5"][0]["worlds"]
    assert c5["8"]["predicted_step_ms"] < 25.0                      # round 3: 102.5 (one corner wave's 80 ms walk); mid-round 4: 43.7
    assert c5["1"]["predicted_step_ms"] < 90.0                     # round 3: 220.5; mid-round 4: 196.2 (before the per-query reach test)


def test_c5_walk_numbers_quoted_in_the_design_notes():
    """DESIGN.md 3.5 (second half): the pruned walk multiplies under 0.3 % of the tile pairs and its kernel is under 75 ms at C5;
    the counters and the cycle breakdown it quotes are committed next to the traces."""
    allc = json.load(open(os.path.join(PROF, "bench_all_configs.json")))
    c5 = allc["configs"]["C5"]
    assert c5["pruned_walk"]["tile_fraction"] < 0.003 and c5["kernel_ms"] < 75.0 and c5["queries_per_s"] > 1.1e8
    assert "lists=9" in c5["kernel"]                                # the nine-entry, three-wave instantiation
    pmc = open(os.path.join(PROF, "c5_pmc.txt")).read()
    vals = {l.split()[0]: float(l.split()[1]) for l in pmc.splitlines() if l.startswith("  SQ_")}
    per_wave = vals["SQ_INSTS_VALU"] / vals["SQ_WAVES"]
    assert 1.5e5 < per_wave < 2.2e5 and vals["SQ_INSTS_MFMA"] / vals["SQ_WAVES"] < 1500
    lines = [l for l in open(os.path.join(PROF, "c5_walk_breakdown.txt")) if l.startswith("[prune prof]")]
    assert len(lines) == 3
    wt = json.load(open(os.path.join(PROF, "c5_wave_times.json")))
    assert wt["quantiles_us"]["1.0"] < 6000 and wt["mean_us"] < 1600          # (no heavy waves left: 81 ms mid-round)


def test_mfma_error_model_histogram_is_committed():
    p = os.path.join(PROF, "mfma_error_model.json")
    if not os.path.exists(p):
        pytest.skip("no histogram committed")
    h = json.load(open(p))
    assert h["tiles_total"] >= 10000 and h["max_over_everything"] < 0.5
