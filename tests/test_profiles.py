"""The committed measurement artefacts are consistent with each other: the roofline figures bench.py prints -- the headline's and,
since round 5, every config's own -- follow from the committed rocprofv3 summaries (profiles/r06_final/): executed MFMA flops from
SQ_INSTS_MFMA, vector instructions from SQ_INSTS_VALU, the kernels' durations from the kernel traces; the profile names the kernel
sources it was taken from; per-config traces and counter passes let every config's figure be recomputed.  CPU only (reads files)."""
import csv
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
PROF = os.path.join(REPO, "profiles", "r06_final")
PROF5 = os.path.join(REPO, "profiles", "r05_final")
PROF4 = os.path.join(REPO, "profiles", "r04_final")


def _avg_ms(stats_file, needle, prof=PROF):
    for r in csv.DictReader(open(os.path.join(prof, stats_file))):
        if needle in r["Name"]:
            return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    raise AssertionError("%s: no kernel matching %r" % (stats_file, needle))


def _per_dispatch(summary, needle, counter):
    for ln in open(os.path.join(PROF, summary)):
        col = ln.strip().split(",")
        if needle in ln and len(col) >= 4 and col[-3] == counter:
            return float(col[-2]) / int(col[-1])
    raise AssertionError("%s: no %s row for %r" % (summary, counter, needle))


def test_roofline_fraction_follows_from_the_profile():
    import bench
    line = json.load(open(os.path.join(PROF, "bench.json")))
    meta = json.load(open(os.path.join(PROF, "meta.json")))
    roof = line["roofline"]
    assert meta["source_hash"] == meta["library_source_hash"]          # the library that was profiled was built from the tree that was hashed
    c = bench.profile_counters(roof["kernel"], library_hash=meta["source_hash"])
    assert c["source"] == "profiles/r06_final/pmc_summary.csv" and c["stale"] is False
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


def test_every_config_carries_a_roofline_that_its_own_counter_pass_reproduces():
    """round 5: `configs.C2 / C4 / C5.roofline` in the bench line -- bound, achieved, peak, frac, the kernel's duration -- and the
    same from that config's own rocprofv3 passes (kernel_stats_<C>.csv, pmc_summary_<C>.csv): C2 and C4 against the fp16 MFMA
    peak from SQ_INSTS_MFMA, C5 (the pruned walk) against the vector-ISSUE peak from SQ_INSTS_VALU, its MFMA fraction beside it"""
    allc = json.load(open(os.path.join(PROF, "bench_all_configs.json")))
    needles = {"C2": "knn_f16_kernelILi1ELi4ELb0ELb0ELi0ELi4ELi2E", "C4": "knn_f16_kernelILi1ELi4ELb0ELb0ELi0ELi4ELi4E", "C5": "knn_f16_kernelILi1ELi12ELb1E"}
    for name, needle in needles.items():
        cfg = allc["configs"][name]
        roof = cfg["roofline"]
        ms, calls = _avg_ms("kernel_stats_%s.csv" % name, needle)
        assert calls <= 2
        assert abs(ms / cfg["kernel_ms"] - 1.0) < 0.10, (name, ms, cfg["kernel_ms"])        # different boxes of the pool: +- 4 %
        assert roof["counters_stale"] is False and roof["counters_source"] == "profiles/r06_final/pmc_summary_%s.csv" % name
        assert abs(roof["kernel_ms"] - cfg["kernel_ms"]) < 1e-3
        if name == "C5":
            # round 6: the walk's peak is its own instruction MIX (SQ_INSTS_VALU_* of this config's committed pass) priced with the issue
            # cost MEASURED per class (profiles/r06_valu/valu_issue_clock.json) -- recomputed here from the two committed files
            assert roof["bound"] == "valu_issue" and roof["unit"] == "Ginst/s"
            valu = _per_dispatch("pmc_summary_C5.csv", needle, "SQ_INSTS_VALU")
            assert abs(valu / roof["valu_insts_per_launch"] - 1.0) < 1e-6
            assert abs(valu / (roof["kernel_ms"] * 1e-3) / 1e9 / roof["achieved"] - 1.0) < 1e-3
            table = json.load(open(os.path.join(REPO, "profiles", "r06_valu", "valu_issue_clock.json")))["classes"]
            cost = lambda cls: table[cls]["simd_cycles_per_inst_saturated"]
            assert 2.5 < cost("v_mov_b32") < cost("v_fma_f32") < 3.0 and 4.0 < cost("v_cmp_lt_f32+v_cndmask_b32") < cost("v_min3_f32") < 4.6      # no class at the guide's 2
            priced = {"ADD_F32": "v_fma_f32", "MUL_F32": "v_fma_f32", "FMA_F32": "v_fma_f32", "ADD_F64": "v_add_f64", "MUL_F64": "v_fma_f64", "FMA_F64": "v_fma_f64",
                      "INT32": "v_add_u32/v_lshlrev/v_and", "INT64": "v_add_u32/v_lshlrev/v_and", "CVT": "v_cvt_f32_f64/v_cvt_f64_f32",
                      "TRANS_F32": "v_cvt_f32_f64/v_cvt_f64_f32", "TRANS_F64": "v_cvt_f32_f64/v_cvt_f64_f32"}
            cycles, classified = 0.0, 0.0
            for cname, cls in priced.items():
                n = _per_dispatch("pmc_summary_C5.csv", needle, "SQ_INSTS_VALU_" + cname)
                cycles += n * cost(cls)
                classified += n
            other = valu - _per_dispatch("pmc_summary_C5.csv", needle, "SQ_INSTS_MFMA") - classified
            assert 0.55 < other / valu < 0.7                                                       # compares, selects, min / max, moves: two thirds of the walk
            cycles += other * cost("v_cmp_lt_f32+v_cndmask_b32")
            floor_ms = cycles / 1024 / 2.4e9 * 1e3
            assert abs(floor_ms / roof["issue_floor_ms"] - 1.0) < 1e-3 and abs(floor_ms / roof["kernel_ms"] / roof["frac"] - 1.0) < 1e-3
            assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 2e-3 and 0.6 < roof["frac"] < 0.75
            assert abs(sum(v["share_of_issue_cycles"] for v in roof["issue_cycles_by_class"]["classes"].values()) - 1.0) < 1e-3
            assert 0.02 < roof["mfma_frac"] < 0.06 and cfg["pruned_walk"]["tile_fraction"] < 0.003
        else:
            assert roof["bound"] == "mfma" and roof["peak"] == 2500.0
            mfma = _per_dispatch("pmc_summary_%s.csv" % name, needle, "SQ_INSTS_MFMA")
            # executed flops of the launch: the library's count (mce_last_search_stats) against the counter (the counter pass also
            # holds the launch that writes the distances: same kernel, same work)
            assert abs(mfma * 32768.0 / roof["executed_flops_per_launch"] - 1.0) < 0.02, name
            assert abs(roof["frac"] - roof["executed_flops_per_launch"] / (roof["kernel_ms"] * 1e-3) / 2.5e15) < 2e-3
    assert 0.30 < allc["configs"]["C4"]["roofline"]["frac"] < 0.42 and "wide" in allc["configs"]["C4"]["kernel"]
    assert 0.15 < allc["configs"]["C2"]["roofline"]["frac"] < 0.25
    # fp64 sweep: 10^12 pairs x 2 x 4 x 7 flop
    ms64, _ = _avg_ms("kernel_stats_fp64.csv", "knn_mfma_kernel<7, 12>")
    frac64 = 1e12 * 56.0 / (ms64 * 1e-3) / 78.6e12
    assert abs(frac64 / allc["fp64_mode"]["roofline"]["frac"] - 1.0) < 0.04 and 0.7 < frac64 < 0.85
    # round 6: a counter pass of the fp64-mode kernel too (executed flops from SQ_INSTS_VALU_MFMA_MOPS_F64 or the MFMA count x 2 x 16 x 16 x 4)
    mf = _per_dispatch("pmc_summary_fp64.csv", "knn_mfma_kernel<7, 12>", "SQ_INSTS_MFMA")
    assert abs(mf * 2048.0 / (1e12 * 56.0) - 1.0) < 0.02                    # v_mfma_f64_16x16x4_f64 = 2048 flop: padded rows and queries, all pairs
    # ln E of every config against the reference's own output, in the same line
    for name in ("C2", "C4", "C5"):
        assert allc["configs"][name]["max_abs_dlnE_vs_reference"] < 1e-9
    assert allc["max_abs_dlnE_vs_reference"] < 1e-9
    assert allc["cpu_baseline"]["kind"] == "reference" and "100000 random query rows" in allc["cpu_baseline"]["sample"]


def test_predicted_scaling_is_labelled_and_adds_up():
    cfgs = json.load(open(os.path.join(PROF, "predicted_scaling.json")))
    assert {c["config"] for c in cfgs} == {"C3", "C4", "C5"}
    for cfg in cfgs:
        assert "PREDICTED" in cfg["label"]
        for w, r in cfg["worlds"].items():
            assert len(r["rank_ms"]) == int(w) and r["predicted_step_ms"] == max(r["rank_ms"])
            assert r["max_rel_dev_of_summed_dotp_vs_1gpu"] < 1e-12
        assert cfg["max_abs_dlnE_vs_reference"] < 1e-9
    by = {c["config"]: c["worlds"] for c in cfgs}
    assert by["C5"]["8"]["predicted_step_ms"] < 18.0                   # round 4: 20.7 (preparation 11.8 of it); round 3: 102.5
    assert by["C5"]["1"]["predicted_step_ms"] < 78.0                   # round 4: 79.1
    assert "wide" in by["C4"]["2"]["kernel"] and by["C4"]["2"]["efficiency"] > 0.95 and by["C4"]["4"]["efficiency"] > 0.93


def test_pairs_once_emulation_is_labelled_adds_up_and_beats_the_default_partition():
    """the all-pairs-once partition at C3, W ranks emulated on one GPU -- the table's figures, the balance of
    the ranks, the sums against the one-GPU call's, and the comparison with the default partition measured in the same process"""
    r = json.load(open(os.path.join(PROF, "pairs_once_emulated.json")))
    assert "PREDICTED" in r["label"] and r["config"] == "C3" and r["n"] == 1_000_000
    eff = {}
    for w, v in r["pairs_once"].items():
        W = int(w)
        assert len(v["rank_ms"]) == W and v["predicted_step_ms"] == max(v["rank_ms"]) and "pairs-once" in v["kernel"]
        assert max(v["rank_ms"]) / min(v["rank_ms"]) < 1.06                          # cyclic ownership: balanced ranks
        assert sum(v["candidates_sent"]) == sum(v["candidates_received"]) and v["flagged_blocks"] == 0
        assert v["max_rel_dev_of_summed_dotp_vs_1gpu"] < 1e-12 and v["max_abs_dlnE_vs_reference"] < 1e-9
        for i in range(W):                                                           # a rank's step is the sum of its parts
            parts = v["prepare_ms"][i] + v["sweep_ms"][i] + v["export_ms"][i] + v["finish_ms"][i] + v["exchange_ms_priced"][i]
            assert abs(parts - v["rank_ms"][i]) < 0.01
        eff[W] = r["one_gpu_ms"] / v["predicted_step_ms"] / W
        assert abs(eff[W] - v["efficiency"]) < 2e-3
        assert v["predicted_step_ms"] < r["todays_partition"][w]["predicted_step_ms"]
    assert eff[2] > 0.85 and 0.75 < eff[4] < 0.82 and 0.62 < eff[8] < 0.72              # the verdict's 0.80: met at two ranks, all but met at four
    first = json.load(open(os.path.join(REPO, "profiles", "r05_mid", "pairs_once_contiguous.json")))
    assert first["pairs_once"]["2"]["predicted_step_ms"] > first["one_gpu_ms"]          # the version with contiguous ranges: worse than one GPU
    assert first["pairs_once"]["2"]["candidates_sent"][0] > 30 * first["pairs_once"]["2"]["candidates_sent"][1]


def test_c5_preparation_in_the_kernel_trace():
    """seven sorts instead of thirteen (key kernel calls of one traced call), the bottom levels, chunk lists
    and merge at their new durations; everything outside the walk under 9 ms"""
    rows = list(csv.DictReader(open(os.path.join(PROF5, "kernel_stats_C5.csv"))))
    calls = max(int(r["Calls"]) for r in rows if "knn_f16_kernelILi1ELi12ELb1E" in r["Name"])        # searches in the traced process
    get = lambda needle: [r for r in rows if needle in r["Name"]]
    assert sum(int(r["Calls"]) for r in get("kd_key_kernel")) == 7 * calls
    assert float(get("kd_bottom_kernel")[0]["AverageNs"]) < 1.3e6          # round 4: 1.41 ms
    assert float(get("chunk_list_kernel")[0]["AverageNs"]) < 0.9e6         # 1.33 ms
    # (this trace is of tools/run_configs.py, which also asks for the distance matrix: the merge then writes 720 MB of rows in the
    #  caller's order; the fused call bench.py times is traced in kernel_stats_C5_fused_call.csv -- tools/c5_ab.sh)
    fused = list(csv.DictReader(open(os.path.join(PROF5, "kernel_stats_C5_fused_call.csv"))))
    assert float([r for r in fused if "merge_lists_kernel<false, true, false>" in r["Name"]][0]["AverageNs"]) < 1.0e6        # round 4: 2.05 ms
    fcalls = max(int(r["Calls"]) for r in fused if "knn_f16_kernelILi1ELi12ELb1E" in r["Name"])
    walk = sum(float(r["TotalDurationNs"]) for r in fused if "knn_f16_kernelILi1ELi12ELb1E" in r["Name"])
    rest = sum(float(r["TotalDurationNs"]) for r in fused if "knn_f16_kernelILi1ELi12ELb1E" not in r["Name"] and "at::native" not in r["Name"]
               and "copyBuffer" not in r["Name"])
    assert rest / fcalls < 8.5e6 and walk / fcalls < 70e6, (rest / fcalls, walk / fcalls)          # round 4: 12.7 ms outside the walk


def test_round4_walk_numbers_quoted_in_the_design_notes():
    """docs/design/pruned_walk.md (round 4): the counters and the cycle breakdown it quotes are committed next to that round's traces"""
    pmc = open(os.path.join(PROF4, "c5_pmc.txt")).read()
    vals = {l.split()[0]: float(l.split()[1]) for l in pmc.splitlines() if l.startswith("  SQ_")}
    per_wave = vals["SQ_INSTS_VALU"] / vals["SQ_WAVES"]
    assert 1.5e5 < per_wave < 2.2e5 and vals["SQ_INSTS_MFMA"] / vals["SQ_WAVES"] < 1500
    lines = [l for l in open(os.path.join(PROF4, "c5_walk_breakdown.txt")) if l.startswith("[prune prof]")]
    assert len(lines) == 3
    wt = json.load(open(os.path.join(PROF4, "c5_wave_times.json")))
    assert wt["quantiles_us"]["1.0"] < 6000 and wt["mean_us"] < 1600          # (no heavy waves left: 81 ms mid-round)


def test_mfma_error_model_histogram_is_committed():
    h = json.load(open(os.path.join(PROF4, "mfma_error_model.json")))
    assert h["tiles_total"] >= 10000 and h["max_over_everything"] < 0.5


def test_deep_and_long_kernels_have_their_own_trace_and_counter_rows():
    """round 6: the kernels beyond the BASELINE configs' shapes -- the deep fp16 filter (64 <= d <= 127) and the long-row fp64 sweep
    (d >= 128) -- are in the final profile too: 100 k x 100 k auto evidence at d = 64 / 100 / 127 / 128 / 256 under the kernel trace and
    an SQ counter pass (kernel_stats_shapes.csv, pmc_summary_shapes.csv, shape_times.txt).  The flops DESIGN.md 3 quotes follow
    from SQ_INSTS_MFMA: 32 768 per fp16 MFMA (32x32x16), 2 048 per fp64 MFMA (16x16x4)."""
    n = 100000.0
    # the long-row sweep at d = 128 and 256 share one kernel row (KCAP = 16): two dispatches per shape x (1 + 3 reps)
    ms_long, calls = _avg_ms("kernel_stats_shapes.csv", "knn_long_kernel<16>")
    assert calls >= 8 and 70.0 < ms_long < 100.0            # mean of the d = 128 (~58 ms) and d = 256 (~116 ms) launches
    mf = _per_dispatch("pmc_summary_shapes.csv", "knn_long_kernel<16>", "SQ_INSTS_MFMA")
    # KSP = 35 (d = 128) and 72 (d = 256) k-steps, padded rows of 1173 x 256 queries by the padded references: mean over the two shapes
    nq_pad, nr_pad = 391 * 256.0, 782 * 128.0
    want = nq_pad * nr_pad * (35 + 72) / 2.0 / 256.0          # one fp64 MFMA = 16 x 16 pairs x 4 dimensions
    assert abs(mf / want - 1.0) < 0.02, (mf, want)
    tf = mf * 2048.0 / (ms_long * 1e-3) / 1e12
    assert 40.0 < tf < 60.0                                    # 0.5 - 0.75 of the 78.6 TFLOP/s fp64 MFMA peak
    for kst, lo, hi in ((5, 2.5, 5.0), (8, 3.0, 6.5)):
        ms_deep, _ = _avg_ms("kernel_stats_shapes.csv", "knn_deep_kernel<%d, 12" % kst)
        assert lo < ms_deep < hi, (kst, ms_deep)
    times = open(os.path.join(PROF, "shape_times.txt")).read()
    assert "knn_long_kernel<KCAP=16>" in times and "knn_deep_kernel<KST=5,KCAP=12>" in times
