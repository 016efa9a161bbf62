"""The committed measurement artefacts are consistent with each other: the roofline figure bench.py prints follows from the
committed rocprofv3 summaries (profiles/r03_panel/) -- executed MFMA flops from SQ_INSTS_MFMA, the kernel's duration from the
kernel trace.  CPU only (reads files)."""
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
PROF = os.path.join(REPO, "profiles", "r03_panel")


@pytest.mark.skipif(not os.path.exists(os.path.join(PROF, "bench.json")), reason="no round-3 profile committed")
def test_roofline_fraction_follows_from_the_profile():
    import bench
    line = json.load(open(os.path.join(PROF, "bench.json")))
    roof = line["roofline"]
    c = bench.profile_counters(roof["kernel"])
    assert c["source"] == "profiles/r03_panel/pmc_summary.csv"
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


def test_predicted_scaling_is_labelled_and_adds_up():
    p = os.path.join(PROF, "predicted_scaling.json")
    if not os.path.exists(p):
        pytest.skip("no predicted scaling committed")
    for cfg in json.load(open(p)):
        assert "PREDICTED" in cfg["label"]
        for w, r in cfg["worlds"].items():
            assert len(r["rank_ms"]) == int(w) and r["predicted_step_ms"] == max(r["rank_ms"])
            assert r["max_rel_dev_of_summed_dotp_vs_1gpu"] < 1e-12
