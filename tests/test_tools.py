"""Developer tools that DESIGN.md quotes results from must keep running: a smoke test of tools/vgpr_liveness.py on a small listing."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LISTING = """
	.text
_Z6kernelPf:                            ; @kernel
	v_mov_b32_e32 v0, 1.0
	v_mov_b32_e32 v1, 2.0
	global_load_dword v2, v[4:5], off
.LBB0_1:
	v_add_f32_e32 v3, v0, v1
	v_fmac_f32_e32 v3, v2, v2
	s_cbranch_scc1 .LBB0_1
	global_store_dword v[4:5], v3, off
	s_endpgm
"""


def test_vgpr_liveness_on_a_small_listing(tmp_path):
    p = tmp_path / "k.s"
    p.write_text(LISTING)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "vgpr_liveness.py"), str(p), "kernel", "--top", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    # v0, v1, v2 (loop-carried uses), v3 and the address pair v4:v5 are live inside the loop: six registers at the peak
    assert "peak live VGPRs 6" in out, out
    assert "global_load_dword v2" in out and "v4,5" in out.replace(" ", "")


def test_fuzz_tools_draw_legal_cases_and_run_without_a_gpu():
    """tools/fuzz_shapes.py and tools/fuzz_class.py are the round's randomised parity evidence (profiles/r06_final/fuzz_*.json): their
    draws must stay legal (K within the usable rows, sizes within the budget, the same draws for the same seed), and the class fuzz's
    whole loop runs here with the oracle on both sides (MCE_FUZZ_CPU_ONLY=1)."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("fuzz_shapes_draws", os.path.join(REPO, "tools", "fuzz_shapes.py"))
    src = open(os.path.join(REPO, "tools", "fuzz_shapes.py")).read()
    # (the module imports the library at the top: take its pure functions only)
    ns = {"np": np}
    start, stop = src.index("KINDS = "), src.index("def run_case")
    exec(compile(src[start:stop], "fuzz_shapes_draws", "exec"), ns)
    a = [ns["draw"](np.random.default_rng(5), 60000) for _ in range(3)]
    rng = np.random.default_rng(5)
    cases = [ns["draw"](rng, 60000) for _ in range(400)]
    assert cases[0] == a[0]
    for c in cases:
        usable = c["nr"] - (1 if c["self_mode"] == 2 else 0)
        assert 1 <= c["d"] <= 140 and 1 <= c["K"] <= 40 and usable >= min(c["K"], usable) and c["nr"] <= 60000 + 41
        assert (c["nq"] == c["nr"]) if c["same"] else c["self_mode"] == 0
        assert ns["make_rows"](np.random.default_rng(c["seed"]), c["kind"], 7, c["d"]).shape == (7, c["d"])
    big = [ns["draw"](np.random.default_rng(i), 250000, 20000, [128, 700]) for i in range(50)]
    assert all(128 <= c["d"] <= 700 and c["nr"] >= 20000 for c in big)
    env = dict(os.environ, MCE_FUZZ_CPU_ONLY="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fuzz_class.py"), "--seconds", "4", "--seed", "3", "--max-rows", "2000"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["draws"] >= 1 and line["mismatches"] == 0
