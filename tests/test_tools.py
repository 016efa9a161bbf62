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
