#!/usr/bin/env python
"""Summary of tools/valu_issue_clock (gpurun_out/valu_issue/clock.jsonl + counters.csv, written by tools/valu_issue.sh on the GPU box)
-> profiles/<tag>/valu_issue_clock.json: per instruction class the issue cost of one wave64 instruction
  * for ONE wave alone on its SIMD (cycles between two independent instructions of the same wave),
  * for the SIMD when 1 - 4 waves issue the class (wall time x SIMDs / instructions, in cycles at the nominal 2.4 GHz -- what a
    roofline needs; the saturated value is the minimum over the wave counts),
  * for a wave that shares its SIMD with a wave streaming v_mfma_f32_32x32x16_f16 (and what that costs the MFMA wave),
and which SQ_INSTS_VALU_* counter counts the class.  bench.py prices the pruned walk's instruction mix with it.
usage: python tools/valu_issue_report.py r06_valu"""
import collections, csv, json, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06_valu"
src = os.path.join(REPO, "gpurun_out", "valu_issue")
rows = [json.loads(l) for l in open(os.path.join(src, "clock.jsonl"))]
names = ["v_fma_f32", "v_min3_f32", "v_cmp_lt_f32", "v_cmp_lt_f32+s_cbranch_vccnz", "v_pk_fma_f32", "v_fma_f64", "v_add_f64", "v_add_u32/v_lshlrev/v_and",
         "v_cndmask_b32", "v_mov_b32", "v_cvt_f32_f64/v_cvt_f64_f32", "v_readlane_b32", "v_max_f32/v_min_f32", "v_cmp_lt_f32+v_cndmask_b32"]
cnt = collections.defaultdict(dict)
if os.path.exists(os.path.join(src, "counters.csv")):
    for r in csv.DictReader(open(os.path.join(src, "counters.csv"))):
        m = re.search(r"_Z1kILi(\d+)ELi(\d+)ELi(\d+)E", r["kernel"])
        if m and m.group(2) == "0":
            cnt[names[int(m.group(1))]][r["counter"]] = float(r["value_last_dispatch"])
out = dict(tool="tools/valu_issue_clock.hip", device="MI355X (gfx950), 256 CUs", nominal_clock_ghz=2.4, classes={}, interleaved=[],
           notes=["cycles_at_2p4GHz_per_inst_per_simd = kernel span (s_memrealtime, 100 MHz) x SIMDs used / instructions issued x 2.4: wall time in nominal cycles",
                  "cycles_per_inst_per_wave = s_memtime ticks of a wave / its instructions (shader cycles; the clock held 2.3 - 2.4 GHz in every 'alone' run)",
                  "SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.00 for EVERY class: the counter counts quad-cycles of issue, not an issue rate",
                  "SQ_INST_CYCLES_VALU reads 0 on this part; SQ_INSTS_VALU includes the MFMAs"])
for name in names:
    rs = [r for r in rows if r["class"] == name]
    if not rs:
        continue
    alone = {r["waves_per_simd"]: r for r in rs if r["role"] == "alone"}
    beside = [r for r in rs if r["role"].startswith("beside")]
    c = cnt.get(name, {})
    tot = c.get("SQ_INSTS_VALU", 0.0)
    counted = {k.replace("SQ_INSTS_VALU_", ""): round(v / tot, 3) for k, v in c.items() if k.startswith("SQ_INSTS_VALU_") and tot and v / tot > 0.01}
    e = dict(one_wave_cycles_per_inst=round(alone[1]["cycles_per_inst_per_wave"], 2) if 1 in alone else None,
             simd_cycles_per_inst_by_waves={str(w): round(alone[w]["cycles_at_2p4GHz_per_inst_per_simd"], 2) for w in sorted(alone)},
             simd_cycles_per_inst_saturated=round(min(r["cycles_at_2p4GHz_per_inst_per_simd"] for r in alone.values()), 2) if alone else None,
             counted_by=counted or "no SQ_INSTS_VALU_* class counter (only SQ_INSTS_VALU)")
    if beside:
        e["beside_an_mfma_wave"] = dict(cycles_per_inst=round(beside[0]["cycles_per_inst_per_wave"], 2), mfma_wave_cycles_per_mfma=beside[0].get("mfma_wave_cycles_per_mfma"))
    out["classes"][name] = e
for r in rows:
    if r["role"].startswith("MFMAs"):
        out["interleaved"].append(dict(cls=r["class"], waves_per_simd=r["waves_per_simd"], mfma_per_32_valu=r["mfma_per_32"], cycles_per_group=r.get("cycles_per_group"),
                                       mfma_floor=r.get("mfma_floor_cycles_per_group")))
dst = os.path.join(REPO, "profiles", tag)
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, "valu_issue_clock.json"), "w"), indent=1)
for k, v in out["classes"].items():
    print("%-32s one wave %5s  SIMD by waves %s  saturated %s  beside MFMA %s  counted by %s" % (k, v["one_wave_cycles_per_inst"], v["simd_cycles_per_inst_by_waves"], v["simd_cycles_per_inst_saturated"],
                                                                                             v.get("beside_an_mfma_wave"), v["counted_by"]))
