#!/bin/bash
# GPU box: everything profiles/<tag>/ holds that depends on the kernel sources, in ONE call -- tools/profile_bench.sh <tag>, its
# results copied into profiles/<tag>/ of the box's copy of the tree (so that the counters match the library's source hash), then a
# full `python bench.py` with them in place -> gpurun_out/prof_<tag>/bench_all_configs.json.  Back in the build container:
#   cp gpurun_out/prof_<tag>/{*.json,*.csv} profiles/<tag>/
# usage: tools/refresh_profiles.sh r05_final
tag=${1:?tag}
cd "$(dirname "$0")/.."
R=$PWD
tools/profile_bench.sh $tag > gpurun_out/refresh_$tag.log 2>&1 || { echo "profile_bench.sh failed"; tail -5 gpurun_out/refresh_$tag.log; exit 1; }
out=$R/gpurun_out/prof_$tag
mkdir -p profiles/$tag
cp $out/bench.json $out/bench_fp64.json $out/kernel_stats*.csv $out/pmc_summary*.csv $out/meta.json profiles/$tag/ || exit 1
# (the issue-cost table of tools/valu_issue_clock.hip is box-independent: profiles/r06_valu/valu_issue_clock.json, committed)
python bench.py > $out/bench_all_configs.json 2> $out/bench_all_configs.err || { echo "bench.py failed"; tail -5 $out/bench_all_configs.err; exit 1; }
python - "$out/bench_all_configs.json" <<'PY'
import json, sys
p = json.load(open(sys.argv[1]))
print("C3", p["value"], p["ms_per_step"], p["roofline"]["kernel_ms"], p["roofline"]["frac"], "stale:", p["roofline"]["traffic_stale"], "cpu", p["cpu_baseline"]["value"])
for c, v in p["configs"].items():
    print(c, v["ms_per_step"], v["kernel_ms"], v["roofline"]["frac"], "stale:", v["roofline"]["counters_stale"], v["max_abs_dlnE_vs_reference"])
PY
