#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + counters for `python bench.py` and for every BASELINE config (round-tagged).
# usage: tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/...
# Every pass writes into a directory of its own that is emptied first, and a pass that leaves no CSV stops the script: a stale
# file of an earlier run can never end up next to a fresh meta.json.
tag=${1:-r05}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQC1="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
SQC2="SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
need() { f=$(find $1 -name "$2" | head -1); [ -n "$f" ] || { echo "profile_bench.sh: $1 holds no $2 -- log:"; tail -5 $3; exit 1; }; echo $f; }
pass() {    # pass <dir> <log> <rocprofv3 options...> -- <program...>
  d=$1; log=$2; shift 2; rm -rf $d
  rocprofv3 "$@" > $log 2>&1
}
summarise() {    # summarise <out csv> <name:dir>...
  python3 - "$@" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
rows = []
for spec in sys.argv[2:]:
    name, d = spec.split(":")
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: sys.exit("no counter_collection.csv under " + d)
    for f in fs:
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = set()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:72]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            seen.add((k, r['Dispatch_Id']))
        for k in agg:
            nd = len([1 for kk, _ in seen if kk == k])
            for c, v in agg[k].items(): rows.append((k, c, v, nd))
with open(out, "w") as fh:
    fh.write("kernel,counter,sum_over_dispatches,dispatches\n")
    for r in rows: fh.write("%s,%s,%.6g,%d\n" % r)
PY
}
B="python3 $R/bench.py --cpu-sample 0 --no-extras"
pass /tmp/kt_$tag $out/bench_under_kernel_trace.log --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o kt -- $B --steps 3 --warmup 1
cp $(need /tmp/kt_$tag "*kernel_stats.csv" $out/bench_under_kernel_trace.log) $out/kernel_stats.csv || exit 1
# HBM traffic counters: separate passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
pass /tmp/pf_$tag $out/bench_under_fetch.log --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$tag -o pf -- $B --steps 1 --warmup 0
pass /tmp/pw_$tag $out/bench_under_write.log --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$tag -o pw -- $B --steps 1 --warmup 0
pass /tmp/ps_$tag $out/bench_under_sq.log --pmc $SQ --kernel-trace --output-format csv -d /tmp/ps_$tag -o ps -- $B --steps 1 --warmup 0
summarise $out/pmc_summary.csv F:/tmp/pf_$tag W:/tmp/pw_$tag S:/tmp/ps_$tag || exit 1
# the line the kernel-trace process printed (its HIP-event bracket over the same launches the trace averaged)
grep '^{"metric"' $out/bench_under_kernel_trace.log | tail -1 > $out/bench.json
[ -s $out/bench.json ] || { echo "profile_bench.sh: no bench line"; tail -5 $out/bench_under_kernel_trace.log; exit 1; }
# per-config kernel traces and SQ counters (one process each, so that a kernel shared by two configs gets one row per config)
for c in C2 C4 C5; do
  pass /tmp/kc_${tag}_$c $out/run_configs_$c.log --kernel-trace --stats --output-format csv -d /tmp/kc_${tag}_$c -o kt -- python3 $R/tools/run_configs.py $c
  cp $(need /tmp/kc_${tag}_$c "*kernel_stats.csv" $out/run_configs_$c.log) $out/kernel_stats_$c.csv || exit 1
  pass /tmp/pc_${tag}_$c $out/run_configs_${c}_sq.log --pmc $SQ --kernel-trace --output-format csv -d /tmp/pc_${tag}_$c -o ps -- python3 $R/tools/run_configs.py $c
  if [ $c = C5 ]; then
    # the pruned walk is priced by its instruction MIX (bench.py: valu_issue_floor): the class counters, in two more passes
    pass /tmp/pd_${tag}_$c $out/run_configs_${c}_sq2.log --pmc $SQC1 --kernel-trace --output-format csv -d /tmp/pd_${tag}_$c -o ps -- python3 $R/tools/run_configs.py $c
    pass /tmp/pe_${tag}_$c $out/run_configs_${c}_sq3.log --pmc $SQC2 --kernel-trace --output-format csv -d /tmp/pe_${tag}_$c -o ps -- python3 $R/tools/run_configs.py $c
    summarise $out/pmc_summary_$c.csv S:/tmp/pc_${tag}_$c C1:/tmp/pd_${tag}_$c C2:/tmp/pe_${tag}_$c || exit 1
  else
    summarise $out/pmc_summary_$c.csv S:/tmp/pc_${tag}_$c || exit 1
  fi
done
# the kernels beyond the BASELINE configs' shapes (round 6): the deep fp16 filter (64 <= d <= 127) and the long-row fp64 sweep (d >= 128),
# 100 k x 100 k auto evidence each -- one process, one kernel row per variant
SHAPES="100000,64,9 100000,100,9 100000,127,9 100000,128,9 100000,256,9"
pass /tmp/kh_$tag $out/shapes_under_kernel_trace.log --kernel-trace --stats --output-format csv -d /tmp/kh_$tag -o kt -- python3 $R/tools/shape_times.py $SHAPES
cp $(need /tmp/kh_$tag "*kernel_stats.csv" $out/shapes_under_kernel_trace.log) $out/kernel_stats_shapes.csv || exit 1
grep "^100000," $out/shapes_under_kernel_trace.log > $out/shape_times.txt
pass /tmp/ph_$tag $out/shapes_under_sq.log --pmc $SQ --kernel-trace --output-format csv -d /tmp/ph_$tag -o ps -- python3 $R/tools/shape_times.py $SHAPES
summarise $out/pmc_summary_shapes.csv S:/tmp/ph_$tag || exit 1
pass /tmp/kf_$tag $out/bench_fp64_under_kernel_trace.log --kernel-trace --stats --output-format csv -d /tmp/kf_$tag -o kt -- $B --mode 1 --steps 2 --warmup 1
cp $(need /tmp/kf_$tag "*kernel_stats.csv" $out/bench_fp64_under_kernel_trace.log) $out/kernel_stats_fp64.csv || exit 1
grep '^{"metric"' $out/bench_fp64_under_kernel_trace.log | tail -1 > $out/bench_fp64.json
pass /tmp/pq_$tag $out/bench_fp64_under_sq.log --pmc $SQ --kernel-trace --output-format csv -d /tmp/pq_$tag -o ps -- $B --mode 1 --steps 1 --warmup 0
summarise $out/pmc_summary_fp64.csv S:/tmp/pq_$tag || exit 1
# what the profile was taken from: bench.py quotes its counters only for a library built from the same kernel sources
python3 - $out $tag <<'PY'
import json, sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from mcevidence_amd import _capi
json.dump(dict(tag=sys.argv[2], source_hash=bench.source_hash(), library_source_hash=_capi.source_hash(), taken=time.strftime("%Y-%m-%d %H:%M:%S"),
               commands=["rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extras",
                         "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras (separate passes)",
                         "rocprofv3 --kernel-trace --stats -- python3 tools/run_configs.py C2 | C4 | C5",
                         "rocprofv3 --pmc SQ_* --kernel-trace -- python3 tools/run_configs.py C2 | C4 | C5 (pmc_summary_<C>.csv: per dispatch = sum / dispatches)",
                         "rocprofv3 --kernel-trace --stats | --pmc SQ_* --kernel-trace -- python3 tools/shape_times.py 100000,64,9 100000,100,9 100000,127,9 100000,128,9 100000,256,9 (kernel_stats_shapes.csv, pmc_summary_shapes.csv, shape_times.txt)",
                         "rocprofv3 --kernel-trace --stats -- python3 bench.py --mode 1 --steps 2 --warmup 1 --cpu-sample 0 --no-extras"]),
          open(os.path.join(sys.argv[1], "meta.json"), "w"), indent=1)
PY
head -6 $out/kernel_stats.csv | cut -c1-200
