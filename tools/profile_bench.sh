#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + HBM counters for `python bench.py` (round-tagged).
# usage: tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/...
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $out/bench_under_kernel_trace.log 2>&1
cp $(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
# HBM traffic counters: separate passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$tag -o pf -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $out/bench_under_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$tag -o pw -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $out/bench_under_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/ps_$tag -o ps -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $out/bench_under_sq.log 2>&1
python3 - $tag $out <<'PY'
import csv, sys, glob, collections
tag, out = sys.argv[1], sys.argv[2]
rows=[]
for name, d in (("FETCH_SIZE","/tmp/pf_"+tag),("WRITE_SIZE","/tmp/pw_"+tag),("SQ","/tmp/ps_"+tag)):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int)
        seen=set()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][:72]
            agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
            seen.add((k, r['Dispatch_Id']))
        for k in agg:
            nd=len([1 for kk,_ in seen if kk==k])
            for c,v in agg[k].items(): rows.append((k,c,v,nd))
with open(out+"/pmc_summary.csv","w") as fh:
    fh.write("kernel,counter,sum_over_dispatches,dispatches\n")
    for r in rows: fh.write("%s,%s,%.6g,%d\n"%r)
print(open(out+"/pmc_summary.csv").read())
PY
# the line the kernel-trace process printed (its HIP-event bracket over the same launches the trace averaged)
grep '^{"metric"' $out/bench_under_kernel_trace.log | tail -1 > $out/bench.json
# per-config kernel traces (one process each, so that a kernel shared by two configs -- knn_f16_kernel<1,4,..,0> serves C2 and
# C4 -- gets one row per config) and the fp64 sweep
for c in C2 C4 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc_${tag}_$c -o kt -- python3 $R/tools/run_configs.py $c > $out/run_configs_$c.log 2>&1
  cp $(find /tmp/kc_${tag}_$c -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$c.csv 2>/dev/null
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kf_$tag -o kt -- python3 $R/bench.py --mode 1 --steps 2 --warmup 1 --cpu-sample 0 --no-extras > $out/bench_fp64_under_kernel_trace.log 2>&1
cp $(find /tmp/kf_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats_fp64.csv 2>/dev/null
grep '^{"metric"' $out/bench_fp64_under_kernel_trace.log | tail -1 > $out/bench_fp64.json
# what the profile was taken from: bench.py quotes its counters only for a library built from the same kernel sources
python3 - $out $tag <<'PY'
import json, sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from mcevidence_amd import _capi
json.dump(dict(tag=sys.argv[2], source_hash=bench.source_hash(), library_source_hash=_capi.source_hash(), taken=time.strftime("%Y-%m-%d %H:%M:%S"),
               commands=["rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extras",
                         "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-extras (separate passes)",
                         "rocprofv3 --kernel-trace --stats -- python3 tools/run_configs.py C2 | C4 | C5", "rocprofv3 --kernel-trace --stats -- python3 bench.py --mode 1 --steps 2 --warmup 1 --cpu-sample 0 --no-extras"]),
          open(os.path.join(sys.argv[1], "meta.json"), "w"), indent=1)
PY
head -12 $out/kernel_stats.csv
