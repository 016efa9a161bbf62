#!/bin/bash
# GPU box: tools/valu_issue_clock (issue cycles per vector-instruction class) and, in separate --pmc passes, how the SQ counters
# count each class (so that a kernel's SQ_INSTS_VALU_* counters can be priced with the measured cycles).
# usage: tools/valu_issue.sh   -> gpurun_out/valu_issue/{clock.jsonl, counters.csv}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/valu_issue; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
[ -x $R/tools/valu_issue_clock ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $R/tools/valu_issue_clock.hip -o $R/tools/valu_issue_clock || exit 1
$R/tools/valu_issue_clock 40000 > $out/clock.jsonl 2> $out/clock.err || { echo "valu_issue_clock failed"; tail -5 $out/clock.err; exit 1; }
i=0
for set in "SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_FMA_F16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/vi_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/vi_$i -o p -- $R/tools/valu_issue_clock 2000 > /tmp/vi_$i.log 2>&1
  [ -n "$(find /tmp/vi_$i -name '*counter_collection.csv' | head -1)" ] || { echo "valu_issue.sh: pass $i produced no counters:"; tail -5 /tmp/vi_$i.log; exit 1; }
done
python3 - $out/counters.csv <<'PY'
import csv, glob, sys, collections
# every kernel runs twice (a warm-up of iters / 8, then the timed launch): keep the LAST dispatch of each kernel name per pass
rows = collections.OrderedDict()
for f in sorted(glob.glob("/tmp/vi_*/**/*counter_collection.csv", recursive=True)):
    last = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        key = (k, r["Counter_Name"])
        d = int(r["Dispatch_Id"])
        if key not in last or d >= last[key][0]:
            last[key] = (d, (last[key][1] if key in last and last[key][0] == d else 0.0) + float(r["Counter_Value"]))
    for (k, c), (d, v) in last.items():
        rows[(k, c)] = v
with open(sys.argv[1], "w") as fh:
    fh.write("kernel,counter,value_last_dispatch\n")
    for (k, c), v in rows.items():
        fh.write('"%s",%s,%.6g\n' % (k, c, v))
PY
wc -l $out/clock.jsonl $out/counters.csv
