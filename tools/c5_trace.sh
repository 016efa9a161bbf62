#!/bin/bash
# GPU box: per-kernel time of one C5 call (rocprofv3 kernel trace) for the library MCE_LIB names.  usage: tools/c5_trace.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5tr_$tag -o t -- python3 $R/tools/c5_time.py --reps 2 "$@" > /tmp/c5tr_$tag.log 2>&1
python3 - <<P
import csv, glob
f = glob.glob("/tmp/c5tr_$tag/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== $tag: all kernels %.2f ms over 3 calls" % (tot / 1e6))
for r in rows[:14]:
    print("  %-70s calls %4s  avg %9.3f ms  total %8.2f ms  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
P
