#!/bin/bash
# GPU box: C5 (10 M x 6, kmax 10) on ONE box with the round-4 k-d tree (one sort per level, MCE_KD_TREE=interleaved) and the
# round-5 one (one sort per dimension), interleaved, then the kernel trace of the default.  usage: tools/c5_ab.sh [c5_time.py args]
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for t in interleaved perdim; do
    echo "== tree=$t run $i"
    MCE_KD_TREE=$t python3 tools/c5_time.py --reps 3 "$@" 2>&1 | grep -v amdgpu.ids
  done
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_c5ab
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_c5ab -o kt -- python3 $GRAFT_REPO_ROOT/tools/c5_time.py --reps 3 "$@" > /tmp/kt_c5ab.log 2>&1
f=$(find /tmp/kt_c5ab -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel trace"; tail -5 /tmp/kt_c5ab.log; exit 1; }
cp $f $GRAFT_REPO_ROOT/gpurun_out/c5ab_kernel_stats.csv
python3 - $f <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['TotalDurationNs']) > 2e5:
        print("%-72s calls %4s avg %9.1f us total %9.1f us" % (r['Name'].replace('void ', '')[:72], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
P
