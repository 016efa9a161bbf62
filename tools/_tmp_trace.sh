#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr1 -o tr -- python3 $GRAFT_REPO_ROOT/tools/shape_times.py $1 > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/tr1/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
