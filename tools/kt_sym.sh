#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of `python bench.py` with the symmetric sweep forced on / off.
# usage: tools/kt_sym.sh <tag> [MCE_SYM value]   -> gpurun_out/kt_<tag>/kernel_stats.csv
tag=${1:-sym}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/kt_$tag
mkdir -p $out
export MCE_SYM=${2:-2}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $out/bench.log 2>&1
cp $(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.2:
        print("%-90s calls %5s  avg %10.3f ms  total %6.2f %%" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e6, float(r['Percentage'])))
PY
tail -c 400 $out/bench.log
python3 - $(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'knn_f16_kernel' in r['Kernel_Name'] or 'sym_merge' in r['Kernel_Name']]
for r in rows[-8:]:
    print("%-70s %10.3f ms  grid %s" % (r['Kernel_Name'][:70], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r.get('Grid_Size_X', r.get('Grid_Size',''))))
PY
