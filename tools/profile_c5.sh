#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of the pruned C5 search (tools/run_configs.py C5) -> gpurun_out/prof_c5/kernel_stats.csv
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_c5
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_c5 -o kt -- python3 $R/tools/run_configs.py C5 > $out/run_configs_under_kernel_trace.log 2>&1
cp $(find /tmp/kt_c5 -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
head -20 $out/kernel_stats.csv
