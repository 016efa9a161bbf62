#!/bin/bash
# usage: tools/prof_pmc.sh <tag> <binary> <args...>   (GPU box) -> gpurun_out/pmc_<tag>_passN.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FLOPS_FP64 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$i -o p -- "$@" > /tmp/pmc_${tag}_$i.log 2>&1
  tail -2 /tmp/pmc_${tag}_$i.log | grep -i -E "error|fail" ; f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f=sys.argv[1]
agg=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if 'knn_' in r.get('Kernel_Name',''):
        agg[r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items(): print("%-34s %.4g"%(k,v))
PY
done
