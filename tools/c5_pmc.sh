#!/bin/bash
# GPU box: instruction mix and wave-cycle counters of the pruned walk at C5 (separate --pmc passes, kernel trace only).
# usage: tools/c5_pmc.sh [c5_time.py arguments]   ->  gpurun_out/c5_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out; mkdir -p $out
rm -rf /tmp/c5pmc_*                       # (stale counter files of an earlier run would be summed in)
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_FMA_F16 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/c5pmc_$i -o p -- python3 $R/tools/c5_time.py --reps 1 "$@" > /tmp/c5pmc_$i.log 2>&1
  [ -n "$(find /tmp/c5pmc_$i -name '*counter_collection.csv' | head -1)" ] || { echo "c5_pmc.sh: pass $i produced no counters:"; tail -5 /tmp/c5pmc_$i.log; exit 1; }
done
python3 - <<P > $out/c5_pmc.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/c5pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "knn_f16_kernel" not in k: continue
        tot[k[:80]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in tot.items():
    print(k)
    for c in sorted(d): print("  %-28s %.6g" % (c, d[c] / 2))      # two timed calls per run (warm-up + 1)
P
cat $out/c5_pmc.txt
