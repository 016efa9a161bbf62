#!/bin/bash
# GPU box: what bounds the exhaustive fp16-filter sweep at ONE k-step (d <= 15: C4, C2, query shards) -- ablation builds of
# tools/knn_f16_bench.hip (D = 15, K = 4, 1 M x 1 M, seeded sweep) on ONE box, each twice, interleaved, then the SQ counters of
# the real kernel.  Build first (build container):
#   for a in 0 1 2 5 6 7 8; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-sched-strategy=max-ilp -DDIM=15 -DBKCAP=4 -DBKSEL=4 \
#       -DMCE_ABLATE=$a tools/knn_f16_bench.hip -o tools/knn_bench_k1_a$a; done
# usage: tools/kst1_ablation.sh [n = 1000000]  -> gpurun_out/kst1_ablation.txt
n=${1:-1000000}
out=$GRAFT_REPO_ROOT/gpurun_out/kst1_ablation.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
: > $out
for i in 1 2; do
  for a in 0 1 7 8 2 5 6; do
    echo -n "ablate=$a run=$i: " >> $out
    timeout 120 tools/knn_bench_k1_a$a $n 1 3 | grep "ms " | tail -1 | sed 's/.*grid=[0-9]*: //' >> $out
  done
done
echo "---- SQ counters of the real kernel (ablate=0), three passes" >> $out
tools/prof_pmc.sh k1 $GRAFT_REPO_ROOT/tools/knn_bench_k1_a0 $n 1 2 >> $out 2>&1
cat $out
