#!/bin/bash
# GPU box, round 5: (a) the one-k-step exhaustive sweep at 12 waves per workgroup (three per SIMD) against the shipped eight,
# (b) static wave priority in the panel kernel.  Each binary twice, interleaved, on ONE box.
cd $GRAFT_REPO_ROOT
echo "---- KST = 1 sweep, D = 15, K = 4, 1 M x 1 M (ms of the last repetition)"
for i in 1 2; do
  for b in g0 g4np1 g4np2 g4np1_a1; do
    echo -n "$b run $i: "; timeout 120 tools/knn_bench_k1_$b 1000000 1 3 | grep "ms " | tail -1 | sed 's/.*grid=[0-9]*: //'
  done
done
timeout 60 tools/knn_bench_k1_g0 1000000 1 1 | grep checksum; timeout 60 tools/knn_bench_k1_g4np1 1000000 1 1 | grep checksum
echo "---- panel kernel, static priority (C3 shape: 1 M x 27, K = 9)"
tools/ab.sh "1000000 3" knn_bench_sym_p0 knn_bench_sym_p1 knn_bench_sym_p2
timeout 180 tools/knn_bench_sym_st0 1000000 2 | tail -4
