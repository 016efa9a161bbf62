#!/bin/bash
# GPU box, round 5: the one-k-step exhaustive sweep at 8 waves x FOUR query tiles (GEOM 5) against the shipped 8 x 2, with and
# without candidates (a1: the gate never passes), each twice, interleaved.
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for b in g0 g5np1 g0_a1 g5np1_a1; do
    echo -n "$b run $i: "; timeout 120 tools/knn_bench_k1_$b 1000000 1 3 | grep "ms " | tail -1 | sed "s/.*grid=[0-9]*: //"
  done
done
timeout 60 tools/knn_bench_k1_g5np1 1000000 1 1 | grep checksum
