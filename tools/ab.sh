#!/bin/bash
# GPU box: A/B of microbench binaries on ONE box (boxes of the pool differ by several per cent): each binary twice, interleaved.
# usage: tools/ab.sh "<args>" bin1 bin2 ...   -> one line per binary: the sweep times of the last repetition of each run
args=$1; shift
declare -A res
for i in 1 2; do
  for b in "$@"; do
    t=$(timeout 180 tools/$b $args | grep sweep | tail -1 | sed 's/.*sweep \([0-9.]*\) ms.*/\1/')
    res[$b]="${res[$b]} $t"
  done
done
for b in "$@"; do printf "%-36s %s\n" $b "${res[$b]}"; done
