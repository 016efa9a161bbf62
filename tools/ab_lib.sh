#!/bin/bash
# GPU box: same-box A/B of two builds of the library (boxes of the pool differ by several per cent): the command is run
# with MCE_LIB pointing at each build in turn, twice, interleaved.   usage: tools/ab_lib.sh "<python command>" libA.so libB.so ...
cmd=$1; shift
for i in 1 2; do
  for lib in "$@"; do
    echo "== $lib (run $i)"
    MCE_LIB=$PWD/$lib bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
  done
done
