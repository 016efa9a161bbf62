#!/bin/bash
# GPU box, round 6: loose chunk hand-over of the panel kernel -- same-box A/B (microbench), statistics, and the symmetric-sweep tests
# on a library built with it
cd $GRAFT_REPO_ROOT
{
echo "== A/B: sweep ms (last repetition), each binary twice, interleaved"
for i in 1 2; do for b in base loose; do echo -n "$b run $i: "; timeout 200 tools/_r06/symx_$b 1000000 3 | grep sweep | tail -1 | sed 's/.*prepass/prepass/'; done; done
echo "== statistics, loose"; timeout 300 tools/_r06/symx_loose_st 1000000 2 | tail -7
echo "== statistics, barrier"; timeout 300 tools/_r06/symx_st 1000000 2 | tail -7
} > gpurun_out/r06_loose_ab.txt 2>&1
cat gpurun_out/r06_loose_ab.txt | cut -c1-250
MCE_LIB=$PWD/tools/_r06/lib_loose.so timeout 900 python -m pytest tests/test_gpu_symmetric.py -x -q 2>&1 | tail -5 > gpurun_out/r06_loose_tests.log
cat gpurun_out/r06_loose_tests.log
python -m pytest tests/test_gpu_parity.py -x -q -k "deep or mfma_error_model" 2>&1 | tail -4 > gpurun_out/r06_deep_tests2.log
cat gpurun_out/r06_deep_tests2.log
