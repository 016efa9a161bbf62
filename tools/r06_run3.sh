#!/bin/bash
# GPU box, round 6: the deep filter's first contact + the panel kernel's barrier-wait statistics
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "deep or beyond or over_dims or mfma_error_model" 2>&1 | tail -15 > gpurun_out/r06_deep_tests.log
tail -3 gpurun_out/r06_deep_tests.log
{
for spec in "100000,64,9" "100000,100,9" "100000,127,9" "100000,63,9" "300000,64,9" "100000,64,4,cross=100000"; do
  MCE_DEEP=1 python tools/shape_times.py "$spec"
  MCE_DEEP=0 python tools/shape_times.py "$spec"
done
} > gpurun_out/r06_deep_times.txt 2>&1
cat gpurun_out/r06_deep_times.txt | cut -c1-220
{
echo "== stats build"; timeout 300 tools/_r06/symx_st 1000000 2 | tail -8
echo "== stats build, gates never pass"; timeout 300 tools/_r06/symx_st_abl1 1000000 2 | tail -8
echo "== plain"; timeout 300 tools/_r06/symx_base 1000000 3 | tail -3
} > gpurun_out/r06_panel_stats.txt 2>&1
cat gpurun_out/r06_panel_stats.txt | cut -c1-260
