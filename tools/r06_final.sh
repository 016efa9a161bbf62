#!/bin/bash
# GPU box: everything profiles/r06_final/ holds, in one call (the kernel sources must not change afterwards: meta.json names their hash)
cd "$(dirname "$0")/.."
tag=${1:-r06_final}
tools/refresh_profiles.sh $tag || exit 1
out=gpurun_out/prof_$tag
python tools/predict_scaling.py C3 C4 C5 > $out/predict_scaling.log 2>&1 && cp gpurun_out/predicted_scaling.json $out/ || { echo "predict_scaling failed"; tail -5 $out/predict_scaling.log; }
python tools/pairs_once_emulate.py > $out/pairs_once_emulate.log 2>&1 && cp gpurun_out/pairs_once_emulated.json $out/ || { echo "pairs_once_emulate failed"; tail -5 $out/pairs_once_emulate.log; }
ls -la $out | head -40
