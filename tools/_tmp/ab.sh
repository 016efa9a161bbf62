#!/bin/bash
# A/B of two library builds on one box
cd $GRAFT_REPO_ROOT
for v in default ilp default ilp; do
  cp tools/_tmp/lib_$v.so mcevidence_amd/libmcevidence_hip.so
  echo "=== $v"
  python tools/run_configs.py C2 C3 C4 C5 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(d['config'], round(d['kernel_ms'],2), 'ms', d['kernel'][:40])
"
  python bench.py --mode 1 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f64 sweep', d['roofline']['kernel_ms'])"
  python tools/prune_bench.py 1000000 45 9 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1M x 45', {k:(v['search_kernel_ms']) for k,v in d.items() if isinstance(v,dict)})"
  python tools/prune_bench.py 1000000 27 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1M x 27 K=20', {k:(v['search_kernel_ms']) for k,v in d.items() if isinstance(v,dict)})"
done
