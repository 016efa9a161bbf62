#!/bin/bash
# GPU box: the all-pairs-once partition at C3, chains per block (MCE_PAIRS_ONCE_SPLIT) x panel length scan -> stdout
cd "$(dirname "$0")/.."
for cfg in ${APO_CFGS:-2,0,0 4,0,0 8,0,0}; do
  IFS=, read -r a1 a2 a3 <<< "$cfg"; set -- $a1 $a2 $a3
  if [ "$2" != "0" ]; then export MCE_PAIRS_ONCE_SPLIT=$2; else unset MCE_PAIRS_ONCE_SPLIT; fi
  if [ "$3" != "0" ]; then export MCE_PAIRS_ONCE_PANEL=$3; else unset MCE_PAIRS_ONCE_PANEL; fi
  python - "$1" <<'PY'
import sys, os, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np, bench, pairs_once_emulate as pe
W = int(sys.argv[1])
cfg = bench.prep_config("C3")
r = pe.emulate(cfg["X"], cfg["weight"], cfg["fs"], cfg["kmax"], W, reps=2)
print("W=%d split=%s panel=%s step %.2f = prepare %.2f + sweep %.2f (kernel %.2f) + export %.2f + finish %.2f + exchange %.2f; sent %d | %s" % (W, os.environ.get("MCE_PAIRS_ONCE_SPLIT"), os.environ.get("MCE_PAIRS_ONCE_PANEL"),
      r["predicted_step_ms"], max(r["prepare_ms"]), max(r["sweep_ms"]), max(r["sweep_kernel_ms"]), max(r["export_ms"]), max(r["finish_ms"]), max(r["exchange_ms_priced"]), sum(r["candidates_sent"]), r["kernel"][60:]), flush=True)
PY
done
