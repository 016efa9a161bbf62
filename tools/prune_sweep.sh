#!/bin/bash
# pruned vs exhaustive over (N, D, K): python tools/prune_bench.py N D K --full
for cfg in "300000 6 10" "1000000 2 10" "1000000 3 10" "1000000 4 10" "2000000 6 10" "4000000 6 10" "1000000 8 10" "4000000 8 10" "4000000 10 10" "10000000 3 10"; do
  python tools/prune_bench.py $cfg --full 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
p, x = r['pruned'], r.get('exhaustive', {})
print('N=%-9d D=%-2d K=%-2d  pruned %8.1f ms (kernel %8.1f, tiles %.4f)   exhaustive %8.1f ms' % (r['N'], r['D'], r['K'], p['ms'], p['search_kernel_ms'], p['tile_fraction'], x.get('ms', float('nan'))))"
done
