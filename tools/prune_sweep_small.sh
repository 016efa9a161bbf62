#!/bin/bash
for cfg in "100000 6 4" "200000 6 4" "100000 3 4" "200000 3 10" "500000 6 10" "2000000 8 10" "3000000 8 10" "1000000 7 10"; do
  python tools/prune_bench.py $cfg --full 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
p, x = r['pruned'], r.get('exhaustive', {})
print('N=%-9d D=%-2d K=%-2d  pruned %8.2f ms (kernel %8.2f, tiles %.4f)   exhaustive %8.2f ms' % (r['N'], r['D'], r['K'], p['ms'], p['search_kernel_ms'], p['tile_fraction'], x.get('ms', float('nan'))))"
done
