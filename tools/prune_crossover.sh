#!/bin/bash
# GPU box: pruned walk vs the automatic exhaustive / symmetric choice around the automatic mode's thresholds
# (capi_common.hpp: kPruneAutoMinRows), auto evidence, resident data.  usage: tools/prune_crossover.sh -> gpurun_out/prune_crossover.txt
out=$GRAFT_REPO_ROOT/gpurun_out/prune_crossover.txt; : > $out
for cfg in "30000 2 4" "50000 2 4" "30000 3 4" "50000 3 4" "100000 3 4" "50000 4 4" "100000 4 4" "150000 4 4" "50000 5 4" "100000 5 4" "150000 5 4" "200000 5 4" \
           "50000 6 4" "100000 6 4" "150000 6 4" "200000 6 4" "300000 6 4" "100000 6 9" "200000 6 9" "300000 6 9" "100000 7 4" "200000 7 4" "300000 7 4" "500000 7 4" \
           "800000 7 4" "300000 7 9" "500000 7 9" "300000 8 4" "500000 8 4" "1000000 8 4" "2000000 8 4" "500000 8 9" "1000000 8 9"; do
  python tools/prune_bench.py $cfg --full 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
p, x = r['pruned'], r.get('exhaustive', {})
print('N=%-9d D=%-2d K=%-2d  pruned %8.2f ms (kernel %8.2f, tiles %.4f)   other %8.2f ms  ratio %.2f  %s' % (r['N'], r['D'], r['K'], p['ms'], p['search_kernel_ms'], p['tile_fraction'], x.get('ms', float('nan')), p['ms'] / x.get('ms', float('nan')), 'symmetric' if 'symmetric' in x.get('kernel', '') else 'exhaustive'))" >> $out
done
cat $out
