#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for lib in base f6 f7 f8 f11 f12 f13 f14 f15; do
  [ -f $R/ab/$lib.so ] || continue
    AOF_LIB=$R/ab/$lib.so timeout -k 10 200 python3 bench.py --cpu-seconds 0 --steps 200 --streams 1 --reduce separate 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'step', round(j['ms_per_step']*1e3,1), 'K2', round(j['roofline']['kernel_ms']*1e3,2), j['parity']['oracle_pairs_bit_exact'], j.get('exact_pruned_search',{}).get('kernel_ms'))"
done
done
