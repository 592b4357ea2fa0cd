#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for lib in c1024 c512; do
for args in "--workload c3 --streams 1" "--workload c3 --streams 2 --reduce fused" "--workload c3 --streams 3 --reduce fused"; do
  AOF_LIB=$R/ab/$lib.so timeout -k 10 200 python3 bench.py --cpu-seconds 0 --steps 100 $args 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', '$args'.ljust(50), round(j['value']), 'step', round(j['ms_per_step']*1e3,1), 'frac_step', j['roofline']['frac_step'], j['kernels_ms'].get('coarse_fused'), j['parity']['oracle_pairs_bit_exact'])"
done
done
done
