#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for lib in base schedbar; do
  for mode in separate fused; do
    AOF_LIB=$R/ab/$lib.so timeout -k 10 200 python3 bench.py --cpu-seconds 0 --steps 200 --streams 1 --reduce $mode 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $mode', 'step', round(j['ms_per_step']*1e3,1), 'K2', round(j['roofline']['kernel_ms']*1e3,2), j['parity']['oracle_pairs_bit_exact'])"
  done
done
done
