#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 300 python -m pytest tests/test_ingest.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for lib in rows128 widen; do
    AOF_LIB=$R/ab/$lib.so timeout -k 10 200 python3 bench.py --workload ingest --pairs 1024 --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'step', round(j['ms_per_step']*1e3,1), 'kernel', round(j['roofline']['kernel_ms']*1e3,2), 'frac', j['roofline']['frac'], j['parity'])"
done
done
