#!/bin/bash
# round 5, lab call 38: the same behind a wave-uniform flag set after the block's first row (every needing lane's best SAD is 0) -- new against the commit before
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab38
mkdir -p $O
cd $R
timeout -k 10 700 python3 -m pytest tests/test_gpu_pruned.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {a: round(b * 1e3, 1) for a, b in d.get("kernels_ms", {}).items() if isinstance(b, float)}
print(f"{sys.argv[2]:26s} value {d['value']/1e6:9.4f} M  step {d['ms_per_step']*1e3:7.1f} us  frac_step {d['roofline'].get('frac_step')}  {k}")
PY
}
for round in 1 2 3; do
  for lib in head new; do
    if [ $lib = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    b c2_${lib}_$round --workload c2
    b c2_n4_${lib}_$round --workload c2 --noise 4
    b c2_n8_${lib}_$round --workload c2 --noise 8
    b c3_${lib}_$round --workload c3
    b c2h_${lib}_$round --workload c2h
    b p128_${lib}_$round --pairs 128 --steps 2000
  done
done
echo done
