#!/bin/bash
# round 5, lab call 28: the 16x16 probe in one round of its workgroup (56 sample blocks instead of 70 at 1280x960) -- new against the commit before (ab/libaof_head.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab28
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_pruned.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {a: round(b * 1e3, 1) for a, b in d.get("kernels_ms", {}).items() if isinstance(b, float)}
print(f"{sys.argv[2]:26s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  frac_step {d['roofline'].get('frac_step')}  {k}  {d.get('adaptive_search')}")
PY
}
for round in 1 2 3; do
  for lib in head new; do
    if [ $lib = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    b c5_${lib}_$round --workload c5
    b c5h_${lib}_$round --workload c5h
    b c5_n6_${lib}_$round --workload c5 --noise 6
    b c5_n16_${lib}_$round --workload c5 --noise 16
  done
done
echo done
