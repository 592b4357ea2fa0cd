#!/bin/bash
# round 5, lab call 36: the refinement's early end also in the exhaustive lane-per-block kernels (ring from memory) and the per-call kernel -- new against the commit before
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab36
mkdir -p $O
cd $R
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {a: round(b * 1e3, 1) for a, b in d.get("kernels_ms", {}).items() if isinstance(b, float)}
print(f"{sys.argv[2]:26s} value {d['value']/1e6:9.4f} M  step {d['ms_per_step']*1e3:7.1f} us  frac_step {d['roofline'].get('frac_step')}  {k}")
PY
}
for round in 1 2; do
  for lib in head new; do
    if [ $lib = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    b c1b_${lib}_$round --workload c1b
    b c1b_n4_${lib}_$round --workload c1b --noise 4
    b c2h_x_${lib}_$round --workload c2h --search exhaustive
    b c2h_n16_${lib}_$round --workload c2h --noise 16
    b c2h_n40_${lib}_$round --workload c2h --noise 40
    b c1_${lib}_$round --workload c1 --pairs 256 --steps 20
  done
done
echo done
