#!/bin/bash
# round 5, lab call 24: the first row of a block without its tests, and rows the vote has summed dropped without summing them again -- new against the commit before
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab24
mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { echo "gpu tests failed"; tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
k = {a: b for a, b in d.get("kernels_ms", {}).items() if a not in ("note",)}
print(f"{sys.argv[2]:30s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  k2 {r.get('kernel_ms')*1e3:7.1f}  {k}")
PY
}
for round in 1 2 3; do
  for lib in head new; do
    if [ $lib = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    b c2_${lib}_$round --workload c2
    b c2_n4_${lib}_$round --workload c2 --noise 4
    b c2_n8_${lib}_$round --workload c2 --noise 8
    b c3_${lib}_$round --workload c3
    b c2_s2_${lib}_$round --workload c2 --streams 2
    b p128_${lib}_$round --pairs 128 --steps 2000
    b p256_${lib}_$round --pairs 256 --steps 1000
  done
done
echo done
