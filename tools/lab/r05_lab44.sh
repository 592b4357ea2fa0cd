#!/bin/bash
# round 5, lab call 44: 16x16 adaptive search with a three-way verdict per pair (exhaustive / two-row bounds / ONE-row bounds in step A) -- new against the commit before
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab44
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_pruned.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {a: round(b * 1e3, 1) for a, b in d.get("kernels_ms", {}).items() if isinstance(b, float)}
print(f"{sys.argv[2]:26s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  frac_step {d['roofline'].get('frac_step')}  {k}")
PY
}
for round in 1 2; do
  for n in head new; do
    if [ $n = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    b c5_${n}_$round --workload c5 --pairs 256
    b c5_n1_${n}_$round --workload c5 --pairs 256 --noise 1
    b c5_n2_${n}_$round --workload c5 --pairs 256 --noise 2
    b c5_n4_${n}_$round --workload c5 --pairs 256 --noise 4
    b c5_n8_${n}_$round --workload c5 --pairs 256 --noise 8
    b c5_real_${n}_$round --workload c5 --pairs 256 --input realistic
    b c5h_${n}_$round --workload c5h --pairs 256
    b c5h_n2_${n}_$round --workload c5h --pairs 256 --noise 2
  done
done
echo done
