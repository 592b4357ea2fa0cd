#!/bin/bash
# round 5, lab call 43: 16x16 pruned search with ONE-row lower bounds in step A (kBoundRows 1, probe ratio 12) -- ab/libaof_b1.so against head
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab43
mkdir -p $O
cd $R
AOF_LIB=$R/ab/libaof_b1.so timeout -k 10 600 python3 -m pytest tests/test_gpu_pruned.py tests/test_gpu_parity.py -m gpu -x -q -k "16 or c5 or tile" > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
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
  for n in head b1; do
    export AOF_LIB=$R/ab/libaof_$n.so
    b c5_${n}_$round --workload c5 --pairs 256
    b c5_n2_${n}_$round --workload c5 --pairs 256 --noise 2
    b c5_n4_${n}_$round --workload c5 --pairs 256 --noise 4
    b c5_n8_${n}_$round --workload c5 --pairs 256 --noise 8
    b c5_n16_${n}_$round --workload c5 --pairs 256 --noise 16
    b c5h_${n}_$round --workload c5h --pairs 256
    b c5_pruned_n16_${n}_$round --workload c5 --pairs 256 --noise 16 --search pruned
  done
done
echo done
