#!/bin/bash
# round 5, lab call 41: k_coarse -- the blocks that fill whole rounds take one lane each for all nine dy rows, the rest three dy rows per lane -- new against the commit before
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab41
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sequence.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {a: round(b * 1e3, 1) for a, b in d.get("kernels_ms", {}).items() if isinstance(b, float)}
print(f"{sys.argv[2]:26s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  frac_step {d['roofline'].get('frac_step')}  {k}")
PY
}
for round in 1 2 3; do
  for lib in head new; do
    if [ $lib = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    b c3_${lib}_$round --workload c3
    b c3_n16_${lib}_$round --workload c3 --noise 16
    b c3_s2_${lib}_$round --workload c3 --streams 2
    b c3_p256_${lib}_$round --workload c3 --pairs 256 --steps 400
  done
done
echo done
