#!/bin/bash
# round 5, lab call 45: 16x16 adaptive search -- four- and eight-row bounds in step A where two rows do not separate the candidates (verdicts 3, 4) -- new against the commit before
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab45
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_pruned.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -40 $O/tests.log; exit 1; }
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
    for nz in 0 4 6 8 12 16 24 40; do b c5_n${nz}_${n}_$round --workload c5 --pairs 256 --noise $nz; done
    b c5_real_${n}_$round --workload c5 --pairs 256 --input realistic
    b c5h_n8_${n}_$round --workload c5h --pairs 256 --noise 8
    b c5h_real_${n}_$round --workload c5h --pairs 256 --input realistic
  done
done
echo done
