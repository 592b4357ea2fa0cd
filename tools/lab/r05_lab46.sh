#!/bin/bash
# round 5, lab call 46: the deeper-bound verdicts with the cheaper probe, short sweep -- new against the commit before
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab46
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_pruned.py -m gpu -x -q -k "16" > $O/tests.log 2>&1 || { echo "tests failed"; tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:26s} {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us")
PY
}
for round in 1 2; do
  for n in head new; do
    if [ $n = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    for nz in 0 4 8 12 16 40; do b c5_n${nz}_${n}_$round --workload c5 --pairs 256 --noise $nz; done
    b c5_real_${n}_$round --workload c5 --pairs 256 --input realistic
    b c5h_${n}_$round --workload c5h --pairs 256
  done
done
echo done
