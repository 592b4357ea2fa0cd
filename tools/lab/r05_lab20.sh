#!/bin/bash
# round 5, lab call 20: 16x16 kernel -- run-time divisions by the grid width through a magic number, lower bounds as packed minima -- new against the commit before
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab20
mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { echo "gpu tests failed"; tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(f"{sys.argv[2]:30s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  frac_step {r.get('frac_step')}")
PY
}
for round in 1 2; do
  for lib in head new; do
    if [ $lib = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    b c5_${lib}_$round --workload c5 --pairs 256
    b c5_pruned_${lib}_$round --workload c5 --pairs 256 --search pruned
    b c5_exh_${lib}_$round --workload c5 --pairs 256 --search exhaustive
    b c5_n8_${lib}_$round --workload c5 --pairs 256 --noise 8
    b c5_n16_${lib}_$round --workload c5 --pairs 256 --noise 16
    b c5h_${lib}_$round --workload c5h --pairs 256
    b c5h_exh_${lib}_$round --workload c5h --pairs 256 --search exhaustive
    b c5p_${lib}_$round --workload c5p --pairs 256
  done
done
echo done
