#!/bin/bash
# round 5, lab call 25: the two changes of lab call 24 apart -- head, first (no tests in a block's first row), both (+ rows the vote has summed are dropped unsummed)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab25
mkdir -p $O
cd $R
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels_ms", {})
print(f"{sys.argv[2]:30s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  search {k.get('search', 0)*1e3:7.1f}")
PY
}
for round in 1 2 3; do
  for lib in head first both; do
    export AOF_LIB=$R/ab/libaof_$lib.so
    b c2_${lib}_$round --workload c2
    b c2_n4_${lib}_$round --workload c2 --noise 4
    b c3_${lib}_$round --workload c3
    b c2h_${lib}_$round --workload c2h
    b p128_${lib}_$round --pairs 128 --steps 2000
  done
done
echo done
