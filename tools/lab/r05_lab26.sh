#!/bin/bash
# round 5, lab call 26: the exhaustive 8x8 search as a column walk (k_search_lane8_colsx, AOF_LAB_COLSX=1) against the chunk-walking flat kernel (=0)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab26
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_pruned.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels_ms", {})
print(f"{sys.argv[2]:30s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  search {k.get('search', 0)*1e3:7.1f}")
PY
}
for round in 1 2; do
  for x in 0 1; do
    export AOF_LAB_COLSX=$x
    b c2x_colsx${x}_$round --workload c2 --search exhaustive
    b c3x_colsx${x}_$round --workload c3 --search exhaustive
    b c3n16_colsx${x}_$round --workload c3 --noise 16
    b c3x_ms0_colsx${x}_$round --workload c3 --search exhaustive --max-shift 0
    b c2n16_colsx${x}_$round --workload c2 --noise 16
  done
done
echo done
