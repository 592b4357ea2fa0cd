#!/bin/bash
# round 5, lab call 13: what the predictor costs C3's level-0 search -- shifts restricted so that windows are (mis)aligned
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab13
mkdir -p $O
cd $R
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {a: b for a, b in d.get("kernels_ms", {}).items() if a not in ("note",)}
print(f"{sys.argv[2]:30s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  {k}")
PY
}
for round in 1 2; do
  for ms in 0 1 2 4 9; do
    b c3_ms${ms}_$round --workload c3 --max-shift $ms
    b c3_ms${ms}_b0_$round --workload c3 --max-shift $ms --brightness 0
    b c3_ms${ms}_exh_$round --workload c3 --max-shift $ms --search exhaustive
  done
  b c3n_$round --workload c3n
  b c2_$round --workload c2
  b c2_exh_$round --workload c2 --search exhaustive
done
echo done
