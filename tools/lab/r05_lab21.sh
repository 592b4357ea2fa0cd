#!/bin/bash
# round 5, lab call 21: does the column walk's code size (81 specialised dy rows, ~80 KB) cost it? -- c2 with every shift alike (one start row hot) against BASELINE's shifts
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab21
mkdir -p $O
cd $R
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(f"{sys.argv[2]:30s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  k2 {r.get('kernel_ms')*1e3:7.1f}")
PY
}
for round in 1 2; do
  for ms in 0 1 2 4; do b c2_ms${ms}_$round --workload c2 --max-shift $ms; done
  b c2_pruned_ms0_$round --workload c2 --max-shift 0 --search pruned
  b c2_pruned_ms4_$round --workload c2 --search pruned
done
echo done
