#!/bin/bash
# round 5, lab call 39: head / ungated (lab 37) / gated (lab 38) forms of the zero-SAD row skip on ONE box, interleaved
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab39
mkdir -p $O
cd $R
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {a: round(b * 1e3, 1) for a, b in d.get("kernels_ms", {}).items() if isinstance(b, float)}
print(f"{sys.argv[2]:26s} value {d['value']/1e6:9.4f} M  step {d['ms_per_step']*1e3:7.1f} us  {k}")
PY
}
for round in 1 2 3 4; do
  for lib in head ungated gated; do
    export AOF_LIB=$R/ab/libaof_$lib.so
    b c2_${lib}_$round --workload c2
    b c2_n4_${lib}_$round --workload c2 --noise 4
    b c3_${lib}_$round --workload c3
  done
done
echo done
