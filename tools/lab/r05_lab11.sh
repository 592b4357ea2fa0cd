#!/bin/bash
# round 5, lab call 11: the 8-pixel first test carried from block to block (new) against the commit before (ab/libaof_head.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab11
mkdir -p $O
cd $R
timeout -k 10 400 python3 -m pytest tests/test_gpu_pruned.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "gpu tests failed"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(f"{sys.argv[2]:34s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  k2 {r.get('kernel_ms')*1e3:7.1f} us")
PY
}
for round in 1 2; do
  for lib in head new; do
    if [ $lib = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    for nz in 0 2 4 8 16; do b c2_n${nz}_${lib}_$round --workload c2 --noise $nz; done
    b c3_${lib}_$round --workload c3
    b c3_n4_${lib}_$round --workload c3 --noise 4
    b c2h_${lib}_$round --workload c2h
    b p128_${lib}_$round --pairs 128 --steps 2000
    b p256_${lib}_$round --pairs 256 --steps 1000
  done
done
echo done
