#!/bin/bash
# round 5, lab call 27: misaligned pairs of the pruned column walk without the lonely lanes' extra dword loads (they load their bytes where they lie
# and hand their left neighbour the end of dword 1) -- new against the commit before (ab/libaof_head.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab27
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
for round in 1 2 3; do
  for lib in head new; do
    if [ $lib = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    b c3_${lib}_$round --workload c3
    b c3_n8_${lib}_$round --workload c3 --noise 8
    b c2h_${lib}_$round --workload c2h
    b c2_${lib}_$round --workload c2
  done
done
echo done
