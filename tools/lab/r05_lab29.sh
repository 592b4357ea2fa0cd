#!/bin/bash
# round 5, lab call 29: the column walk's waves placed so that a pair's waves all run on ONE XCD (pair % 8), pairs interleaved over the XCDs
# (one-wave workgroups, AOF_LAB_COLS_XCD=1) -- against launch order (=0); libaof_xcd_nt0.so: the same with the tiles loaded with the default cache policy
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab29
mkdir -p $O
cd $R
export AOF_LAB_COLS_XCD=1
timeout -k 10 600 python3 -m pytest tests/test_gpu_pruned.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 300 python3 bench.py "$@" --cpu-seconds 0 --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels_ms", {})
print(f"{sys.argv[2]:30s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  search {k.get('search', 0)*1e3:7.1f}  traffic {d['roofline'].get('traffic')} {d['roofline'].get('traffic_source','')[:40]}")
PY
}
for round in 1 2; do
  for v in "0 xcd" "1 xcd" "0 xcd_nt0" "1 xcd_nt0"; do
    set -- $v
    export AOF_LAB_COLS_XCD=$1 AOF_LIB=$R/ab/libaof_$2.so
    if [ $round = 1 ]; then t=live; else t=file; fi
    b c2_x$1_$2_$round --workload c2 --traffic $t
    b c3_x$1_$2_$round --workload c3 --traffic file
  done
done
echo done
