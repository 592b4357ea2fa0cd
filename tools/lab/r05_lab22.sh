#!/bin/bash
# round 5, lab call 22: the column walk with its rows through LDS (buffer_load ... lds, one step ahead) against the register walk
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab22
mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests/test_gpu_pruned.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(f"{sys.argv[2]:30s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  k2 {r.get('kernel_ms')*1e3:7.1f}")
PY
}
for round in 1 2; do
  for dma in 0 1; do
    export AOF_LAB_COLS_DMA=$dma
    b c2_dma${dma}_$round --workload c2
    b c3_dma${dma}_$round --workload c3
    b c2n8_dma${dma}_$round --workload c2 --noise 8
    b c3n16_dma${dma}_$round --workload c3 --noise 16
  done
done
echo done
