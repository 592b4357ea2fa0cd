#!/bin/bash
# round 5, lab call 23: longer column walks (less halo between a column's segments), register walk and LDS walk
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab23
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
  for rows in 8 10 15 20 30; do
    for dma in 0 1; do
      export AOF_LAB_COLS_DMA=$dma AOF_LAB_COLS_ROWS=$rows
      b c2_rows${rows}_dma${dma}_$round --workload c2
    done
  done
  for rows in 8 15; do
    export AOF_LAB_COLS_DMA=0 AOF_LAB_COLS_ROWS=$rows
    b c3_rows${rows}_$round --workload c3
  done
done
echo done
