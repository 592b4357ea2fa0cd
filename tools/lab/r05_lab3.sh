#!/bin/bash
# round 5, lab call 3: shape of the pruned column walk on small launches (workgroup size, walk length, batches in flight)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab3
mkdir -p $O
cd $R
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
k = {a: b for a, b in d.get("kernels_ms", {}).items() if a != "note"}
print(f"{sys.argv[2]:34s} value {d['value']/1e6:7.3f} M  step {d['ms_per_step']*1e3:7.1f} us  median {d.get('ms_per_step_median')}  k2 {r.get('kernel_ms')}  {k}")
PY
}
export AOF_LAB_PRUNE_MIN_CHUNKS=1
for n in 64 96 128 192; do
  for sw in 0 100000; do
    for wv in 3072 2048 1280; do
      AOF_LAB_COLS_SMALL_WAVES=$sw AOF_LAB_COLS_WAVES=$wv b p${n}_sw${sw}_wv${wv}_sep --pairs $n --reduce separate
    done
  done
  AOF_LAB_COLS_SMALL_WAVES=100000 AOF_LAB_COLS_WAVES=2048 b p${n}_sw100000_wv2048_fused --pairs $n --reduce fused
done
for st in 3 4; do
  AOF_LAB_COLS_SMALL_WAVES=0 AOF_LAB_COLS_WAVES=3072 b p128_streams${st}_sep --pairs 128 --reduce separate --streams $st
  AOF_LAB_COLS_SMALL_WAVES=100000 AOF_LAB_COLS_WAVES=2048 b p128_streams${st}_sw_sep --pairs 128 --reduce separate --streams $st
done
AOF_LAB_PRUNE_MIN_CHUNKS=1000000000 b p128_exh_streams3 --pairs 128 --streams 3
b c2_1024_streams1 --workload c2
echo done
