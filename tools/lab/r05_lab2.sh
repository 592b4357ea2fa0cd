#!/bin/bash
# round 5, lab call 2: column walk with the reduction in its launch; launch-size threshold of the adaptive 8x8 search;
# C3 in sequential sub-batches (does the level-0 search of a sub-batch find its frames in the Infinity Cache?)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab2
mkdir -p $O
cd $R
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { echo "gpu tests failed"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
k = {a: b for a, b in d.get("kernels_ms", {}).items() if a != "note"}
print(f"{sys.argv[2]:30s} value {d['value']/1e6:7.3f} M  step {d['ms_per_step']*1e3:7.1f} us  median {d.get('ms_per_step_median')}  k2 {r.get('kernel_ms')}  frac_step {r.get('frac_step')}  {k}  {d['config'].get('adaptive_search')}")
PY
}
b c2 --workload c2
b c2_fused --workload c2 --reduce fused
b c2_streams2 --workload c2 --streams 2
b c2_streams2_fused --workload c2 --streams 2 --reduce fused
b c3 --workload c3
b c3_fused --workload c3 --reduce fused
b c2h --workload c2h
b c2h_fused --workload c2h --reduce fused
for n in 32 48 64 96 128 192 256; do
  AOF_LAB_PRUNE_MIN_CHUNKS=1 b p${n}_prune --pairs $n
  AOF_LAB_PRUNE_MIN_CHUNKS=1000000000 b p${n}_exh --pairs $n
done
AOF_LAB_PRUNE_MIN_CHUNKS=1 b p128_prune_sep --pairs 128 --reduce separate
AOF_LAB_PRUNE_MIN_CHUNKS=1 b p128_prune_n8 --pairs 128 --noise 8
AOF_LAB_PRUNE_MIN_CHUNKS=1 b p128_prune_n16 --pairs 128 --noise 16
AOF_LAB_PRUNE_MIN_CHUNKS=1000000000 b p128_exh_n16 --pairs 128 --noise 16
export AOF_LAB_OVERLAP_SAME_STREAM=1
for ov in 128 192 256 512; do b c3_seq$ov --workload c3 --overlap $ov; done
AOF_LAB_PRUNE_MIN_CHUNKS=1 b c3_seq128_prune --workload c3 --overlap 128
AOF_LAB_PRUNE_MIN_CHUNKS=1 b c3_seq192_prune --workload c3 --overlap 192
echo done
