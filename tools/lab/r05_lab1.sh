#!/bin/bash
# round 5, lab call 1: start-row vote (8x8 pruned kernels) + sub-batch overlap of the C3 step.  Same box for every line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab1
mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { echo "gpu tests failed"; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(f"{sys.argv[2]:34s} value {d['value']/1e6:7.3f} M  step {d['ms_per_step']:.4f} ms  median {d.get('ms_per_step_median')}  k2 {r.get('kernel_ms')}  frac_step {r.get('frac_step')}  kernels {d.get('kernels_ms')}  adaptive {d['config'].get('adaptive_search')}")
PY
}
b c2 --workload c2
b c2_n8 --workload c2 --noise 8
b c2_n16 --workload c2 --noise 16
for ov in 0 -1 512 128 192; do b c3_ov$ov --workload c3 --overlap $ov; done
for ov in 0 -1 512; do b c3_n16_ov$ov --workload c3 --noise 16 --overlap $ov; done
b c3_streams2_ov0 --workload c3 --streams 2 --overlap 0
b c3_streams2_ov256 --workload c3 --streams 2 --overlap -1
b p128 --pairs 128
b p256 --pairs 256
for mc in 1; do
  export AOF_LAB_PRUNE_MIN_CHUNKS=$mc
  b p128_prune --pairs 128
  b p128_prune_sep1 --pairs 128 --streams 1 --reduce separate
  b p256_prune --pairs 256
  b p64_prune --pairs 64
  b p128_prune_n16 --pairs 128 --noise 16
  unset AOF_LAB_PRUNE_MIN_CHUNKS
done
b p64 --pairs 64
b p128_n16 --pairs 128 --noise 16
echo done
