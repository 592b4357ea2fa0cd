#!/bin/bash
# round 5, lab call 42: k_coarse with 640 / 768 lanes per workgroup (the three-dy-row search made the kernel 148 VGPRs: three waves per SIMD fit) -- ab/libaof_ctN.so against head
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab42
mkdir -p $O
cd $R
for n in 640 768; do
  AOF_LIB=$R/ab/libaof_ct$n.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c3 or coarse or fused or two_level" > $O/tests$n.log 2>&1 || { echo "tests failed ($n lanes)"; tail -30 $O/tests$n.log; exit 1; }
  tail -1 $O/tests$n.log
done
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {a: round(b * 1e3, 1) for a, b in d.get("kernels_ms", {}).items() if isinstance(b, float)}
print(f"{sys.argv[2]:26s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  frac_step {d['roofline'].get('frac_step')}  {k}")
PY
}
for round in 1 2 3; do
  for n in head ct640 ct768; do
    export AOF_LIB=$R/ab/libaof_$n.so
    b c3_${n}_$round --workload c3
    b c3_s2_${n}_$round --workload c3 --streams 2
    b c3_p256_${n}_$round --workload c3 --pairs 256 --steps 400
  done
done
echo done
