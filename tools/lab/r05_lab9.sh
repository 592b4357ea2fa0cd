#!/bin/bash
# round 5, lab call 9: the column walk that reduces in its launch, votes of a walk added once (WalkVotes) -- against search + K3
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab9
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
print(f"{sys.argv[2]:34s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  frac_step {r.get('frac_step')}  {k}")
PY
}
for round in 1 2; do
  b c2_sep_$round --workload c2 --reduce separate
  b c2_fused_$round --workload c2 --reduce fused
  b c2_s2_sep_$round --workload c2 --streams 2 --reduce separate
  b c2_s2_fused_$round --workload c2 --streams 2 --reduce fused
  b c3_sep_$round --workload c3 --reduce separate
  b c3_fused_$round --workload c3 --reduce fused
  b c2_n8_sep_$round --workload c2 --noise 8 --reduce separate
  b c2_n8_fused_$round --workload c2 --noise 8 --reduce fused
  b p128_sep_$round --pairs 128 --reduce separate --steps 2000
  b p128_fused_$round --pairs 128 --reduce fused --steps 2000
  b p256_sep_$round --pairs 256 --reduce separate --steps 1000
  b p256_fused_$round --pairs 256 --reduce fused --steps 1000
  b p512_sep_$round --pairs 512 --reduce separate --steps 1000
  b p512_fused_$round --pairs 512 --reduce fused --steps 1000
done
echo done
