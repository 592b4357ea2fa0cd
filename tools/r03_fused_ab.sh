#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fused_ab
rm -rf $O; mkdir -p $O
cd $R
run() { tag=$1; shift; timeout -k 10 200 python3 bench.py --cpu-seconds 0 --steps 200 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -5 $O/$tag.err; exit 1; }; }
for rep in 1 2; do
  run p1024_s1_sep_$rep --streams 1 --reduce separate
  run p1024_s1_fused_$rep --streams 1 --reduce fused
  run p1024_s2_sep_$rep --streams 2 --reduce separate
  run p1024_s2_fused_$rep --streams 2 --reduce fused
  run p128_s1_sep_$rep --pairs 128 --streams 1 --reduce separate
  run p128_s1_fused_$rep --pairs 128 --streams 1 --reduce fused
  run p128_s2_sep_$rep --pairs 128 --streams 2 --reduce separate
  run p128_s2_fused_$rep --pairs 128 --streams 2 --reduce fused
  run c2h_s1_sep_$rep --workload c2h --streams 1 --reduce separate
  run c2h_s1_fused_$rep --workload c2h --streams 1 --reduce fused
  run c3_s1_sep_$rep --workload c3 --streams 1 --reduce separate
  run c3_s1_fused_$rep --workload c3 --streams 1 --reduce fused
done
python3 - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:24s} {j['value']:>10.0f} pairs/s  step {j['ms_per_step']*1e3:8.2f} us  median {j['ms_per_step_median']*1e3:8.2f}  K2 {j['roofline']['kernel_ms']*1e3:7.2f}  {j['kernels_ms']} parity {j['parity']['oracle_pairs_bit_exact']} {j['parity']['all_pairs_return_known_shift']}")
PY
