#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/streams3
rm -rf $O; mkdir -p $O
cd $R
run() { tag=$1; shift; timeout -k 10 200 python3 bench.py --cpu-seconds 0 --steps 100 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -5 $O/$tag.err; exit 1; }; }
for s in 1 2 3; do
  run c3_s$s --workload c3 --streams $s
  run c3_p512_s$s --workload c3 --pairs 512 --streams $s
  run c2h_s$s --workload c2h --streams $s
  run c5_s$s --workload c5 --pairs 256 --streams $s
  run c5h_s$s --workload c5h --pairs 256 --streams $s
  run c1b_s$s --workload c1b --pairs 65536 --streams $s
done
run c3_split_s2 --workload c3 --streams 2 --coarse split
run c3_split_s3 --workload c3 --streams 3 --coarse split
python3 - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:20s} {j['value']:>12.0f} pairs/s  step {j['ms_per_step']*1e3:8.2f} us  median {j['ms_per_step_median']*1e3:8.2f}  frac_step {j['roofline']['frac_step']:.4f}  {j['kernels_ms']}  parity {j['parity']['oracle_pairs_bit_exact']} {j['parity']['all_pairs_return_known_shift']}")
PY
