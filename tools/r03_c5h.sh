#!/bin/bash
# c5h (1280x960, 16x16 SAD +-8, half-pixel refinement): search -> refine in sub-batches sized for the
# 256 MiB memory-side cache (ab/sub<MB>.so; sub100000 = one pass over the whole batch), same box.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c5h
rm -rf $O; mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in sub100000 sub64 sub96 sub128 sub160; do
    AOF_LIB=$R/ab/$lib.so timeout -k 10 200 python3 bench.py --workload c5h --pairs 256 --cpu-seconds 0 --steps 100 > $O/${lib}_$rep.json 2> $O/${lib}_$rep.err || { echo "$lib failed"; tail -5 $O/${lib}_$rep.err; exit 1; }
  done
done
python3 - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:22s} {j['value']:>10.0f} pairs/s  step {j['ms_per_step']*1e3:8.2f} us  K2 {j['roofline']['kernel_ms']*1e3:7.2f} us  parity {j['parity']['oracle_pairs_bit_exact']}")
PY
cd /tmp && export TMPDIR=/tmp
for lib in sub100000 sub96; do
AOF_LIB=$R/ab/$lib.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$lib -- python3 $R/bench.py --workload c5h --pairs 256 --cpu-seconds 0 --steps 100 > $O/kt_$lib.log 2>&1 || { echo "kernel trace failed"; exit 1; }
python3 $R/tools/summarize_rocprof.py $(ls $O/kt_$lib/*/*kernel_stats.csv | head -1) "AOF_LIB=ab/$lib.so bench.py --workload c5h --pairs 256 --cpu-seconds 0 --steps 100" | grep -v "at::native\|Memset\|elementwise\|Cijk\|rocprim\|vectorized" | tee $O/kernel_stats_$lib.txt
rm -rf $O/kt_$lib
done
