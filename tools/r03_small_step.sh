#!/bin/bash
# configs[3] per-GPU share on ONE GPU (run through gpurun): the 128-pair step next to the 1024-pair step
# on the same box, launched eagerly and as a replayed hipGraph, plus the kernel trace of the 128-pair run.
#   usage: tools/r03_small_step.sh [tag]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-base}
O=$R/gpurun_out/small_step_$T
mkdir -p $O
cd $R
for rep in 1 2; do
  timeout -k 10 200 python3 bench.py --cpu-seconds 0 > $O/c2_1024_$rep.json 2> $O/c2_1024_$rep.err || { echo "1024 failed"; tail -3 $O/c2_1024_$rep.err; exit 1; }
  timeout -k 10 200 python3 bench.py --pairs 128 --steps 200 --cpu-seconds 0 > $O/c2_128_eager_$rep.json 2> $O/c2_128_eager_$rep.err || { echo "128 eager failed"; tail -3 $O/c2_128_eager_$rep.err; exit 1; }
  timeout -k 10 200 python3 bench.py --pairs 128 --steps 200 --cpu-seconds 0 --graph > $O/c2_128_graph_$rep.json 2> $O/c2_128_graph_$rep.err || { echo "128 graph failed"; tail -3 $O/c2_128_graph_$rep.err; exit 1; }
  timeout -k 10 200 python3 bench.py --steps 200 --cpu-seconds 0 --graph > $O/c2_1024_graph_$rep.json 2> $O/c2_1024_graph_$rep.err || { echo "1024 graph failed"; exit 1; }
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt128 -- python3 $R/bench.py --pairs 128 --steps 200 --cpu-seconds 0 > $O/kt128.log 2>&1 || { echo "kernel trace failed"; exit 1; }
python3 $R/tools/summarize_rocprof.py $(ls $O/kt128/*/*kernel_stats.csv | head -1) "bench.py --pairs 128 --steps 200 --cpu-seconds 0" | grep -v "at::native\|Memset\|elementwise\|Cijk\|rocprim\|vectorized" > $O/kernel_stats_128.txt
rm -rf $O/kt128
cd $R
python3 - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:28s} {j['value']:>10.0f} pairs/s  step {j['ms_per_step']*1e3:8.2f} us  K2 {j['roofline']['kernel_ms']*1e3:7.2f} us  {j['kernels_ms']}")
PY
cat $O/kernel_stats_128.txt
