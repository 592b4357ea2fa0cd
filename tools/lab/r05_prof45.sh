#!/bin/bash
# round 5: how long does k_tile16_probe run with its deeper phases? (kernel trace of c5 at +-40 and +-8 LSB)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof45
mkdir -p $O
cd $R
for nz in 40 8; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n$nz -- python3 $R/bench.py --workload c5 --pairs 256 --noise $nz --cpu-seconds 0 --traffic file --legs none > $O/n$nz.log 2>&1 < /dev/null || { echo "trace $nz failed"; tail -3 $O/n$nz.log; exit 1; }
  f=$(find $O/n$nz -name "*kernel_stats.csv" | head -1)
  echo "noise $nz: $f"
  if [ -n "$f" ]; then head -5 "$f" | cut -c1-150; fi
done
echo done
