#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kt_lanes; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in 1 2; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_s$s -- python3 $R/bench.py --steps 200 --cpu-seconds 0 --streams $s > $O/kt_s$s.log 2>&1 || { echo "kernel trace failed"; tail -5 $O/kt_s$s.log; exit 1; }
python3 $R/tools/summarize_rocprof.py $(ls $O/kt_s$s/*/*kernel_stats.csv | head -1) "bench.py --steps 200 --cpu-seconds 0 --streams $s" | grep -v "at::native\|Memset\|elementwise\|Cijk\|rocprim\|vectorized"
grep "^{" $O/kt_s$s.log | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 'step', j['ms_per_step'], 'K2 events', j['roofline']['kernel_ms'])"
rm -rf $O/kt_s$s
done
