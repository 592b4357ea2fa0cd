#!/bin/bash
# round 5, lab call 15: dword-aligned window loads + scalar byte shift in the column walk (new) against the commit before (ab/libaof_head.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab15
mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { echo "gpu tests failed"; tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 tools/lab/align_probe.py 2>&1 | grep shift
b() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
k = {a: b for a, b in d.get("kernels_ms", {}).items() if a not in ("note",)}
print(f"{sys.argv[2]:30s} value {d['value']/1e6:7.4f} M  step {d['ms_per_step']*1e3:7.1f} us  frac_step {r.get('frac_step')}  {k}")
PY
}
for round in 1 2; do
  for lib in head new; do
    if [ $lib = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
    b c3_${lib}_$round --workload c3
    b c3_n8_${lib}_$round --workload c3 --noise 8
    b c3_s2_${lib}_$round --workload c3 --streams 2
    b c3n_${lib}_$round --workload c3n
    b c2_${lib}_$round --workload c2
    b c2_n8_${lib}_$round --workload c2 --noise 8
    b c2h_${lib}_$round --workload c2h
    b p128_${lib}_$round --pairs 128 --steps 2000
  done
done
echo done
