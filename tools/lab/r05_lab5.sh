#!/bin/bash
# round 5, lab call 5: where do the column walk's beyond-L2 bytes above the algorithmic ones come from?
# variants (ab/libaof_*.so): ref0 = reference tiles loaded with the default cache policy instead of nt; remap = every XCD gets a
# contiguous chunk of the workgroups (a pair's workgroups share an L2), tail segments last per XCD; remap_ref0 = both
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lab5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -i -E "TCC_(HIT|MISS|REQ|READ|EA0_RDREQ|EA0_WRREQ|TAG_STALL|BUBBLE)" $O/counters.txt | head -40 > $O/counters_tcc.txt
b() { tag=$1; shift; timeout -k 10 200 python3 $R/bench.py "$@" --cpu-seconds 0 --traffic file --legs none > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
      python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
k = {a: b for a, b in d.get("kernels_ms", {}).items() if a != "note"}
print(f"{sys.argv[2]:34s} value {d['value']/1e6:7.3f} M  step {d['ms_per_step']*1e3:7.1f} us  k2 {r.get('kernel_ms')}  {k}")
PY
}
for round in 1 2; do
  for lib in new ref0 remap remap_ref0; do
    if [ $lib = new ]; then unset AOF_LIB; else export AOF_LIB=$R/ab/libaof_$lib.so; fi
    b c2_${lib}_$round --workload c2
    b c3_${lib}_$round --workload c3
    b c2h_${lib}_$round --workload c2h
  done
done
for lib in new ref0 remap remap_ref0; do
  if [ $lib = new ]; then unset AOF_LIB; else export AOF_LIB=$R/ab/libaof_$lib.so; fi
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
    tag=$(echo $set | cut -d" " -f1)
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$lib/$tag -- python3 $R/bench.py --workload c2 --search adaptive --steps 5 --warmup 2 --settle-steps 30 --cpu-seconds 0 --traffic file --legs none > $O/pmc_${lib}_$tag.log 2>&1 || { echo "pmc $lib $tag failed"; tail -3 $O/pmc_${lib}_$tag.log; }
  done
  python3 $R/tools/pmc_summary.py $O/pmc_$lib c2 1024 $O/pmc_c2_$lib.txt > /dev/null
  rm -rf $O/pmc_$lib
  echo "pmc $lib done"
done
unset AOF_LIB
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $set | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_exh/$tag -- python3 $R/bench.py --workload c2 --search exhaustive --steps 5 --warmup 2 --settle-steps 30 --cpu-seconds 0 --traffic file --legs none > $O/pmc_exh_$tag.log 2>&1 || { echo "pmc exh $tag failed"; tail -3 $O/pmc_exh_$tag.log; }
done
python3 $R/tools/pmc_summary.py $O/pmc_exh c2 1024 $O/pmc_c2_exh.txt > /dev/null
rm -rf $O/pmc_exh $O/pmc_traffic.json
echo done
