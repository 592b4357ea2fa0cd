#!/bin/bash
# Run on the GPU box (through gpurun): bench lines, rocprofv3 kernel-trace stats and the
# HBM-traffic PMC passes for the bench workloads, all under gpurun_out/evidence/.
# rocprofv3 gets the program itself after "--" and --pmc is never combined with other traces
# than --kernel-trace.   usage: tools/collect_evidence.sh [workload ...]   (default: all)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WL=${@:-c2 c3 c5 c2h c1b c5h ingest derotate}
args_of() {   # bench.py arguments and the launch size key of a workload
    case $1 in
        c5|c5h) echo "--pairs 256";;
        c1b) echo "--pairs 65536";;
        ingest) echo "--pairs 1024";;
        derotate) echo "--pairs 1024";;
        *) echo "";;
    esac
}
key_of() { case $1 in c5|c5h) echo 256;; c1b) echo 65536;; ingest) echo 8192;; derotate) echo 1048576;; *) echo 1024;; esac; }
for wl in $WL; do
    extra=$(args_of $wl)
    timeout -k 10 300 python3 $R/bench.py --workload $wl $extra > $O/bench_$wl.json 2> $O/bench_$wl.err || { echo "bench $wl failed"; tail -3 $O/bench_$wl.err; exit 1; }
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$wl -- python3 $R/bench.py --workload $wl $extra --cpu-seconds 0 > $O/kt_$wl.log 2>&1 || { echo "kernel trace $wl failed"; exit 1; }
    python3 $R/tools/summarize_rocprof.py $(ls $O/kt_$wl/*/*kernel_stats.csv | head -1) "bench.py --workload $wl $extra --cpu-seconds 0" | grep -v "at::native\|Memset\|elementwise\|Cijk\|rocprim\|vectorized" > $O/kernel_stats_$wl.txt
    rm -rf $O/kt_$wl
    echo "$wl done"
done
for wl in $WL; do
    extra=$(args_of $wl); pairs=$(key_of $wl)
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"; do
        tag=$(echo $set | cut -d" " -f1)
        timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$wl/$tag -- python3 $R/bench.py --workload $wl $extra --steps 5 --warmup 2 --settle-steps 0 --cpu-seconds 0 > $O/pmc_${wl}_$tag.log 2>&1 || { echo "pmc $wl $tag failed"; exit 1; }
    done
    python3 $R/tools/pmc_summary.py $O/pmc_$wl $wl $pairs $O/pmc_$wl.txt > /dev/null
    rm -rf $O/pmc_$wl
    echo "pmc $wl done"
done
