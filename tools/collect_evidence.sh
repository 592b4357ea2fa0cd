#!/bin/bash
# Run on the GPU box (through gpurun): bench lines, rocprofv3 kernel-trace stats and the
# HBM-traffic PMC passes for the bench workloads, all under gpurun_out/evidence/.
# rocprofv3 gets the program itself after "--" and --pmc is never combined with other traces
# than --kernel-trace.   usage: tools/collect_evidence.sh [workload ...]   (default: all)
# Besides the workloads of bench.py --workload, "share" collects configs[3]'s per-GPU share: the
# 128-pair C2 step with one batch in flight / separate K3 and with two batches in flight / the
# reduction inside the search launch, next to the 1 024-pair step of the same box; "lanes" the
# multi-kernel workloads with two batches in flight; "latency" the per-call path; "noise" the search modes of the 8x8
# workloads against noise on the newer frame.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WL=${@:-c2 c3 c5 c2h c1b c5h ingest derotate seq share lanes inputs ab latency noise}
args_of() {   # bench.py arguments and the launch size key of a workload
    case $1 in
        c5|c5h) echo "--pairs 256";;
        c1b) echo "--pairs 65536";;
        ingest) echo "--pairs 1024";;
        derotate) echo "--pairs 1024";;
        seq) echo "--pairs 1024 --steps 20";;
        *) echo "";;
    esac
}
# profiler runs name the headline's search mode: the default line also times the other two modes behind its timed region,
# and the pruned kernels serve both the adaptive and the always-pruned mode
mode_of() { case $1 in c2|c3|c2h|c5|c5h) echo "--search adaptive";; *) echo "";; esac; }
key_of() { case $1 in c5|c5h) echo 256;; c1b) echo 65536;; ingest) echo 8192;; derotate) echo 1048576;; seq) echo 65536;; *) echo 1024;; esac; }
trace() {   # tag, bench arguments...: kernel-trace summary of one bench command
    tag=$1; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 $R/bench.py "$@" --cpu-seconds 0 --legs none > $O/kt_$tag.log 2>&1 || { echo "kernel trace $tag failed"; tail -3 $O/kt_$tag.log; exit 1; }
    python3 $R/tools/summarize_rocprof.py $(ls $O/kt_$tag/*/*kernel_stats.csv | head -1) "bench.py $* --cpu-seconds 0 --legs none" | grep -v "at::native\|Memset\|elementwise\|Cijk\|rocprim\|vectorized" > $O/kernel_stats_$tag.txt
    rm -rf $O/kt_$tag
}
line() {   # tag, bench arguments...
    tag=$1; shift
    timeout -k 10 300 python3 $R/bench.py "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err || { echo "bench $tag failed"; tail -3 $O/bench_$tag.err; exit 1; }
}
for wl in $WL; do
    case $wl in
    share)
        # (a 20 us step: thousands of steps, or the clocks have not settled -- bench.py settles for ~0.2 s by itself)
        line share_p1024 --cpu-seconds 0 --traffic file --legs none
        line share_p1024_two_batches --streams 2 --cpu-seconds 0 --traffic file --legs none
        line share_p1024_exhaustive --search exhaustive --cpu-seconds 0 --traffic file --legs none
        line share_p1024_exhaustive_two_batches --search exhaustive --streams 2 --cpu-seconds 0 --traffic file --legs none
        line share_p512 --pairs 512 --steps 1000 --cpu-seconds 0 --traffic file --legs none
        line share_p256 --pairs 256 --steps 1000 --cpu-seconds 0 --traffic file --legs none
        line share_p128 --pairs 128 --steps 2000 --cpu-seconds 0 --traffic file --legs none
        line share_p128_separate --pairs 128 --steps 2000 --reduce separate --cpu-seconds 0 --traffic file --legs none
        line share_p128_one_batch --pairs 128 --steps 2000 --streams 1 --cpu-seconds 0 --traffic file --legs none
        line share_p128_eager --pairs 128 --steps 2000 --graph off --cpu-seconds 0 --traffic file --legs none
        line share_p128_exhaustive --pairs 128 --steps 2000 --search exhaustive --cpu-seconds 0 --traffic file --legs none
        line share_p128_noise16 --pairs 128 --steps 2000 --noise 16 --cpu-seconds 0 --traffic file --legs none
        line share_p128_realistic --pairs 128 --steps 2000 --input realistic --cpu-seconds 0 --traffic file --legs none
        line share_p64 --pairs 64 --steps 2000 --cpu-seconds 0 --traffic file --legs none
        trace share_p128 --pairs 128 --steps 500 --legs none
        trace share_p128_separate --pairs 128 --steps 500 --reduce separate --legs none
        echo "share done";;
    lanes)
        for w2 in c2 c3 c2h c1b c5 c5h; do
            line lanes_$w2 --workload $w2 $(args_of $w2) --streams 2 --reduce auto --cpu-seconds 0 --traffic file --legs none
        done
        line lanes_c3_p512 --workload c3 --pairs 512 --streams 2 --cpu-seconds 0 --traffic file --legs none
        line lanes_c3_noise16 --workload c3 --noise 16 --streams 2 --cpu-seconds 0 --traffic file --legs none
        echo "lanes done";;
    inputs)
        # the workloads on the other inputs: +-16 LSB noise, the realistic input
        for w2 in c2 c3 c2h c5 c5h; do
            line ${w2}_noise16 --workload $w2 $(args_of $w2) --noise 16 --cpu-seconds 0 --traffic file --legs none
            line ${w2}_realistic --workload $w2 $(args_of $w2) --input realistic --cpu-seconds 0 --traffic file --legs none
        done
        echo "inputs done";;
    ab)
        # this round's kernels against round 4's library (ab/libaof_r04.so), interleaved, one box
        if [ -f $R/ab/libaof_r04.so ]; then
            for round in 1 2; do
                for lib in r04 r05; do
                    if [ $lib = r04 ]; then export AOF_LIB=$R/ab/libaof_r04.so; else unset AOF_LIB; fi
                    line ab_${lib}_c2_$round --workload c2 --cpu-seconds 0 --traffic file --legs none
                    line ab_${lib}_c2_noise8_$round --workload c2 --noise 8 --cpu-seconds 0 --traffic file --legs none
                    line ab_${lib}_c3_$round --workload c3 --cpu-seconds 0 --traffic file --legs none
                    line ab_${lib}_c2h_$round --workload c2h --cpu-seconds 0 --traffic file --legs none
                    line ab_${lib}_c5_$round --workload c5 --pairs 256 --cpu-seconds 0 --traffic file --legs none
                    line ab_${lib}_c5h_$round --workload c5h --pairs 256 --cpu-seconds 0 --traffic file --legs none
                    line ab_${lib}_p128_$round --pairs 128 --steps 2000 --cpu-seconds 0 --traffic file --legs none
                    line ab_${lib}_p256_$round --pairs 256 --steps 1000 --cpu-seconds 0 --traffic file --legs none
                done
            done
            unset AOF_LIB
            python3 - $O <<'PY' > $O/ab_r04.txt
import glob, json, os, sys
O = sys.argv[1]
print("# this round's kernels (r05) against round 4's library (r04 = ab/libaof_r04.so, commit db57870), interleaved, two rounds, one box;")
print("# bench.py --legs none: M pairs/s, us per step, K2 us")
for f in sorted(glob.glob(os.path.join(O, "bench_ab_*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{os.path.basename(f)[9:-5]:28s} {d['value']/1e6:8.3f} M  step {d['ms_per_step']*1e3:7.1f} us  K2 {d['roofline']['kernel_ms']*1e3:7.1f} us  {d['config'].get('adaptive_search', '')}")
PY
            rm -f $O/bench_ab_*.json $O/bench_ab_*.err
        fi
        echo "ab done";;
    noise)
        cd $R && tools/search_modes_vs_noise.sh $O/noise c2 c3 c2h > $O/search_modes_vs_noise.txt 2>&1 || { echo "noise sweep failed"; tail -3 $O/search_modes_vs_noise.txt; exit 1; }
        cd /tmp; rm -rf $O/noise
        echo "noise done";;
    latency)
        cd $R && tools/stream_latency.sh > $O/stream_latency.txt 2> $O/stream_latency.err; cd /tmp
        timeout -k 10 200 python3 $R/bench.py --workload c1 --pairs 256 --steps 20 > $O/bench_c1.json 2> $O/bench_c1.err || { echo "bench c1 failed"; exit 1; }
        echo "latency done";;
    *)
        extra=$(args_of $wl)
        line $wl --workload $wl $extra
        trace $wl --workload $wl $extra $(mode_of $wl)
        echo "$wl done";;
    esac
done
for wl in $WL; do
    case $wl in share|lanes|inputs|ab|latency|noise) continue;; esac
    extra="$(args_of $wl) $(mode_of $wl)"; pairs=$(key_of $wl)
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"; do
        tag=$(echo $set | cut -d" " -f1)
        timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$wl/$tag -- python3 $R/bench.py --workload $wl $extra --steps 5 --warmup 2 --settle-steps 0 --cpu-seconds 0 --legs none > $O/pmc_${wl}_$tag.log 2>&1 || { echo "pmc $wl $tag failed"; exit 1; }
    done
    python3 $R/tools/pmc_summary.py $O/pmc_$wl $wl $pairs $O/pmc_$wl.txt > /dev/null
    rm -rf $O/pmc_$wl
    echo "pmc $wl done"
done
