set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/strips; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --workload c2 --search strips > $O/bench_c2_strips.json 2> $O/err.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload c2 --search strips --cpu-seconds 0 > $O/kt.log 2>&1 || exit 1
python3 $R/tools/summarize_rocprof.py $(ls $O/kt/*/*kernel_stats.csv | head -1) "bench.py --workload c2 --search strips --cpu-seconds 0" | grep -v "at::native\|Memset\|elementwise\|Cijk\|rocprim\|vectorized" > $O/kernel_stats_c2_strips.txt
rm -rf $O/kt; cat $O/kernel_stats_c2_strips.txt
