set -o pipefail
O=gpurun_out/final_bench; mkdir -p $O
for wl in c2 c3 c2h c5 c5h c1b ingest derotate; do
  extra=""; case $wl in c5|c5h) extra="--pairs 256";; c1b) extra="--pairs 65536";; ingest|derotate) extra="--pairs 1024";; esac
  timeout -k 10 300 python3 bench.py --workload $wl $extra > $O/bench_$wl.json 2> $O/bench_$wl.err || { echo "bench $wl failed"; tail -3 $O/bench_$wl.err; exit 1; }
  echo "$wl done"
done
timeout -k 10 100 python3 bench.py --workload c1 --pairs 256 --steps 20 > $O/bench_c1.json 2> $O/bench_c1.err; echo c1 done
