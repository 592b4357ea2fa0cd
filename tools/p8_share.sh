mkdir -p gpurun_out/sh
for s in auto pruned exhaustive; do
  python bench.py --pairs 128 --steps 200 --search $s --cpu-seconds 0 --traffic file > gpurun_out/sh/p128_$s.json 2>/dev/null
  python bench.py --pairs 256 --steps 200 --search $s --cpu-seconds 0 --traffic file > gpurun_out/sh/p256_$s.json 2>/dev/null
done
python bench.py --streams 2 --cpu-seconds 0 --traffic file > gpurun_out/sh/p1024_two.json 2>/dev/null
python bench.py --cpu-seconds 0 --traffic file > gpurun_out/sh/p1024_one.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/sh/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["config"]
    print(f.split("/")[-1], round(d["value"]/1e6,3), d["ms_per_step"]*1e3, c["search"], c["reduce"], c["streams"], c["graph_replay"], c.get("adaptive_search"))
PY
