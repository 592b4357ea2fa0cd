"""One line of the box-to-box spread table from a bench.py JSON line on stdin (tools/box_spread.sh)."""
import json
import sys

j = json.loads(sys.stdin.read().strip().splitlines()[-1])
kernels = "  ".join(f"{n} {v:.5f}" for n, v in j.get("kernels_ms", {}).items())
print(f"{sys.argv[1]:24s} {j['value']:>12,.0f}  {j['ms_per_step']:.4f} ms/step  "
      f"median {j.get('ms_per_step_median') or 0:.4f}  {kernels}")
