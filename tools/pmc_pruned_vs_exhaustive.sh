#!/bin/bash
# PMC comparison of the exhaustive and the pruned 8x8 search kernel on the headline batch (c2), one box.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_pruned; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for mode in exhaustive adaptive; do
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS"; do
    tag=$(echo $set | cut -d" " -f1)
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${mode}_$tag -- python3 $R/bench.py --search $mode --steps 30 --settle-steps 100 --cpu-seconds 0 --traffic file > $O/${mode}_$tag.log 2>&1 || { echo "$mode $tag failed"; tail -3 $O/${mode}_$tag.log; }
  done
done
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_pruned"
for mode in ("exhaustive","adaptive"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{mode}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "k_search_lane8" not in k: continue
            k=k.split("(anonymous namespace)::", 1)[-1].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        if len(next(iter(acc[k].values()))) < 20: continue
        print(mode, k)
        for c in sorted(acc[k]):
            v=acc[k][c]; print(f"    {c:<24} n={len(v):<4} mean={sum(v)/len(v):.6g}")
PY
