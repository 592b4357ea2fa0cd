#!/usr/bin/env python3
"""Re-runs one case of tools/fuzz_gpu.py's main mode (seed s) through several device paths and prints where they differ
from the oracle.   python tools/repro_seed.py <seed>"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
aof = ge.load_package()
from oracle import pyoracle as orc
import importlib
synth = importlib.import_module("aero_optical_flow_amd.synth")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_gpu

s = int(sys.argv[1])
rng = np.random.default_rng(50000 + s)
kw = fuzz_gpu.case(rng)
p = aof.default_params(**kw)
reach = 2 * p.search + 1 if p.pyramid_levels == 2 else p.search
n = 2
prevs, curs, _ = synth.make_batch(p.width, p.height, n, reach, 90000 + 3 * s, noise=int(rng.choice([0, 0, 3, 25])),
                                  brightness=int(rng.integers(-40, 41)), contrast=float(rng.choice([1.0, 1.0, 3.0, 0.15])))
style = int(rng.integers(0, 7))
print("case", kw, "style", style)
if style != 0:
    print("style != 0: this script reproduces style 0 cases only"); sys.exit(2)
po = orc.params_from(p)
refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(n)]
dev = torch.device("cuda:0")
tp, tc = torch.from_numpy(prevs).to(dev), torch.from_numpy(curs).to(dev)
for mode in ("exhaustive", "pruned", "generic", "split"):
    eng = aof.FlowEngine(p, 0)
    if mode == "exhaustive": eng.set_search_mode(aof.SEARCH_EXHAUSTIVE)
    if mode == "pruned": eng.set_search_mode(aof.SEARCH_PRUNED)
    if mode == "generic": eng.force_generic(True)
    if mode == "split": eng.set_split_coarse(True)
    nb = eng.nblocks(0)
    sub = torch.full((n, nb), 99, dtype=torch.uint8, device=dev) if p.subpixel else None
    blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
    torch.cuda.synchronize()
    gb = aof.blocks_view(blocks)
    for i in range(n):
        d = np.nonzero(gb[i].view(np.uint32) != refs[i]["blocks"].view(np.uint32))[0]
        ds = np.nonzero(sub[i].cpu().numpy() != refs[i]["subdirs"])[0] if sub is not None else []
        print(mode, eng.variant, "pair", i, "records differ at", d[:8], "directions differ at", ds[:8],
              [(int(k), gb[i][k], int(sub[i][k]), int(refs[i]["subdirs"][k])) for k in ds[:4]])
    eng.close()
g = aof.grid(p, 0)
print("grid", g, "flows", [r["flow"] for r in refs], "l1 flows", [orc.flow_pair(po, prevs[i], curs[i]).get("flow_l1") for i in range(n)])
