#!/usr/bin/env python3
"""C5 (1280x960, 16x16 SAD, +-8): K2 time of the three search modes over a sweep of sensor noise, one box,
one process:  python tools/c5_adaptive_sweep.py [--pairs 256] [--steps 30] [--noise 0,2,4,...] [--subpixel]
(AOF_LIB selects another build of libaof.so).  Also checks that the three modes write identical records."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--noise", default="0,2,4,6,8,10,12,20,40")
    ap.add_argument("--subpixel", action="store_true")
    ap.add_argument("--max-shift", type=int, default=8)
    args = ap.parse_args()
    aof = ge.load_package()
    dev = torch.device("cuda:0")
    W, H = 1280, 960
    p = aof.default_params(W, H, tile=16, search=8, value_threshold=12000, subpixel=int(args.subpixel))
    prev, cur0, _ = bench.make_batch_gpu(W, H, args.pairs, args.max_shift, 0xA0F, dev)
    alg = aof.algorithmic_bytes(p)
    modes = [("exhaustive", aof.SEARCH_EXHAUSTIVE), ("pruned", aof.SEARCH_PRUNED), ("adaptive", aof.SEARCH_ADAPTIVE)]
    print(f"# lib {os.environ.get('AOF_LIB', 'in-tree')}  pairs {args.pairs}  steps {args.steps}  subpixel {int(args.subpixel)}  max shift {args.max_shift}")
    print("# noise   " + "   ".join(f"{m:>10s} ms" for m, _ in modes) + "   adaptive/exhaustive  adaptive/best   % of 8 TB/s (adaptive)  records")
    for nz in [int(x) for x in args.noise.split(",")]:
        cur = cur0
        if nz:
            g = torch.Generator(device=dev)
            g.manual_seed(99)
            noise = torch.randint(-nz, nz + 1, cur0.shape, generator=g, device=dev, dtype=torch.int16)
            cur = (cur0.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
            del noise
        ms, recs = {}, {}
        for name, mode in modes:
            eng = aof.FlowEngine(p, 0)
            eng.set_search_mode(mode)
            blocks, flows, ws = eng.flow_batch(prev, cur)
            for _ in range(5):
                eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
            torch.cuda.synchronize()
            eng.set_profiling(True, kernels=[aof.K_SEARCH])
            for _ in range(args.steps):
                eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
            torch.cuda.synchronize()
            ms[name] = float(np.mean(eng.profile_ms(aof.K_SEARCH)))
            recs[name] = (blocks.clone(), flows.clone())
            eng.close()
        same = all(torch.equal(recs[m][0], recs["exhaustive"][0]) and torch.equal(recs[m][1], recs["exhaustive"][1]) for m, _ in modes)
        best = min(ms["exhaustive"], ms["pruned"])
        frac = alg * args.pairs / (ms["adaptive"] * 1e-3) / 8e12
        print(f"  {nz:3d}     " + "   ".join(f"{ms[m]:13.4f}" for m, _ in modes) +
              f"   {ms['adaptive'] / ms['exhaustive']:19.3f}  {ms['adaptive'] / best:13.3f}   {100 * frac:22.1f}  {'identical' if same else 'DIFFER'}")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
