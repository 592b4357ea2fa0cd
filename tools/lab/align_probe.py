#!/usr/bin/env python3
"""Lab: C3's level-0 search against the byte alignment of its windows -- every pair of the batch shifted by the same (sx, 0)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
aof = ge.load_package()
dev = torch.device("cuda:0")
W, H, n, reach = 640, 480, 1024, 9
g = torch.Generator(device=dev); g.manual_seed(7)
p = aof.default_params(W, H, pyramid_levels=2, mean_subtract=int(os.environ.get("EQ", "1")))
prev = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
canv = []
for s in range(0, n, 64):
    raw = torch.randint(0, 256, (64, H + 2 * reach + 2, W + 2 * reach + 2), generator=g, device=dev, dtype=torch.int32)
    hc, wc = H + 2 * reach, W + 2 * reach
    acc = torch.zeros((64, hc, wc), dtype=torch.int32, device=dev)
    for oy in range(3):
        for ox in range(3):
            acc += raw[:, oy:oy + hc, ox:ox + wc]
    c = ((acc + 4) // 9).to(torch.uint8)
    canv.append(c)
    prev[s:s + 64] = c[:, reach:reach + H, reach:reach + W]
eng = aof.FlowEngine(p, 0)
for sx, sy in [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (8, 0), (0, 3), (4, 4), (7, 5)]:
    cur = torch.empty_like(prev)
    for k, c in enumerate(canv):
        cc = c[:, reach - sy:reach - sy + H, reach - sx:reach - sx + W]
        cur[64 * k:64 * k + 64] = (cc.to(torch.int32) + (9 if p.mean_subtract else 0)).clamp_(0, 255).to(torch.uint8)
    for _ in range(300):
        eng.flow_batch(prev, cur)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    for _ in range(100):
        eng.flow_batch(prev, cur)
    torch.cuda.synchronize()
    ms = eng.profile_ms(aof.K_SEARCH)
    co = eng.profile_ms(aof.K_PYRAMID)
    eng.set_profiling(False)
    print(f"shift ({sx:2d},{sy:2d}): level-0 search {np.mean(ms)*1e3:7.1f} us   coarse {np.mean(co)*1e3:7.1f} us   {eng.search_stats()}", flush=True)
