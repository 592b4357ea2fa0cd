#!/usr/bin/env python3
"""Event-timed kernels of small device-resident pairs (the published sparse grid): the one-workgroup
kernel k_flow_small against the separate kernels (aof_set_split_coarse), 1 .. 256 pairs per call.
Each figure includes the ~6 us an event pair costs around any kernel.
    python tools/small_pairs_timing.py            (also: rocprofv3 --kernel-trace --stats -- python3 tools/...)"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

aof = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".synth")
dev = torch.device('cuda:0')
for w, h in ((64, 64), (128, 128)):
    for levels in (1, 2):
        for n in (1, 64, 256):
            kw = dict(pyramid_levels=2, mean_subtract=1) if levels == 2 else {}
            p = aof.px4flow_params(w, h, **kw)
            prevs, curs, _ = synth.make_batch(w, h, n, 4, 5)
            tp, tc = torch.from_numpy(prevs).to(dev), torch.from_numpy(curs).to(dev)
            for split in (0, 1):
                if levels == 1 and split: continue
                eng = aof.FlowEngine(p, 0)
                eng.set_split_coarse(bool(split))
                eng.set_profiling(True)
                for _ in range(20):
                    eng.flow_batch(tp, tc)
                torch.cuda.synchronize()
                ms = [np.median(eng.profile_ms(k)) * 1e3 if eng.profile_ms(k) else 0 for k in range(5)]
                print(f"{w}x{h} levels={levels} n={n} split={split}: us per kernel {np.round(ms, 2)} sum {sum(ms):.2f}")
