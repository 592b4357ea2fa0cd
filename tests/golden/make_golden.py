#!/usr/bin/env python3
"""Generates tests/golden/flow_golden.npz: small seeded frame pairs with the outputs
of this repo's CPU oracle (block records, half-pixel directions, flow records).

The reference ships no golden vectors and its engine cannot be built or imported
here, so these fixtures pin THIS BUILD's spec: they detect
any later drift of the oracle or the HIP path.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

CASES = [
    ("c1_px4flow_64", dict(width=64, height=64, grid_mode=1, subpixel=1), 4, dict(noise=2)),
    ("c1_dense_64", dict(width=64, height=64), 4, dict(noise=4)),
    ("two_level_mean_96x64", dict(width=96, height=64, pyramid_levels=2, mean_subtract=1), 9,
     dict(noise=3, brightness=11)),
    ("tile16_search8_96", dict(width=96, height=96, tile=16, search=8, value_threshold=12000), 8,
     dict(noise=3)),
    ("average_subpixel_72x56", dict(width=72, height=56, subpixel=1, hist_filter=0), 4, dict(noise=6)),
]


def main():
    ge.load_package()
    synth = importlib.import_module("aero_optical_flow_amd.synth")
    out = {}
    for name, kw, reach, skw in CASES:
        p = orc.default_params(**kw)
        n = 4
        prevs, curs, shifts = synth.make_batch(p.width, p.height, n, reach, 5000, **skw)
        half = synth.make_pair(p.width, p.height, reach, 5001, shift=(1, -1), half=(1, 0))
        prevs[3], curs[3] = half[0], half[1]
        g = orc.grid(p, 0)
        blocks = np.zeros((n, g.nx * g.ny), orc.BLOCK_DTYPE)
        subdirs = np.zeros((n, g.nx * g.ny), np.uint8)
        flows = np.zeros(n, orc.FLOW_DTYPE)
        for i in range(n):
            r = orc.flow_pair(p, prevs[i], curs[i])
            blocks[i], subdirs[i], flows[i] = r["blocks"], r["subdirs"], r["flow"]
        out[name + "/params"] = np.array([getattr(p, f) for f in orc.PARAM_FIELDS], np.int32)
        out[name + "/prev"], out[name + "/cur"] = prevs, curs
        out[name + "/blocks"] = blocks.view(np.uint32)
        out[name + "/subdirs"] = subdirs
        out[name + "/flows"] = flows.view(np.uint8).reshape(n, 16)
    path = os.path.join(ROOT, "tests", "golden", "flow_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
