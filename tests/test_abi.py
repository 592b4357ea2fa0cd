"""CPU tests of the C ABI library: it loads, exports every symbol that
include/aof.h declares, and its host-only entry points (parameters, grids,
workspace layout) behave -- no compute calls, no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "aof.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aof_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(aof):
    names = declared_symbols()
    assert len(names) >= 19
    lib = ctypes.CDLL(aof.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/aof.h but not exported"
    assert set(names) == set(aof.EXPORTS), "python binding and header disagree"
    assert aof.lib.aof_version() == 102


def test_struct_layouts_match_header(aof, orc):
    assert ctypes.sizeof(aof.Params) == 13 * 4
    assert ctypes.sizeof(aof.SearchStats) == 3 * 8 + 2 * 4   # aof_search_stats
    assert aof.BLOCK_DTYPE.itemsize == 4 and aof.FLOW_DTYPE.itemsize == 16
    assert aof.BLOCK_DTYPE == orc.BLOCK_DTYPE and aof.FLOW_DTYPE == orc.FLOW_DTYPE
    assert [n for n, _ in aof.Params._fields_] == list(orc.PARAM_FIELDS)


def test_default_params_are_the_baseline_config(aof):
    p = aof.default_params(640, 480)
    assert (p.tile, p.search, p.grid_mode, p.feature_threshold, p.value_threshold) == (8, 4, 0, 30, 3000)
    assert (p.subpixel, p.hist_filter, p.pyramid_levels, p.mean_subtract, p.min_valid) == (0, 1, 1, 0, 10)
    assert aof.grid(p, 0) == (4, 4, 8, 8, 79, 59)
    # 2*W*H + 4*nx*ny + 16
    assert aof.algorithmic_bytes(p) == 633060
    assert aof.abs_diffs(p) == 24162624
    p5 = aof.default_params(1280, 960, tile=16, search=8)
    assert aof.grid(p5, 0) == (8, 8, 16, 16, 79, 59)
    assert aof.algorithmic_bytes(p5) == 2476260 and aof.abs_diffs(p5) == 344839424
    p1 = aof.default_params(64, 64)
    assert aof.algorithmic_bytes(p1) == 8404
    px = aof.px4flow_params(64, 64)
    assert (px.grid_mode, px.subpixel, px.num_blocks) == (1, 1, 5)
    assert aof.grid(px, 0) == (5, 5, 10, 10, 5, 5)


def test_param_validation_matches_oracle(aof, orc):
    cases = [dict(), dict(tile=16, search=8), dict(tile=12), dict(search=0), dict(search=9),
             dict(pyramid_levels=2), dict(pyramid_levels=3), dict(grid_mode=1), dict(grid_mode=5),
             dict(feature_threshold=-3), dict(subpixel=1)]
    for size in ((64, 64), (65, 63), (16, 16), (24, 200), (640, 480)):
        for kw in cases:
            pa = aof.default_params(*size, **kw)
            po = orc.default_params(*size, **kw)
            assert (aof.check_params(pa) == 0) == (orc.lib.orc_params_check(ctypes.byref(po)) == 0), (size, kw)
            if aof.check_params(pa) == 0:
                for level in range(pa.pyramid_levels):
                    g = orc.grid(po, level)
                    assert aof.grid(pa, level) == (g.x0, g.y0, g.step_x, g.step_y, g.nx, g.ny)
    assert aof.check_params(aof.default_params(8192, 4096)) != 0  # > 2^24 px: u32 sums


def test_grid_level_out_of_range(aof):
    p = aof.default_params(64, 64)
    with pytest.raises(aof.AofError):
        aof.grid(p, 1)


def test_workspace_layout(aof):
    p = aof.default_params(640, 480, pyramid_levels=2, mean_subtract=1, subpixel=1)
    L = aof.workspace_layout(p, 10)
    offs = [L.sums, L.l1_prev, L.l1_cur, L.l1_blocks, L.l1_subdirs, L.l1_flows, L.l0_blocks, L.l0_subdirs]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert L.l1_cur - L.l1_prev >= 10 * 320 * 240
    nb0 = aof.grid(p, 0)[4] * aof.grid(p, 0)[5]
    assert L.total_bytes >= L.l0_subdirs + 10 * nb0
    L1 = aof.workspace_layout(aof.default_params(640, 480), 10)
    assert L1.total_bytes < L.total_bytes
    assert aof.workspace_layout(aof.default_params(64, 64), 0).total_bytes >= 256


def test_workspace_layout_reserves_chunk_histograms_for_large_grids(aof):
    """Grids of more than 8 192 blocks are reduced in two steps (one workgroup per chunk of 4 096 records
    votes into its own histogram, then one per pair sums them): the layout reserves those per-chunk
    histograms, and nothing for grids one workgroup reads alone."""
    for levels in (1, 2):
        big = aof.default_params(1024, 768, pyramid_levels=levels)        # 127 x 95 = 12 065 blocks
        nb = aof.grid(big, 0)[4] * aof.grid(big, 0)[5]
        assert nb > 8192
        chunks = (nb + 4095) // 4096
        bins = 2 * (2 * (4 if levels == 1 else 13) + 1) + 1
        for n in (1, 2, 9):
            L = aof.workspace_layout(big, n)
            size = (L.l1_hist if levels == 2 else L.total_bytes) - L.l0_hist   # level 0's region
            assert size >= n * chunks * 2 * bins * 4
    vga = aof.workspace_layout(aof.default_params(640, 480), 2)
    assert vga.total_bytes == vga.l0_hist or vga.total_bytes - vga.l0_hist < 256


def test_strerror(aof):
    assert aof.lib.aof_strerror(0) == b"ok"
    assert b"gfx950" in aof.lib.aof_strerror(-19)


def test_no_cpu_fallback_without_gpu(aof):
    """On a machine without a GPU the engine must refuse, not silently compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(aof.AofError) as e:
        aof.FlowEngine(aof.default_params(64, 64))
    assert e.value.code == -19  # -ENODEV


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing shipped may import, link or call it."""
    pkg = os.path.join(ROOT, "aero-optical-flow_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".txt", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"liboracle|pyoracle|aof_oracle|orc_flow|from oracle|import oracle", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    hdr = open(os.path.join(ROOT, "include", "aof.h")).read()
    assert "oracle" not in hdr.lower()


def test_synth_pair_is_a_pure_translation(synth):
    prev, cur, (dx, dy) = synth.make_pair(64, 48, 4, 9)
    ys, xs = np.mgrid[8:40, 8:56]
    assert np.array_equal(cur[ys, xs], prev[ys - dy, xs - dx])
    a = synth.make_pair(64, 48, 4, 9)
    assert np.array_equal(a[0], prev) and np.array_equal(a[1], cur), "deterministic per index"
    b = synth.make_pair(64, 48, 4, 10)
    assert not np.array_equal(b[0], prev)
    frames, steps = synth.make_sequence(64, 48, 5, 4, seed=3, max_step=2)
    for k in range(4):
        sx, sy = steps[k]
        assert np.array_equal(frames[k + 1][ys, xs], frames[k][ys - sy, xs - sx])


def test_facade_without_gpu_never_publishes(aof):
    """No CPU fallback in the facade either: without a gfx950 device the classes still
    construct (the reference cannot handle a failing constructor, mainloop.cpp:423-428),
    say why, and calcFlow() returns -1 for every frame so nothing is published."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    for cls in (aof.OpticalFlowPX4, aof.OpticalFlowOpenCV):
        flow = cls(216.0, 216.0, 15, 64, 64)
        assert "no usable gfx950 device" in flow.lastError()
        assert (flow.getImageWidth(), flow.getImageHeight()) == (64, 64)
        img = np.zeros((64, 64), np.uint8)
        t = 0
        for _ in range(12):
            assert flow.calcFlow(img, t)[0] == -1
            t += 13333
        flow.close()
