"""CPU tests pinning the oracle (no GPU).

The reference ships no golden vectors and its engine source is absent
(parity unpinned), so the oracle is pinned by
(1) analytic known answers -- a pure integer translation must come back
exactly, (2) hand-computed miniatures of each building block, and
(3) agreement with an independently written numpy restatement (npref.py)."""
import numpy as np
import pytest

import npref


def pdict(p):
    return {n: getattr(p, n) for n, _ in p._fields_}


# ---- (2) hand-computed miniatures ------------------------------------------

def test_compute_diff_hand(orc):
    img = np.zeros((8, 8), np.uint8)
    # 4x4 patch at (2,2): rows 10,20,30,40 constant along x
    for r in range(4):
        img[2 + r, :] = 10 * (r + 1)
    # vertical diffs: 3 row pairs x 4 cols x 10 = 120 ; horizontal 0
    assert orc.compute_diff(img, 0, 0, 8) == 120
    img2 = img.T.copy()
    assert orc.compute_diff(img2, 0, 0, 8) == 120
    # a single bright pixel inside the patch touches 4 neighbours
    img3 = np.zeros((8, 8), np.uint8)
    img3[3, 3] = 7
    assert orc.compute_diff(img3, 0, 0, 8) == 28
    # pixels outside the 4x4 patch do not count
    img4 = np.zeros((8, 8), np.uint8)
    img4[0, :] = 255
    img4[:, 7] = 255
    assert orc.compute_diff(img4, 0, 0, 8) == 0
    # 16x16 tile: patch at offset 6
    img5 = np.zeros((16, 16), np.uint8)
    img5[7, 7] = 5
    assert orc.compute_diff(img5, 0, 0, 16) == 20


def test_sad_hand(orc):
    a = np.full((16, 16), 10, np.uint8)
    b = np.full((16, 16), 13, np.uint8)
    assert orc.sad(a, 0, 0, b, 0, 0, 8) == 64 * 3
    assert orc.sad(a, 4, 4, b, 8, 8, 8) == 64 * 3
    assert orc.sad(a, 0, 0, b, 0, 0, 16) == 256 * 3
    b[0, 0] = 255
    assert orc.sad(a, 0, 0, b, 0, 0, 8) == 63 * 3 + 245
    assert orc.sad(a, 0, 0, b, 1, 0, 8) == 64 * 3


def test_subpixel_hand(orc):
    # horizontal ramp p(x) = 2x: right half-pixel = 2x+1, left = 2x-1, vertical = 2x
    b = np.tile((2 * np.arange(16)).astype(np.uint8), (16, 1))
    a = b.copy()
    acc = orc.subpixel(a, 4, 4, b, 4, 4, 8)
    # dirs 0 (right), 4 (left): |2x - (2x+-1)| = 1 per pixel; 2, 6 (down/up): 0
    assert acc[0] == 64 and acc[4] == 64 and acc[2] == 0 and acc[6] == 0
    # diagonals: avg of (s0=2x+1, s1=2x+1) = 2x+1 -> 64 ; t3 = avg(s3=2x-1, s4=2x-1) -> 64
    assert list(acc[[1, 3, 5, 7]]) == [64, 64, 64, 64]
    # flooring: p=(1,2) pairs -> (1+2)>>1 = 1
    c = np.zeros((12, 12), np.uint8)
    c[:, 0::2] = 1
    c[:, 1::2] = 2
    ref = np.ones((12, 12), np.uint8)
    acc = orc.subpixel(ref, 2, 2, c, 2, 2, 8)
    assert acc[0] == 0 and acc[4] == 0      # floor((1+2)/2) = 1 == ref everywhere
    assert acc[2] == 32 and acc[6] == 32    # vertical avg keeps 1,2,1,2 -> half the pixels differ by 1


def test_mean_pyramid_equalise_hand(orc):
    img = np.array([[0, 1], [2, 4]], np.uint8)
    assert orc.frame_mean(img) == 2                      # 7/4 = 1.75 -> 2
    assert orc.frame_mean(np.array([[1, 2]], np.uint8)) == 2  # 1.5 rounds half up
    assert orc.pyramid_down(img)[0, 0] == 2              # (7+2)>>2
    assert orc.pyramid_down(np.array([[1, 1], [1, 2]], np.uint8))[0, 0] == 1  # (5+2)>>2
    assert orc.pyramid_down(np.array([[1, 1], [2, 2]], np.uint8))[0, 0] == 2  # 1.5 -> 2
    e = orc.equalise(np.array([[0, 5, 250, 255]], np.uint8), 10)
    assert list(e[0]) == [10, 15, 255, 255]
    e = orc.equalise(np.array([[0, 5, 250, 255]], np.uint8), -7)
    assert list(e[0]) == [0, 0, 243, 248]


def test_grid_numbers(orc):
    # SURVEY.md section 8d block counts
    g = orc.grid(orc.default_params(640, 480))
    assert (g.x0, g.y0, g.step_x, g.nx, g.ny) == (4, 4, 8, 79, 59)
    g = orc.grid(orc.default_params(1280, 960, tile=16, search=8))
    assert (g.x0, g.step_x, g.nx, g.ny) == (8, 16, 79, 59)
    g = orc.grid(orc.default_params(64, 64))
    assert (g.nx, g.ny) == (7, 7)
    # published sparse grid on 64x64: 5,15,25,35,45
    g = orc.grid(orc.px4flow_params(64, 64))
    assert (g.x0, g.step_x, g.nx, g.ny) == (5, 10, 5, 5)
    g = orc.grid(orc.default_params(640, 480, pyramid_levels=2), 1)
    assert (g.nx, g.ny) == (39, 29)
    # half-pixel refinement needs one more pixel of margin
    g = orc.grid(orc.default_params(640, 480, subpixel=1))
    assert (g.x0, g.nx, g.ny) == (5, 78, 58)


def _mk_blocks(orc, entries):
    b = np.zeros(len(entries), orc.BLOCK_DTYPE)
    for k, (dx, dy, s) in enumerate(entries):
        b[k] = (dx, dy, s)
    return b


def test_reduce_hist_filter_hand(orc):
    p = orc.default_params(64, 64, min_valid=2)
    # 6 votes dx=1, 2 votes dx=2, 1 vote dx=-4 (outlier, outside the +-2-bin window)
    ent = [(1, 0, 100)] * 6 + [(2, 0, 100)] * 2 + [(-4, 0, 100)] + [(0, 0, 0xFFFF)] + [(3, 3, 5000)]
    f, px, py = orc.reduce(p, _mk_blocks(orc, ent), None, 4)
    assert f["count"] == 9 and f["flags"] & 1
    # bins: centre 9; dx=1 -> 11 (6), dx=2 -> 13 (2): window 9..13 -> (11*6+13*2)/8 = 11.5 -> (11.5-9)/2
    assert f["flow_x"] == np.float32(1.25) and f["flow_y"] == np.float32(0.0)
    assert f["quality"] == 9 * 255 // 11
    assert (px, py) == (3, 0)  # round-half-up of 2.5 half-pixels
    # plain average instead: (6*1 + 2*2 - 4)/9
    p.hist_filter = 0
    f, px, py = orc.reduce(p, _mk_blocks(orc, ent), None, 4)
    assert f["flow_x"] == np.float32(6.0) / np.float32(9.0)
    assert (px, py) == (1, 0)  # 12/9 half-pixels = 1.33 -> 1


def test_reduce_edges(orc):
    p = orc.default_params(64, 64, min_valid=10)
    # not enough accepted blocks: invalid, quality 0
    f, px, py = orc.reduce(p, _mk_blocks(orc, [(1, 1, 5)] * 10), None, 4)
    assert f["count"] == 10 and f["flags"] == 0 and f["quality"] == 0 and f["flow_x"] == 0
    f, _, _ = orc.reduce(p, _mk_blocks(orc, [(1, 1, 5)] * 11), None, 4)
    assert f["flags"] == 1 and f["quality"] == 255 and f["flow_x"] == 1.0
    # peak at the histogram ends uses the clipped windows
    f, px, _ = orc.reduce(p, _mk_blocks(orc, [(-4, 4, 5)] * 11), np.full(11, 4, np.uint8), 4)
    assert f["flow_x"] == np.float32(-4.5) and px == -9      # bin 0
    f, _, py = orc.reduce(p, _mk_blocks(orc, [(-4, 4, 5)] * 11), np.full(11, 2, np.uint8), 4)
    assert f["flow_y"] == np.float32(4.5) and py == 9        # last bin
    # negative predictor rounding is floor(x + 0.5)
    p.hist_filter = 0
    ent = [(-1, 0, 5)] * 6 + [(-2, 0, 5)] * 6  # mean -1.5 px = -3 half px
    _, px, _ = orc.reduce(p, _mk_blocks(orc, ent), None, 4)
    assert px == -3
    ent = [(-1, 0, 5)] * 9 + [(-2, 0, 5)] * 3  # -2.5 half px -> -2 (half up)
    _, px, _ = orc.reduce(p, _mk_blocks(orc, ent), None, 4)
    assert px == -2


# ---- (1) analytic known answers ---------------------------------------------

@pytest.mark.parametrize("shift", [(0, 0), (4, -4), (-3, 2), (1, 4), (-4, -4)])
def test_translation_known_answer_dense(orc, synth, shift):
    p = orc.default_params(64, 64)  # BASELINE config C1 geometry, dense grid
    prev, cur, _ = synth.make_pair(64, 64, 4, 3, shift=shift)
    r = orc.flow_pair(p, prev, cur)
    b = r["blocks"]
    assert (b["sad"] != 0xFFFF).all(), "blurred-noise texture passes the gradient gate"
    assert (b["dx"] == shift[0]).all() and (b["dy"] == shift[1]).all() and (b["sad"] == 0).all()
    f = r["flow"]
    assert f["flow_x"] == shift[0] and f["flow_y"] == shift[1]
    assert f["quality"] == 255 and f["count"] == 49 and f["flags"] == 1


@pytest.mark.parametrize("tile,search,size", [(8, 4, (96, 80)), (16, 8, (160, 128)), (8, 2, (48, 40))])
def test_translation_known_answer_shapes(orc, synth, tile, search, size):
    w, h = size
    p = orc.default_params(w, h, tile=tile, search=search, value_threshold=3000 * (tile // 8) ** 2)
    prev, cur, shift = synth.make_pair(w, h, search, 5)
    r = orc.flow_pair(p, prev, cur)
    ok = r["blocks"]["sad"] != 0xFFFF
    assert ok.all()
    assert (r["blocks"]["dx"] == shift[0]).all() and (r["blocks"]["dy"] == shift[1]).all()
    assert r["flow"]["flow_x"] == shift[0] and r["flow"]["flow_y"] == shift[1]


def test_translation_known_answer_px4flow_grid(orc, synth):
    p = orc.px4flow_params(64, 64)
    prev, cur, _ = synth.make_pair(64, 64, 4, 8, shift=(-2, 3))
    r = orc.flow_pair(p, prev, cur)
    assert len(r["blocks"]) == 25
    assert (r["blocks"]["dx"] == -2).all() and (r["blocks"]["dy"] == 3).all()
    assert (r["subdirs"] == 8).all(), "an exact integer match (SAD 0) cannot be improved"
    assert r["flow"]["flow_x"] == -2 and r["flow"]["flow_y"] == 3 and r["flow"]["quality"] == 255


def test_two_level_reaches_beyond_search(orc, synth):
    # shift of 9 px is outside +-4 at level 0 but 4.5 px at level 1
    p = orc.default_params(160, 128, pyramid_levels=2)
    prev, cur, _ = synth.make_pair(160, 128, 12, 21, shift=(9, -7))
    r = orc.flow_pair(p, prev, cur, want_l1=True)
    f = r["flow"]
    assert f["flags"] == 3 and abs(int(f["pred_x"]) - 9) <= 1 and abs(int(f["pred_y"]) + 7) <= 1
    ok = r["blocks"]["sad"] != 0xFFFF
    assert ok.sum() > 0.6 * ok.size
    assert (r["blocks"]["dx"][ok] == 9).all() and (r["blocks"]["dy"][ok] == -7).all()
    assert f["flow_x"] == 9 and f["flow_y"] == -7
    # one level cannot see it
    p1 = orc.default_params(160, 128)
    r1 = orc.flow_pair(p1, prev, cur)
    assert not ((r1["blocks"]["dx"] == 9) & (r1["blocks"]["sad"] == 0)).any()


def test_mean_equalisation_removes_exposure_step(orc, synth):
    prev, cur, _ = synth.make_pair(96, 80, 4, 2, shift=(2, -1), brightness=0)
    lo, hi = int(min(prev.min(), cur.min())), int(max(prev.max(), cur.max()))
    assert lo > 0 and hi < 230, "test needs headroom so the step does not saturate"
    cur_b = (cur.astype(np.int32) + 25).astype(np.uint8)
    on = orc.flow_pair(orc.default_params(96, 80, mean_subtract=1), prev, cur_b)
    assert (on["blocks"]["sad"] == 0).all() and on["flow"]["flow_x"] == 2 and on["flow"]["flow_y"] == -1
    off = orc.flow_pair(orc.default_params(96, 80, mean_subtract=0), prev, cur_b)
    assert (off["blocks"]["sad"] >= 900).all(), "without equalisation every SAD carries the step"


def test_flat_image_is_skipped(orc):
    p = orc.default_params(64, 64)
    flat = np.full((64, 64), 77, np.uint8)
    r = orc.flow_pair(p, flat, flat)
    assert (r["blocks"]["sad"] == 0xFFFF).all()
    assert r["flow"]["count"] == 0 and r["flow"]["quality"] == 0 and r["flow"]["flags"] == 0


def test_first_minimum_wins(orc):
    # vertical stripes of period 2: shifts dx=-4,-2,0,2,4 all give SAD 0 -> scan order picks (-4,-4)
    p = orc.default_params(64, 64)
    img = np.zeros((64, 64), np.uint8)
    img[:, 0::2] = 200
    r = orc.flow_pair(p, img, img)
    assert (r["blocks"]["dx"] == -4).all() and (r["blocks"]["dy"] == -4).all()
    assert (r["blocks"]["sad"] == 0).all()


def test_param_validation(orc):
    assert orc.lib.orc_params_check(orc.default_params(64, 64)) == 0
    for kw in (dict(tile=12), dict(search=0), dict(search=9), dict(pyramid_levels=3),
               dict(grid_mode=2), dict(feature_threshold=-1)):
        assert orc.lib.orc_params_check(orc.default_params(64, 64, **kw)) != 0, kw
    assert orc.lib.orc_params_check(orc.default_params(8, 64)) != 0
    assert orc.lib.orc_params_check(orc.default_params(65, 64, pyramid_levels=2)) != 0


# ---- (3) independent numpy restatement ---------------------------------------

CASES = [
    dict(width=64, height=64),
    dict(width=64, height=64, grid_mode=1, subpixel=1),
    dict(width=72, height=56, subpixel=1, hist_filter=0),
    dict(width=96, height=64, pyramid_levels=2, mean_subtract=1),
    dict(width=96, height=64, pyramid_levels=2, subpixel=1, grid_mode=1, num_blocks=4),
    dict(width=80, height=80, tile=16, search=8, value_threshold=12000),
    dict(width=50, height=46, search=3, mean_subtract=1),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_oracle_matches_numpy_restatement(orc, synth, case):
    kw = CASES[case]
    p = orc.default_params(**kw)
    rng = np.random.default_rng(100 + case)
    for trial in range(3):
        prev, cur, _ = synth.make_pair(p.width, p.height, p.search, 40 + 7 * case + trial,
                                       noise=3 * trial, brightness=(-6, 0, 11)[trial])
        if trial == 2:  # unrelated noise frames: ties, rejections, outliers
            cur = rng.integers(0, 256, cur.shape, dtype=np.uint8)
        r = orc.flow_pair(p, prev, cur)
        n = npref.flow_pair(pdict(p), prev, cur)
        recs = [(int(b["dx"]), int(b["dy"]), int(b["sad"])) for b in r["blocks"]]
        assert recs == n["recs"]
        assert list(r["subdirs"]) == n["subs"]
        f = r["flow"]
        assert f["flow_x"] == n["flow_x"] and f["flow_y"] == n["flow_y"]
        assert f["count"] == n["count"] and f["quality"] == n["quality"]
        assert (f["pred_x"], f["pred_y"]) == (n["pred_x"], n["pred_y"])
        assert bool(f["flags"] & 1) == n["valid"] and bool(f["flags"] & 2) == n["pred_valid"]


def test_building_blocks_match_numpy(orc):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (40, 48), dtype=np.uint8)
    b = rng.integers(0, 256, (40, 48), dtype=np.uint8)
    for B in (8, 16):
        for _ in range(20):
            ax, ay = rng.integers(1, 48 - B - 1), rng.integers(1, 40 - B - 1)
            bx, by = rng.integers(1, 48 - B - 1), rng.integers(1, 40 - B - 1)
            assert orc.sad(a, ax, ay, b, bx, by, B) == npref.sad(a, ax, ay, b, bx, by, B)
            assert orc.compute_diff(a, ax, ay, B) == npref.compute_diff(a, ax, ay, B)
            assert list(orc.subpixel(a, ax, ay, b, bx, by, B)) == npref.subpixel(a, ax, ay, b, bx, by, B)
    assert orc.frame_mean(a) == npref.frame_mean(a)
    assert np.array_equal(orc.pyramid_down(a), npref.pyramid_down(a))
    for d in (-300, -17, 0, 9, 300):
        assert np.array_equal(orc.equalise(a, d), npref.equalise(a, d))


def test_batch_equals_single(orc, synth):
    p = orc.default_params(64, 64)
    prevs, curs, _ = synth.make_batch(64, 64, 5, 4, 70)
    blocks, flows, used = orc.flow_batch(p, prevs, curs, threads=2)
    assert used >= 1
    for i in range(5):
        r = orc.flow_pair(p, prevs[i], curs[i])
        assert blocks[i].tobytes() == r["blocks"].tobytes() and flows[i].tobytes() == r["flow"].tobytes()


# ---- facade semantics (calcFlow) ---------------------------------------------

def test_px4_rate_limit_semantics(orc, synth):
    """Contract visible at /root/reference/src/mainloop.cpp:322-331: first call
    returns 0, then negative until 1/output_rate elapsed, then the integrated
    flow as an angle, dt_us = time since the last publication."""
    p = orc.px4flow_params(64, 64)
    fx = fy = 216.0
    o = orc.Px4(p, fx, fy, 15)
    frames, steps = synth.make_sequence(64, 64, 14, reach=4, seed=1, max_step=2)
    t, period = 0, 13333  # 75 fps
    out = []
    for k in range(14):
        out.append(o.calc_flow(frames[k], t))
        t += period
    assert out[0][0] == 0
    # 1e6/15 = 66666.7 us -> first publication when t - 0 > 66666 i.e. frame index 6 (t=79998)
    assert [r[0] < 0 for r in out[1:6]] == [True] * 5
    q, dt, ax, ay = out[6]
    assert q == 255 and dt == 6 * period
    sx, sy = steps[:6, 0].sum(), steps[:6, 1].sum()
    assert ax == pytest.approx(np.arctan2(np.float32(sx), np.float32(fx)), abs=1e-7)
    assert ay == pytest.approx(np.arctan2(np.float32(sy), np.float32(fy)), abs=1e-7)
    # next publication integrates only the frames since
    # 5 periods = 66665 us is not > 66666.7 us, so the next one is 6 frames later
    assert [r[0] < 0 for r in out[7:12]] == [True] * 5
    assert out[12][0] == 255 and out[12][1] == 6 * period
    assert out[12][2] == pytest.approx(np.arctan2(np.float32(steps[6:12, 0].sum()), np.float32(fx)), abs=1e-7)
    assert out[13][0] < 0


def test_px4_unlimited_rate_and_wraparound(orc, synth):
    p = orc.px4flow_params(64, 64)
    o = orc.Px4(p, 200.0, 200.0, 0)  # output_rate <= 0: publish every frame
    frames, steps = synth.make_sequence(64, 64, 4, reach=4, seed=2, max_step=3)
    t0 = (1 << 32) - 20000  # the u32 microsecond clock wraps (mainloop.cpp:313-315)
    r0 = o.calc_flow(frames[0], t0)
    r1 = o.calc_flow(frames[1], t0 + 13000)
    r2 = o.calc_flow(frames[2], t0 + 26000)   # wrapped
    assert r0[0] == 0 and r1[0] == 255 and r2[0] == 255
    assert r2[1] == 13000
    assert r2[2] == pytest.approx(np.arctan2(np.float32(steps[1, 0]), np.float32(200.0)), abs=1e-7)


def test_fast_sad_equals_byte_loop(orc, synth):
    """bench.py times the oracle with the host's SAD instruction (SSE2 psadbw); the checker uses
    the byte loop.  Both must give the same records and flows."""
    if not orc.fast_sad_available():
        pytest.skip("no SSE2 on this host")
    rng = np.random.default_rng(5)
    try:
        for kw, w, h in ((dict(), 128, 96), (dict(tile=16, search=8, value_threshold=12000), 160, 128),
                         (dict(pyramid_levels=2, mean_subtract=1, subpixel=1), 160, 128),
                         (dict(grid_mode=1, subpixel=1, num_blocks=5), 64, 64)):
            p = orc.default_params(w, h, **kw)
            prevs, curs, _ = synth.make_batch(w, h, 6, 4, 77, noise=6, brightness=5)
            curs[5] = rng.integers(0, 256, curs[5].shape, dtype=np.uint8)
            out = []
            for fast in (False, True):
                orc.set_fast_sad(fast)
                out.append([orc.flow_pair(p, prevs[i], curs[i]) for i in range(6)])
            for a, b in zip(*out):
                assert a["blocks"].tobytes() == b["blocks"].tobytes()
                assert a["flow"].tobytes() == b["flow"].tobytes()
        a = rng.integers(0, 256, (40, 48), dtype=np.uint8)
        b = rng.integers(0, 256, (40, 48), dtype=np.uint8)
        for tile in (8, 16):
            for _ in range(50):
                ax, ay, bx, by = (int(v) for v in rng.integers(0, 24, 4))
                orc.set_fast_sad(False)
                ref = orc.sad(a, ax, ay, b, bx, by, tile)
                orc.set_fast_sad(True)
                assert orc.sad(a, ax, ay, b, bx, by, tile) == ref
    finally:
        orc.set_fast_sad(False)
