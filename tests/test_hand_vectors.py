"""Hand-derived end-to-end vectors: tiny frames built from a closed formula whose block
records, histogram votes, +-2-bin weighted mean and quality follow from the construction and
are written down here as literals -- no expectation in this file is produced by oracle code.
The CPU test holds the oracle to them, the GPU test holds the HIP path to them, so the two
are pinned to something other than each other (upstream PX4 source stays unavailable: this
pins the published algorithm as DESIGN.md section 2 states it, nothing more)."""
import numpy as np
import pytest


def texture(h, w):
    """Deterministic, strongly non-periodic grey pattern (every 8x8 tile is unique)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    return ((x * 37 + y * 91 + x * y * 13 + (x * x) * 7 + (y * y * y) * 3) % 251).astype(np.uint8)


def case_uniform_shift():
    """24x24, dense grid, S=4, B=8: origin 4, step 8 -> 2x2 blocks at x,y in {4, 12}.
    cur[y][x] = prev[y+1][x-2] wherever both exist: every tile of prev at (i, j) reappears
    in cur at (i+2, j-1).  Hence every record is (dx, dy, sad) = (2, -1, 0).
    Votes: x bin = 2*dx + (2R+1) = 4 + 9 = 13, y bin = -2 + 9 = 7, four votes each; a lone
    peak -> weighted mean = the bin itself -> flow = ((13 - 9)/2, (7 - 9)/2) = (2.0, -1.0);
    count 4 > min_valid 0; quality = 4*255/4 = 255."""
    prev = texture(24, 24)
    cur = np.full((24, 24), 3, np.uint8)
    cur[0:23, 2:24] = prev[1:24, 0:22]
    params = dict(width=24, height=24, min_valid=0)
    blocks = [(2, -1, 0)] * 4
    flow = dict(flow_x=2.0, flow_y=-1.0, count=4, quality=255, flags=1)
    return prev, cur, params, blocks, flow


def case_two_motions_and_a_gated_block():
    """40x24: 4x2 blocks at x in {4, 12, 20, 28}, y in {4, 12}.  The left two block columns
    move by dx=+1, the right two by dx=+2, dy=0 everywhere (columns of cur: 5..20 <- prev
    4..19, 22..37 <- prev 20..35; nothing overlaps).  Block (x=4, y=12) of prev is painted
    flat: its 4x4 gradient sum is 0 < 30 -> gate -> skipped (sad 0xFFFF, no vote).
    Votes x: bin 2*1+9 = 11 three times, bin 2*2+9 = 13 four times; y: bin 9 seven times.
    Peak x = bin 13 (4 votes; first maximum).  Window 11..15 (peak +-2):
    sum k*h = 11*3 + 13*4 = 85, sum h = 7 -> mean bin 85/7 -> flow_x = (85/7 - 9)/2 in
    float32 arithmetic; flow_y = (9 - 9)/2 = 0.  count 7; quality = floor(7*255/8) = 223."""
    prev = texture(24, 40)
    prev[12:20, 4:12] = 77
    cur = np.full((24, 40), 200, np.uint8)
    cur[:, 5:21] = prev[:, 4:20]
    cur[:, 22:38] = prev[:, 20:36]
    params = dict(width=40, height=24, min_valid=0)
    blocks = [(1, 0, 0), (1, 0, 0), (2, 0, 0), (2, 0, 0),
              (0, 0, 0xFFFF), (1, 0, 0), (2, 0, 0), (2, 0, 0)]
    fx = (np.float32(85) / np.float32(7) - np.float32(9)) / np.float32(2)
    flow = dict(flow_x=float(fx), flow_y=0.0, count=7, quality=223, flags=1)
    return prev, cur, params, blocks, flow


def case_plain_average_and_min_valid():
    """The frames of the previous case without the histogram filter: flow_x = sum(dx)/count
    = (1+1+2+2+1+2+2)/7 = 11/7 (as (sum of 2*dx * 0.5)/count in float32), flow_y = 0; and
    with min_valid = 7 the same seven votes are NOT enough (count must exceed min_valid):
    flow 0, quality 0, flags 0, count still 7."""
    prev, cur, params, blocks, _ = case_two_motions_and_a_gated_block()
    fx = np.float32(22 * 0.5) / np.float32(7)
    plain = dict(params, hist_filter=0), dict(flow_x=float(fx), flow_y=0.0, count=7, quality=223, flags=1)
    starved = dict(params, min_valid=7), dict(flow_x=0.0, flow_y=0.0, count=7, quality=0, flags=0)
    return prev, cur, blocks, plain, starved


def case_exposure_step():
    """24x24, no motion, cur = prev + 4 grey levels (the texture stays below 251: nothing saturates).
    mean = floor((sum + N/2)/N): mean(cur) = mean(prev) + 4 exactly, so delta = -4 and the equalised
    cur' = clamp(cur - 4) = prev: every record (0, 0, 0), where the raw frames would give SAD
    64*4 = 256 at the same place.  Votes: bins 9 / 9, four each -> flow (0, 0), quality 255."""
    prev = texture(24, 24)
    cur = (prev.astype(np.int32) + 4).astype(np.uint8)
    params = dict(width=24, height=24, min_valid=0, mean_subtract=1)
    flow = dict(flow_x=0.0, flow_y=0.0, count=4, quality=255, flags=1)
    return prev, cur, params, [(0, 0, 0)] * 4, flow


def case_two_levels_reach_six_pixels():
    """48x48, two levels, cur[y][x] = prev[y+2][x-6]: a shift of (+6, -2), beyond the +-4 of one level.
    Level 1 = 2x2 box means; an EVEN shift commutes with the box filter, so cur1[y][x] =
    prev1[y+1][x-3] and the four level-1 blocks (24x24, origin 4, step 8) match at (3, -1) with SAD 0:
    level-1 flow (3, -1) in level-1 pixels, predictor = (6, -2) level-0 pixels, valid.
    Level 0: 5x5 blocks at 4, 12, .., 36.  Under the predictor the window of a block at (i, j) spans
    x = i+2 .. i+17, y = j-6 .. j+9: inside the 48x48 frame for i <= 30 and j >= 6, i.e. columns
    4..28 and rows 12..36 -- 16 blocks, each matching at the window's centre: (6, -2, 0); the other nine
    are skipped (0, 0, 0xFFFF).  Histogram range R = 3S+1 = 13, centre 27: x bin 2*6+27 = 39, y bin
    -4+27 = 23, 16 votes each, lone peaks -> flow (6.0, -2.0); count 16; quality = floor(16*255/25) =
    163; flags = flow valid | predictor valid = 3."""
    prev = texture(48, 48)
    cur = np.full((48, 48), 3, np.uint8)
    cur[0:46, 6:48] = prev[2:48, 0:42]
    params = dict(width=48, height=48, min_valid=0, pyramid_levels=2)
    blocks = []
    for j in (4, 12, 20, 28, 36):
        for i in (4, 12, 20, 28, 36):
            blocks.append((6, -2, 0) if i <= 30 and j >= 6 else (0, 0, 0xFFFF))
    flow = dict(flow_x=6.0, flow_y=-2.0, count=16, quality=163, flags=3)
    return prev, cur, params, blocks, flow


def case_tile16_uniform_shift():
    """64x64, 16x16 tiles, S=8 (BASELINE configs[4]'s kernel class): dense grid origin 8, step 16,
    nx = ny = floor((64 - 16)/16) = 3 -> 9 blocks at x,y in {8, 24, 40}.
    cur[y][x] = prev[y-5][x+7] wherever both exist (y >= 5, x <= 56): the tile of prev at (i, j)
    reappears in cur at (i-7, j+5) -- x = 1..49 and y = 13..60 for the nine tiles, all inside the
    defined region and inside every block's window [i-8, i+23] x [j-8, j+23] -- so every record is
    (dx, dy, sad) = (-7, +5, 0).  R = S = 8, centre 2R+1 = 17: x bin = -14 + 17 = 3, y bin =
    10 + 17 = 27, nine votes each, lone peaks -> flow (-7.0, 5.0); count 9 > min_valid 0;
    quality = 9*255/9 = 255."""
    prev = texture(64, 64)
    cur = np.full((64, 64), 3, np.uint8)
    cur[5:64, 0:57] = prev[0:59, 7:64]
    params = dict(width=64, height=64, tile=16, search=8, min_valid=0)
    flow = dict(flow_x=-7.0, flow_y=5.0, count=9, quality=255, flags=1)
    return prev, cur, params, [(-7, 5, 0)] * 9, flow


def case_tile16_two_motions_and_a_gated_block():
    """96x64, 16x16 tiles, S=8: 5x3 blocks at x in {8, 24, 40, 56, 72}, y in {8, 24, 40}.  Block columns
    8..55 (three columns) move by dx = -3, columns 56..87 (two columns) by dx = -2, dy = 0 everywhere
    (columns of cur: 5..52 <- prev 8..55, 54..85 <- prev 56..87; the destination ranges do not overlap
    and every tile lands inside its own window).  Block (x=24, y=24) of prev is painted flat: 4x4
    gradient sum 0 < 30 -> gated (sad 0xFFFF, no vote).
    Votes x: bin 2*(-3)+17 = 11 eight times (nine tiles minus the gated one), bin 2*(-2)+17 = 13 six
    times; y: bin 17 fourteen times.  Peak x = bin 11 (8 votes); window 9..13: sum k*h = 11*8 + 13*6 =
    166, sum h = 14 -> flow_x = (166/14 - 17)/2 in float32 arithmetic; flow_y = 0; count 14;
    quality = floor(14*255/15) = 238."""
    prev = texture(64, 96)
    prev[24:40, 24:40] = 77
    cur = np.full((64, 96), 200, np.uint8)
    cur[:, 5:53] = prev[:, 8:56]
    cur[:, 54:86] = prev[:, 56:88]
    params = dict(width=96, height=64, tile=16, search=8, min_valid=0)
    row = [(-3, 0, 0)] * 3 + [(-2, 0, 0)] * 2
    blocks = row + [(-3, 0, 0), (0, 0, 0xFFFF), (-3, 0, 0), (-2, 0, 0), (-2, 0, 0)] + row
    fx = (np.float32(166) / np.float32(14) - np.float32(17)) / np.float32(2)
    flow = dict(flow_x=float(fx), flow_y=0.0, count=14, quality=238, flags=1)
    return prev, cur, params, blocks, flow


def case_tile16_two_levels_reach_twelve_pixels():
    """96x96, 16x16 tiles, S=8, two levels, cur[y][x] = prev[y+6][x-12]: a shift of (+12, -6), beyond the
    +-8 of one level.  An even shift commutes with the 2x2 box filter: cur1[y][x] = prev1[y+3][x-6], so
    the four level-1 blocks (48x48, origin 8, step 16: x,y in {8, 24}) match at (6, -3) with SAD 0
    (tiles land at x = 14..45, y = 5..36: inside the defined region x >= 6, y < 45 and the windows):
    level-1 flow (6, -3), predictor (12, -6) level-0 pixels, valid.
    Level 0: 5x5 blocks at 8, 24, .., 72.  Under the predictor the window of a block at (i, j) spans
    x = i+4 .. i+35, y = j-14 .. j+17: inside the frame for i <= 60 and j >= 14, i.e. columns 8..56 and
    rows 24..72 -- 16 blocks, each matching at the window's centre: (12, -6, 0); the other nine are
    skipped (0, 0, 0xFFFF).  R = 3S+1 = 25, centre 51: x bin 24+51 = 75, y bin -12+51 = 39, 16 votes
    each, lone peaks -> flow (12.0, -6.0); count 16; quality = floor(16*255/25) = 163; flags = flow
    valid | predictor valid = 3."""
    prev = texture(96, 96)
    cur = np.full((96, 96), 3, np.uint8)
    cur[0:90, 12:96] = prev[6:96, 0:84]
    params = dict(width=96, height=96, tile=16, search=8, min_valid=0, pyramid_levels=2)
    blocks = []
    for j in (8, 24, 40, 56, 72):
        for i in (8, 24, 40, 56, 72):
            blocks.append((12, -6, 0) if i <= 60 and j >= 14 else (0, 0, 0xFFFF))
    flow = dict(flow_x=12.0, flow_y=-6.0, count=16, quality=163, flags=3)
    return prev, cur, params, blocks, flow


def case_half_pixel():
    """40x40 with half-pixel refinement (dense grid at origin S+1 = 5, step 8: 3x3 blocks).
    prev[y][x] = (cur[y][x] + cur[y][x+1]) >> 1: prev is cur sampled half a pixel to the right.  Whatever
    integer match a block finds -- dx = 0 or dx = +1, the texture decides, so the records are not
    written down here --, the half-pixel image between the two, (a+b)>>1 of horizontally adjacent cur
    pixels, IS the tile: direction +x from dx = 0 or direction -x from dx = +1 has SAD 0 and wins.
    Either way the vote is 2*dx +- 1 = +1 half-pixel: bin 1 + 9 = 10 nine times, y bin 9 ->
    flow = (0.5, 0.0); count 9; quality 255.  (A smooth texture here: on the rough one of the other
    cases a half-pixel-misaligned tile loses the integer search to some unrelated place.)"""
    y, x = np.mgrid[0:40, 0:41].astype(np.float64)   # smooth and non-periodic: the true match must win the integer search
    wide = np.rint(128 + 55 * np.sin(0.37 * x + 0.11 * y) + 45 * np.sin(0.083 * x - 0.29 * y + 0.004 * x * y)
                   + 20 * np.sin(0.9 * y + 0.05 * x * x / 40)).astype(np.uint8)
    cur = np.ascontiguousarray(wide[:, :40])
    prev = ((wide[:, :40].astype(np.int32) + wide[:, 1:41]) >> 1).astype(np.uint8)
    params = dict(width=40, height=40, min_valid=0, subpixel=1)
    flow = dict(flow_x=0.5, flow_y=0.0, count=9, quality=255, flags=1)
    return prev, cur, params, flow


def check_half_pixel(blocks_got, subdirs_got, flow_got, flow):
    assert len(blocks_got) == 9
    for b, d in zip(blocks_got, subdirs_got):
        assert (int(b["dx"]), int(d)) in ((0, 0), (1, 4)) and int(b["dy"]) == 0, (b, d)   # +x from 0 or -x from +1
    check(blocks_got, flow_got, blocks_got.tolist(), flow)


def check(blocks_got, flow_got, blocks, flow):
    want = np.array(blocks, dtype=[("dx", "i1"), ("dy", "i1"), ("sad", "<u2")])
    assert blocks_got.tobytes() == want.tobytes(), (blocks_got, want)
    assert np.float32(flow_got["flow_x"]).tobytes() == np.float32(flow["flow_x"]).tobytes(), flow_got
    assert np.float32(flow_got["flow_y"]).tobytes() == np.float32(flow["flow_y"]).tobytes(), flow_got
    for k in ("count", "quality", "flags"):
        assert int(flow_got[k]) == flow[k], (k, flow_got)


def all_cases():
    p, c, params, blocks, flow = case_uniform_shift()
    yield p, c, params, blocks, flow
    p, c, params, blocks, flow = case_two_motions_and_a_gated_block()
    yield p, c, params, blocks, flow
    p, c, blocks, plain, starved = case_plain_average_and_min_valid()
    yield p, c, plain[0], blocks, plain[1]
    yield p, c, starved[0], blocks, starved[1]
    yield case_exposure_step()
    yield case_two_levels_reach_six_pixels()
    yield case_tile16_uniform_shift()
    yield case_tile16_two_motions_and_a_gated_block()
    yield case_tile16_two_levels_reach_twelve_pixels()


def test_oracle_reproduces_the_hand_vectors(orc):
    for prev, cur, params, blocks, flow in all_cases():
        p = orc.default_params(**params)
        r = orc.flow_pair(p, prev, cur)
        check(r["blocks"], r["flow"], blocks, flow)


def test_oracle_reproduces_the_half_pixel_vector(orc):
    prev, cur, params, flow = case_half_pixel()
    r = orc.flow_pair(orc.default_params(**params), prev, cur)
    check_half_pixel(r["blocks"], r["subdirs"], r["flow"], flow)


@pytest.mark.gpu
def test_hip_path_reproduces_the_half_pixel_vector(aof, gpu_device):
    prev, cur, params, flow = case_half_pixel()
    p = aof.default_params(40, 40, **{k: v for k, v in params.items() if k not in ("width", "height")})
    for generic, split in ((False, False), (True, False), (False, True)):
        eng = aof.FlowEngine(p, 0)
        eng.force_generic(generic)
        eng.set_split_coarse(split)
        got_blocks, got_subdirs, got_flow = eng.flow_pair_host(prev, cur)
        check_half_pixel(got_blocks, got_subdirs, got_flow, flow)
        eng.close()


@pytest.mark.gpu
def test_hip_path_reproduces_the_hand_vectors(aof, gpu_device):
    for prev, cur, params, blocks, flow in all_cases():
        w, h = params["width"], params["height"]
        p = aof.default_params(w, h, **{k: v for k, v in params.items() if k not in ("width", "height")})
        # default / generic / separate kernels / exact pruned search (16x16: k_search_tile16<PRUNE>)
        for generic, split, mode in ((False, False, aof.SEARCH_EXHAUSTIVE), (True, False, aof.SEARCH_EXHAUSTIVE),
                                     (False, True, aof.SEARCH_EXHAUSTIVE), (False, False, aof.SEARCH_PRUNED)):
            eng = aof.FlowEngine(p, 0)
            eng.force_generic(generic)
            eng.set_split_coarse(split)
            eng.set_search_mode(mode)
            if params.get("tile") == 16 and not generic:
                assert eng.variant == "tile16_lds"
            got_blocks, _, got_flow = eng.flow_pair_host(prev, cur)
            check(got_blocks, got_flow, blocks, flow)
            eng.close()
