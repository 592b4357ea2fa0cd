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


def test_oracle_reproduces_the_hand_vectors(orc):
    for prev, cur, params, blocks, flow in all_cases():
        p = orc.default_params(**params)
        r = orc.flow_pair(p, prev, cur)
        check(r["blocks"], r["flow"], blocks, flow)


@pytest.mark.gpu
def test_hip_path_reproduces_the_hand_vectors(aof, gpu_device):
    for prev, cur, params, blocks, flow in all_cases():
        w, h = params["width"], params["height"]
        p = aof.default_params(w, h, **{k: v for k, v in params.items() if k not in ("width", "height")})
        for generic in (False, True):
            eng = aof.FlowEngine(p, 0)
            eng.force_generic(generic)
            got_blocks, _, got_flow = eng.flow_pair_host(prev, cur)
            check(got_blocks, got_flow, blocks, flow)
            eng.close()
