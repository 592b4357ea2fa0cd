"""The one transcendental of the path: pixel flow -> angular flow, atan2(flow_px, focal_px), as a FIXED
sequence of IEEE double operations (include/aof_math.h) so that the C++ facade, the oracle's restatement and
the device pipeline agree bit for bit.  CPU tests: the product's function (through the C ABI) against the
oracle's own restatement (bit-identical) and against math.atan2 (at most one float ulp, exact on the axes)."""
import math

import numpy as np


def ulp_distance(a, b):
    ia = np.float32(a).view(np.int32).astype(np.int64)
    ib = np.float32(b).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2 ** 31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2 ** 31) - ib, ib)
    return int(abs(int(ia) - int(ib)))


def test_angle_matches_the_oracle_restatement_and_libm_within_one_ulp(aof, orc):
    rng = np.random.default_rng(17)
    flows = np.concatenate([rng.normal(0, 3, 4000), rng.uniform(-40, 40, 2000), [0.0, -0.0, 0.5, -0.5, 4.5, 13.5, 1e-6, -1e-6,
                                                                                 216.6677, -216.6677, 89.747574, 1e-30]]).astype(np.float32)
    focals = np.concatenate([np.full(3000, 216.6677), np.full(3000, 216.2457), rng.uniform(20, 2000, 12)]).astype(np.float32)
    worst = 0
    for f, fo in zip(flows, focals):
        got = np.float32(aof.flow_angle(float(f), float(fo)))
        assert got.tobytes() == np.float32(orc.angle(float(f), float(fo))).tobytes(), (f, fo)
        want = np.float32(math.atan2(float(f), float(fo)))     # double atan2, rounded once
        worst = max(worst, ulp_distance(got, want))
    assert worst <= 1, worst


def test_angle_covers_all_quadrants_and_the_special_cases(aof, orc):
    pi = math.pi
    cases = [(0.0, 1.0, 0.0), (1.0, 0.0, pi / 2), (-1.0, 0.0, -pi / 2), (0.0, -1.0, pi), (-0.0, -1.0, -pi),
             (1.0, 1.0, pi / 4), (1.0, -1.0, 3 * pi / 4), (-1.0, -1.0, -3 * pi / 4), (-1.0, 1.0, -pi / 4),
             (0.0, 0.0, 0.0), (-0.0, 0.0, -0.0), (0.0, -0.0, pi), (float("inf"), 1.0, pi / 2),
             (1.0, float("inf"), 0.0), (float("inf"), float("inf"), pi / 4), (2.0, 5.0, math.atan2(2, 5)),
             (5.0, 2.0, math.atan2(5, 2)), (-3.0, -7.0, math.atan2(-3, -7)), (0.41421354, 1.0, math.atan2(0.41421354, 1.0)),
             (0.4142136, 1.0, math.atan2(0.4142136, 1.0))]
    for y, x, want in cases:
        for fn in (aof.flow_angle, orc.angle):
            got = np.float32(fn(y, x))
            assert ulp_distance(got, np.float32(want)) <= 1, (y, x, got, want)
            assert math.copysign(1.0, float(got)) == math.copysign(1.0, want) or want != 0.0, (y, x)
    assert math.isnan(aof.flow_angle(float("nan"), 1.0)) and math.isnan(orc.angle(1.0, float("nan")))


def test_sequence_layout_is_consistent(aof):
    """aof_sequence_layout: regions in order, 256-byte aligned, sized by the frame count; the crop must be the
    context's frame."""
    import pytest
    p = aof.px4flow_params(128, 128, pyramid_levels=2, mean_subtract=1)
    sp = aof.sequence_params(320, 240, 128, 128, derotate=(4.5, 0.01))
    for n in (0, 1, 2, 75, 4097):
        L = aof.sequence_layout(p, sp, n)
        offs = [L.cropped, L.exposure, L.flows, L.derotated, L.count, L.records, L.frames, L.frame_len, L.scratch, L.total_bytes]
        assert offs == sorted(offs) and all(o % 256 == 0 for o in offs[:-1])
        assert L.exposure - L.cropped >= n * 128 * 128 and L.records - L.count >= 16
        assert L.frames - L.records >= 32 * n and L.frame_len - L.frames >= 56 * n
        assert L.total_bytes - L.scratch >= 21 * (n + 1) + aof.workspace_layout(p, max(n - 1, 0)).total_bytes - 256
    with pytest.raises(aof.AofError):
        aof.sequence_layout(p, aof.sequence_params(320, 240, 64, 64), 10)       # crop != context frame
    with pytest.raises(aof.AofError):
        aof.sequence_layout(p, aof.sequence_params(100, 100, 128, 128), 10)     # crop larger than the sensor
