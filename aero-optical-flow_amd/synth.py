"""Synthetic frame pairs for the flow path (BASELINE.md "Synthetic inputs").

A seeded uniform-random u8 field on a (W+2R)x(H+2R) canvas is 3x3 box-blurred
so that sub-tile structure exists, then cropped twice at an integer offset:
``cur(x, y) == prev(x - dx, y - dy)`` wherever both are defined, so every
textured interior block must report exactly (dx, dy) -- an analytic
known-answer test that does not depend on any implementation.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 0xA0F


def canvas(width, height, reach, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = rng.integers(0, 256, size=(height + 2 * reach + 2, width + 2 * reach + 2), dtype=np.int32)
    acc = np.zeros((height + 2 * reach, width + 2 * reach), dtype=np.int32)
    for oy in range(3):
        for ox in range(3):
            acc += raw[oy:oy + acc.shape[0], ox:ox + acc.shape[1]]
    return ((acc + 4) // 9).astype(np.uint8)


def make_pair(width, height, reach=4, pair_index=0, shift=None, noise=0, brightness=0, contrast=1.0,
              half=(0, 0)):
    """Returns (prev, cur, (dx, dy)).  ``shift=None`` draws (dx, dy) uniformly from
    [-reach, reach]^2 with the pair's RNG.  ``noise``: +-noise LSB added to cur;
    ``brightness``: constant added to cur (saturating) -- the exposure step the
    reference's auto-exposure loop produces (/root/reference/src/mainloop.cpp:197-275);
    ``half``: (hx, hy) in {-1,0,1}: an extra half-pixel displacement, made by averaging
    the crop with its one-pixel neighbour (needs |shift| < reach on that axis)."""
    seed = SEED_BASE + int(pair_index)
    c = canvas(width, height, reach, seed)
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    if shift is None:
        dx, dy = (int(v) for v in rng.integers(-reach, reach + 1, size=2))
    else:
        dx, dy = int(shift[0]), int(shift[1])
    assert abs(dx) <= reach and abs(dy) <= reach
    prev = c[reach:reach + height, reach:reach + width].copy()
    cur = c[reach - dy:reach - dy + height, reach - dx:reach - dx + width].astype(np.int32)
    if half != (0, 0):
        hx, hy = half
        assert abs(dx + hx) <= reach and abs(dy + hy) <= reach
        nb = c[reach - dy - hy:reach - dy - hy + height, reach - dx - hx:reach - dx - hx + width]
        cur = (cur + nb.astype(np.int32)) >> 1
    if contrast != 1.0:
        cur = np.rint((cur - 128) * contrast + 128).astype(np.int32)
    if noise:
        cur = cur + rng.integers(-noise, noise + 1, size=cur.shape)
    cur = np.clip(cur + brightness, 0, 255).astype(np.uint8)
    return prev, cur, (dx, dy)


def make_batch(width, height, n_pairs, reach=4, first_index=0, **kw):
    prevs = np.empty((n_pairs, height, width), dtype=np.uint8)
    curs = np.empty_like(prevs)
    shifts = np.empty((n_pairs, 2), dtype=np.int32)
    for i in range(n_pairs):
        prevs[i], curs[i], shifts[i] = make_pair(width, height, reach, first_index + i, **kw)
    return prevs, curs, shifts


def make_sequence(width, height, n_frames, reach=4, seed=0, max_step=None):
    """A frame SEQUENCE (camera panning over one canvas): frame k+1 is frame k
    displaced by a per-step shift of at most ``max_step`` (default ``reach``)."""
    if max_step is None:
        max_step = reach
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + 7919 * (seed + 1)))
    steps = rng.integers(-max_step, max_step + 1, size=(n_frames - 1, 2))
    pos = np.zeros((n_frames, 2), dtype=np.int64)
    pos[1:] = np.cumsum(steps, axis=0)
    span = int(np.abs(pos).max()) + 1
    c = canvas(width, height, span, SEED_BASE + 104729 * (seed + 1))
    frames = np.empty((n_frames, height, width), dtype=np.uint8)
    for k in range(n_frames):
        ox, oy = span - int(pos[k, 0]), span - int(pos[k, 1])
        frames[k] = c[oy:oy + height, ox:ox + width]
    return frames, steps.astype(np.int32)
