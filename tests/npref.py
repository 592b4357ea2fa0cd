"""Second, independent restatement of DESIGN.md "Spec" in numpy (tests only).

The reference has no golden vectors (parity unpinned), so the C oracle is
pinned three ways: analytic known answers, hand-computed miniatures, and
agreement with this separately written numpy version on random inputs."""
import numpy as np

SKIPPED = 0xFFFF


def grid(p, level):
    w, h = p["width"] >> level, p["height"] >> level
    B, S = p["tile"], p["search"]
    if p["grid_mode"] == 0:
        M = S + (1 if p["subpixel"] else 0)
        return M, M, B, B, (w - 2 * M) // B, (h - 2 * M) // B
    lo = S + 1
    hix, hiy = w - (S + 1) - B, h - (S + 1) - B
    sx, sy = (hix - lo) // p["num_blocks"] + 1, (hiy - lo) // p["num_blocks"] + 1
    return lo, lo, sx, sy, len(range(lo, hix, sx)), len(range(lo, hiy, sy))


def level_range(p, level):
    return 3 * p["search"] + 1 if (p["pyramid_levels"] == 2 and level == 0) else p["search"]


def compute_diff(img, x, y, B):
    o = B // 2 - 2
    q = img[y + o:y + o + 4, x + o:x + o + 4].astype(np.int64)
    return int(np.abs(np.diff(q, axis=0)).sum() + np.abs(np.diff(q, axis=1)).sum())


def sad(a, ax, ay, b, bx, by, B):
    return int(np.abs(a[ay:ay + B, ax:ax + B].astype(np.int64) - b[by:by + B, bx:bx + B].astype(np.int64)).sum())


def subpixel(a, ax, ay, b, bx, by, B):
    bb = b.astype(np.int64)

    def win(ox, oy):
        return bb[by + oy:by + oy + B, bx + ox:bx + ox + B]

    c = win(0, 0)
    s0 = (c + win(1, 0)) >> 1
    s1 = (win(0, 1) + win(1, 1)) >> 1
    s2 = (c + win(0, 1)) >> 1
    s3 = (win(0, 1) + win(-1, 1)) >> 1
    s4 = (c + win(-1, 0)) >> 1
    s5 = (win(0, -1) + win(-1, -1)) >> 1
    s6 = (c + win(0, -1)) >> 1
    s7 = (win(0, -1) + win(1, -1)) >> 1
    dirs = [s0, (s0 + s1) >> 1, s2, (s3 + s4) >> 1, s4, (s4 + s5) >> 1, s6, (s7 + s0) >> 1]
    ref = a[ay:ay + B, ax:ax + B].astype(np.int64)
    return [int(np.abs(ref - d).sum()) for d in dirs]


def frame_mean(img):
    n = img.size
    return (int(img.astype(np.int64).sum()) + n // 2) // n


def pyramid_down(img):
    q = img.astype(np.int64)
    return ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2).astype(np.uint8)


def equalise(img, d):
    return np.clip(img.astype(np.int64) + d, 0, 255).astype(np.uint8)


def level_search(p, prev, cur, level, px, py):
    h, w = prev.shape
    B, S, m = p["tile"], p["search"], (1 if p["subpixel"] else 0)
    x0, y0, sx, sy, nx, ny = grid(p, level)
    vthr = min(p["value_threshold"], 0xFFFF)
    recs, subs = [], []
    for by in range(ny):
        for bx in range(nx):
            i, j = x0 + bx * sx, y0 + by * sy
            rec, sd = (0, 0, SKIPPED), 8
            ok = (i + px - S - m >= 0 and j + py - S - m >= 0 and i + px + S + m + B <= w
                  and j + py + S + m + B <= h)
            if ok and compute_diff(prev, i, j, B) >= p["feature_threshold"]:
                best = None
                for jj in range(-S, S + 1):
                    for ii in range(-S, S + 1):
                        t = sad(prev, i, j, cur, i + px + ii, j + py + jj, B)
                        if best is None or t < best[0]:
                            best = (t, ii, jj)
                rec = (px + best[1], py + best[2], best[0])
                if p["subpixel"] and best[0] < vthr:
                    acc = subpixel(prev, i, j, cur, i + rec[0], j + rec[1], B)
                    mind = best[0]
                    for k in range(8):
                        if acc[k] < mind:
                            mind, sd = acc[k], k
            recs.append(rec)
            subs.append(sd)
    return recs, subs


def reduce(p, recs, subs, R):
    centre = 2 * R + 1
    n = 2 * centre + 1
    vthr = min(p["value_threshold"], 0xFFFF)
    hx, hy = [0] * n, [0] * n
    s2x = s2y = cnt = 0
    for (dx, dy, s), sd in zip(recs, subs):
        if s == SKIPPED or s >= vthr:
            continue
        ax = 1 if sd in (0, 1, 7) else (-1 if sd in (3, 4, 5) else 0)
        ay = 1 if sd in (1, 2, 3) else (-1 if sd in (5, 6, 7) else 0)
        hx[2 * dx + ax + centre] += 1
        hy[2 * dy + ay + centre] += 1
        s2x += 2 * dx + ax
        s2y += 2 * dy + ay
        cnt += 1
    out = dict(flow_x=np.float32(0), flow_y=np.float32(0), count=cnt, quality=0, valid=False, px=0, py=0)
    if not (cnt > p["min_valid"] and cnt > 0):
        return out
    f32 = np.float32

    def peak(hist):
        pos = int(np.argmax(hist))  # first maximum
        if 1 < pos < n - 2:
            lo, hi = pos - 2, pos + 2
        elif pos == 0:
            lo, hi = 0, 2
        elif pos == n - 1:
            lo, hi = pos - 2, pos
        elif pos == 1:
            lo, hi = 0, 3
        else:
            lo, hi = pos - 2, pos + 1
        v = sum(k * hist[k] for k in range(lo, hi + 1))
        wgt = sum(hist[k] for k in range(lo, hi + 1))
        return v, wgt

    if p["hist_filter"]:
        vx, wx = peak(hx)
        vy, wy = peak(hy)
        out["flow_x"] = (f32(vx) / f32(wx) - f32(centre)) / f32(2)
        out["flow_y"] = (f32(vy) / f32(wy) - f32(centre)) / f32(2)
        out["px"] = (2 * vx + wx) // (2 * wx) - centre
        out["py"] = (2 * vy + wy) // (2 * wy) - centre
    else:
        out["flow_x"] = (f32(s2x) * f32(0.5)) / f32(cnt)
        out["flow_y"] = (f32(s2y) * f32(0.5)) / f32(cnt)
        out["px"] = (2 * s2x + cnt) // (2 * cnt)
        out["py"] = (2 * s2y + cnt) // (2 * cnt)
    out["quality"] = cnt * 255 // len(recs)
    out["valid"] = True
    return out


def flow_pair(p, prev, cur):
    px = py = 0
    pred_valid = False
    if p["pyramid_levels"] == 2:
        p1, c1 = pyramid_down(prev), pyramid_down(cur)
        if p["mean_subtract"]:
            c1 = equalise(c1, frame_mean(p1) - frame_mean(c1))
        r1, s1 = level_search(p, p1, c1, 1, 0, 0)
        f1 = reduce(p, r1, s1, level_range(p, 1))
        px, py, pred_valid = f1["px"], f1["py"], f1["valid"]
    c0 = cur
    if p["mean_subtract"]:
        c0 = equalise(cur, frame_mean(prev) - frame_mean(cur))
    r0, s0 = level_search(p, prev, c0, 0, px, py)
    f0 = reduce(p, r0, s0, level_range(p, 0))
    f0.update(pred_x=px, pred_y=py, pred_valid=pred_valid, recs=r0, subs=s0)
    return f0
