#!/usr/bin/env python3
"""CPU prototype for VERDICT r4 item 4: does an exact bound that survives noise exist for the 8x8 search?

Successive elimination with row sums: for a tile T and a candidate C (both 8x8),
    SAD(T, C) >= sum_r | rowsum_T(r) - rowsum_C(r) |  >=  | sum T - sum C |
so a candidate whose bound already exceeds the lane's best SAD cannot win or tie and need not be summed.  The kernels
drop work for a whole WAVE (64 lanes = 64 neighbouring blocks of a block row; a dy row of nine candidates is the unit):
a dy row is skipped only when, for EVERY lane that needs a result, ALL nine bounds of the row exceed that lane's best.
This script measures, on the bench's own synthetic frames (synth.make_pair) at several noise levels, for the wave
shape of k_search_lane8_cols:
  * rows dropped by the shipped partial-distortion test (16 then 32 of 64 pixels summed; aof_lane8.hpp pruned_row),
  * rows dropped by the row-sum bound (8 row sums per candidate), by the finer 2x... column+row bound, and by the
    coarse block-sum bound,
each with the visiting order of the kernel (start in the true row, outwards), and prices them in SAD-instruction
equivalents against the exhaustive block (432 SAD instructions).  Numbers only; nothing here is shipped.
usage: tools/sea_bound_prototype.py [pairs]      the bounds over sensor noise
       tools/sea_bound_prototype.py --inputs     the shipped test on the components of bench.py --input realistic
"""
import importlib.util, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "aero-optical-flow_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)

W, H, B, S = 640, 480, 8, 4
NX, NY = (W - 2 * S) // B, (H - 2 * S) // B


def all_sads(prev, cur):
    """sad[by, bx, dy, dx], rowsum bound rb[...], block-sum bound bb[...], row+col bound rcb[...], partial16/32[...]"""
    p = prev.astype(np.int32); c = cur.astype(np.int32)
    tiles = np.stack([[p[S + by * B:S + by * B + B, S + bx * B:S + bx * B + B] for bx in range(NX)] for by in range(NY)])   # [NY,NX,8,8]
    sad = np.empty((NY, NX, 9, 9), np.int32); rb = np.empty_like(sad); bb = np.empty_like(sad); rcb = np.empty_like(sad)
    p16 = np.empty_like(sad); p32 = np.empty_like(sad)
    trs = tiles.sum(3); tcs = tiles.sum(2); tbs = tiles.sum((2, 3))
    for dy in range(9):
        for dx in range(9):
            cand = np.stack([[c[by * B + dy:by * B + dy + B, bx * B + dx:bx * B + dx + B] for bx in range(NX)] for by in range(NY)])
            ad = np.abs(tiles - cand)
            sad[:, :, dy, dx] = ad.sum((2, 3))
            p16[:, :, dy, dx] = ad[:, :, [0, 4]].sum((2, 3))
            p32[:, :, dy, dx] = ad[:, :, [0, 4, 2, 6]].sum((2, 3))
            r = np.abs(trs - cand.sum(3)).sum(2); cc = np.abs(tcs - cand.sum(2)).sum(2)
            rb[:, :, dy, dx] = r
            rcb[:, :, dy, dx] = np.maximum(r, cc)
            bb[:, :, dy, dx] = np.abs(tbs - cand.sum((2, 3)))
    return sad, rb, rcb, bb, p16, p32


def visit(start):
    order = [start]
    for k in range(1, 9):
        for d in (start - k, start + k):
            if 0 <= d <= 8:
                order.append(d)
    return order


def simulate(sad, bound_rows, partial=None):
    """bound_rows[by,bx,dy] = min over dx of a lower bound of the row's SADs; partial: (p16,p32) for the staged test.
    Returns (rows evaluated fully, rows dropped at stage 1, at stage 2) per (wave, row) visit, as fractions of 9 rows."""
    full = s1 = s2 = tot = 0
    for by in range(NY):
        for x0 in range(0, NX, 64):
            sl = slice(x0, min(NX, x0 + 64))
            sd = sad[by, sl]                                   # [lanes,9,9]
            true_row = np.bincount(np.argmin(sd.reshape(sd.shape[0], -1), 1) // 9, minlength=9).argmax()
            best = np.full(sd.shape[0], 1 << 30)
            for k, d in enumerate(visit(int(true_row))):
                tot += 1
                if partial is None:
                    if k and np.all(bound_rows[by, sl, d] > best):
                        s1 += 1; continue
                else:
                    if k and np.all(partial[0][by, sl, d].min(1) > best):
                        s1 += 1; continue
                    if k and np.all(partial[1][by, sl, d].min(1) > best):
                        s2 += 1; continue
                full += 1
                best = np.minimum(best, sd[:, d].min(1))
    return full / tot, s1 / tot, s2 / tot


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
    print("noise | best SAD (median) | shipped partial-distortion: rows full / dropped@16px / dropped@32px -> SAD instr per block | "
          "row-sum bound: rows full -> instr | max(row,col)-sum bound: rows full -> instr | block-sum bound: rows full -> instr")
    for noise in (0, 4, 8, 16, 40):
        acc = []
        for i in range(pairs):
            prev, cur, _ = synth.make_pair(W, H, 4, 100 + i, noise=noise)
            sad, rb, rcb, bb, p16, p32 = all_sads(prev, cur)
            pd = simulate(sad, None, (p16, p32))
            r_ = simulate(sad, rb.min(3)); rc_ = simulate(sad, rcb.min(3)); b_ = simulate(sad, bb.min(3))
            acc.append((np.median(sad.reshape(NY, NX, -1).min(2)), pd, r_, rc_, b_))
        med = np.mean([a[0] for a in acc])
        pd = np.mean([a[1] for a in acc], 0); r_ = np.mean([a[2] for a in acc], 0); rc_ = np.mean([a[3] for a in acc], 0); b_ = np.mean([a[4] for a in acc], 0)
        # instruction model per block (SAD-class instructions; exhaustive = 432): a row summed completely = 48, dropped at the
        # 16-pixel test = 12 (+9 test), at the 32-pixel test = 24 (+18).  Row-sum bound: tile row sums 8 + per NEW window row
        # (8 of 16 with the column walk) nine sliding 8-byte sums = 3 qsad + 1 sad + 5 packed adds ~ 6.5 SAD-equivalents ->
        # 52; 81 candidates x 8 |differences| of u16 sums at two per v_sad_u16 = 324 (quantised to bytes and transposed so
        # that v_qsad slides along dy: 54 + 144 byte packs ~ 90) -> the bound costs ~ 150-380 per block BEFORE any candidate.
        cost_pd = 9 * (pd[0] * 48 + pd[1] * (12 + 2.25) + pd[2] * (24 + 4.5))
        def cost(b, pre): return pre + 9 * b[0] * 48
        print(f"+-{noise:2d} | {med:7.0f} | {pd[0]:.3f} / {pd[1]:.3f} / {pd[2]:.3f} -> {cost_pd:5.0f} | {r_[0]:.3f} -> {cost(r_, 150):5.0f} .. {cost(r_, 380):5.0f} | "
              f"{rc_[0]:.3f} -> {cost(rc_, 300):5.0f} .. {cost(rc_, 760):5.0f} | {b_[0]:.3f} -> {cost(b_, 60):5.0f}")
    print("exhaustive block: 432 SAD instructions (+222 other VALU).  A bound pays at >= 1.2x only below ~360.")


def inputs():
    """What the shipped partial-distortion test can drop on the components of bench.py's "realistic" input, one pair each:
    best and second-best SAD of a block (medians) and the rows a wave still sums completely."""
    cases = {"clean": {}, "noise 4": dict(noise=4), "noise 8": dict(noise=8), "noise 16": dict(noise=16),
             "half pixel (1,1)": dict(half=(1, 1)), "half pixel (1,0)": dict(half=(1, 0)), "contrast 0.8": dict(contrast=0.8),
             "contrast 0.5": dict(contrast=0.5), "noise 4 + half pixel": dict(noise=4, half=(1, 1)),
             "realistic (noise 4, half pixel, contrast 0.5)": dict(noise=4, half=(1, 1), contrast=0.5)}
    for name, kw in cases.items():
        prev, cur, _ = synth.make_pair(W, H, 4, 100, shift=(2, -1), **kw)
        sad, _, _, _, p16, p32 = all_sads(prev, cur)
        pd = simulate(sad, None, (p16, p32))
        flat = np.sort(sad.reshape(NY, NX, -1), 2)
        cost = 9 * (pd[0] * 48 + pd[1] * 14.25 + pd[2] * 28.5)
        print(f"{name:46s} best SAD {np.median(flat[:, :, 0]):6.0f}  runner-up {np.median(flat[:, :, 1]):6.0f}  rows full {pd[0]:.3f}  "
              f"dropped @16 px {pd[1]:.3f}  @32 px {pd[2]:.3f}  -> {cost:4.0f} SAD instructions per block (exhaustive 432)")


if __name__ == "__main__":
    inputs() if "--inputs" in sys.argv else main()
