"""AOF_SEARCH_PRUNED (exact partial-distortion elimination in the lane8 and tile16 kernels) must return
the same bytes as the exhaustive search and as the oracle on every kind of input: clean
translations (maximal pruning), noise (no pruning), periodic textures and identical frames
(ties everywhere -- the tie-break must survive the changed visiting order), flat regions
(whole waves without a live block), saturating equalisation, displaced windows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(aof, p, prevs, curs, device, mode, hint_fill=None, reps=1):
    import torch
    eng = aof.FlowEngine(p, 0)
    eng.set_search_mode(mode)
    tp, tc = torch.from_numpy(prevs).to(device), torch.from_numpy(curs).to(device)
    n = prevs.shape[0]
    L = aof.workspace_layout(p, n)
    ws = torch.zeros(L.total_bytes, dtype=torch.uint8, device=device)
    if hint_fill is not None:  # the workspace may hold anything on entry
        ws.fill_(hint_fill & 0xFF)
    for _ in range(reps):
        blocks, flows, _ = eng.flow_batch(tp, tc, workspace=ws)
    torch.cuda.synchronize()
    return aof.blocks_view(blocks), aof.flows_view(flows)


def both_modes_match_oracle(aof, orc, p, prevs, curs, device):
    po = orc.params_from(p)
    refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(prevs.shape[0])]
    # both search modes; the workspace may hold anything on entry, launches may repeat
    for mode, hint, reps in ((aof.SEARCH_EXHAUSTIVE, None, 1), (aof.SEARCH_PRUNED, None, 1), (aof.SEARCH_PRUNED, 0, 2),
                             (aof.SEARCH_PRUNED, 8, 2), (aof.SEARCH_PRUNED, -12345, 1), (aof.SEARCH_PRUNED, 77, 3)):
        b, f = run(aof, p, prevs, curs, device, mode, hint, reps)
        for i, r in enumerate(refs):
            assert b[i].tobytes() == r["blocks"].tobytes(), (mode, hint, i)
            assert f[i].tobytes() == r["flow"].tobytes(), (mode, hint, i)


@pytest.mark.parametrize("noise", [0, 3, 40])
def test_pruned_vga(aof, orc, synth, gpu_device, noise):
    p = aof.default_params(640, 480)
    prevs, curs, _ = synth.make_batch(640, 480, 3, 4, 8100 + noise, noise=noise)
    both_modes_match_oracle(aof, orc, p, prevs, curs, gpu_device)


def test_pruned_unrelated_noise_and_identical_frames(aof, orc, gpu_device):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (3, 96, 128), dtype=np.uint8)
    b = rng.integers(0, 256, (3, 96, 128), dtype=np.uint8)
    b[1] = a[1]                                   # identical: SAD 0 at (0,0) only
    a[2, :, :64] = 128                            # half flat: waves without live blocks
    for kw in (dict(), dict(value_threshold=70000, feature_threshold=0), dict(mean_subtract=1)):
        both_modes_match_oracle(aof, orc, aof.default_params(128, 96, **kw), a, b, gpu_device)


def test_pruned_keeps_first_minimum_on_ties(aof, orc, gpu_device):
    img = np.zeros((96, 128), np.uint8)
    img[:, 0::2] = 200                            # period 2 in x
    img2 = np.zeros((96, 128), np.uint8)
    img2[0::2, :] = 150                           # period 2 in y: every other dy row ties at SAD 0
    img3 = np.zeros((96, 128), np.uint8)
    img3[0::4, 0::4] = 255                        # period 4 both ways
    yy, xx = np.mgrid[0:96, 0:128]
    chk = (((xx // 8 + yy // 8) % 2) * 255).astype(np.uint8)
    prevs = np.stack([img, img2, img3, chk, chk])
    curs = np.stack([img, img2, img3, chk, 255 - chk])
    for kw in (dict(), dict(feature_threshold=0, value_threshold=70000)):
        p = aof.default_params(128, 96, **kw)
        both_modes_match_oracle(aof, orc, p, prevs, curs, gpu_device)
        b, _ = run(aof, p, prevs, curs, gpu_device, aof.SEARCH_PRUNED, hint_fill=8)
        ok = b[1]["sad"] != 0xFFFF
        assert (b[1]["dy"][ok] == -4).all() and (b[1]["dx"][ok] == -4).all(), "first minimum in scan order"


def test_pruned_two_level_with_predictor(aof, orc, synth, gpu_device):
    p = aof.default_params(192, 160, pyramid_levels=2, mean_subtract=1)
    for shift, bright in (((9, 8), 14), ((-9, -7), -20), ((3, -2), 0), ((8, -9), 5)):
        prev, cur, _ = synth.make_pair(192, 160, 12, 31, shift=shift, noise=2, brightness=bright)
        both_modes_match_oracle(aof, orc, p, prev[None], cur[None], gpu_device)


@pytest.mark.parametrize("seed", range(12))
def test_pruned_fuzz(aof, orc, synth, gpu_device, seed):
    rng = np.random.default_rng(400 + seed)
    w = int(rng.integers(3, 20)) * 16
    h = int(rng.integers(40, 200))
    kw = dict(width=w, height=h, mean_subtract=int(rng.integers(0, 2)), hist_filter=int(rng.integers(0, 2)),
              feature_threshold=int(rng.choice([0, 30, 200])), value_threshold=int(rng.choice([0, 500, 3000, 70000])),
              min_valid=int(rng.choice([0, 10])))
    if rng.random() < 0.4 and h % 2 == 0:
        kw["pyramid_levels"] = 2
    p = aof.default_params(**kw)
    if aof.check_params(p) != 0:
        pytest.skip("geometry too small")
    reach = 9 if p.pyramid_levels == 2 else 4
    prevs, curs, _ = synth.make_batch(w, h, 3, reach, 8800 + seed, noise=int(rng.integers(0, 30)),
                                      brightness=int(rng.integers(-25, 26)), contrast=float(rng.choice([1.0, 2.0, 0.2])))
    if seed % 3 == 0:
        curs[0] = rng.integers(0, 256, curs[0].shape, dtype=np.uint8)
    both_modes_match_oracle(aof, orc, p, prevs, curs, gpu_device)


def tiled(torch, arr, reps, device):
    """`reps` copies of a small set of frames, interleaved, as one device batch."""
    t = torch.from_numpy(arr).to(device)
    return t.repeat((reps,) + (1,) * (t.dim() - 1)).contiguous()


def replicas_equal(aof, blocks, flows, refs):
    gb, gf = aof.blocks_view(blocks), aof.flows_view(flows)
    k = len(refs)
    for i in range(gb.shape[0]):
        r = refs[i % k]
        if gb[i].tobytes() != r["blocks"].tobytes() or gf[i].tobytes() != r["flow"].tobytes():
            return i
    return -1


def half_pixel_modes_match_oracle(aof, orc, p, prevs, curs, device):
    """Records, directions and flow of the exhaustive and the pruned 8x8 kernel against the oracle."""
    import torch
    po = orc.params_from(p)
    refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(prevs.shape[0])]
    tp, tc = torch.from_numpy(prevs).to(device), torch.from_numpy(curs).to(device)
    for mode in (aof.SEARCH_EXHAUSTIVE, aof.SEARCH_PRUNED):
        eng = aof.FlowEngine(p, 0)
        eng.set_search_mode(mode)
        sub = torch.full((prevs.shape[0], eng.nblocks(0)), 0x77, dtype=torch.uint8, device=device)
        for _ in range(2):
            blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
        torch.cuda.synchronize()
        gb, gf, gs = aof.blocks_view(blocks), aof.flows_view(flows), sub.cpu().numpy()
        for i, r in enumerate(refs):
            assert gb[i].tobytes() == r["blocks"].tobytes(), (mode, i)
            assert gs[i].tobytes() == r["subdirs"].tobytes(), (mode, i, np.flatnonzero(gs[i] != r["subdirs"])[:8])
            assert gf[i].tobytes() == r["flow"].tobytes(), (mode, i)
        eng.close()


def test_pruned_half_pixel_ring_from_the_window_registers(aof, orc, synth, gpu_device):
    """The pruned 8x8 kernel refines from its window registers where the best match is not on the window's rim and the
    wave's refining lanes share a dy, and from memory otherwise: every shift incl. the rim (+-4: the ring leaves the
    window), frames whose halves move differently (several dy in one wave), equalisation, a predictor, noise."""
    W, H = 192, 160
    shifts = [(0, 0), (4, 4), (-4, -4), (4, -3), (-3, 4), (1, -1), (3, 3), (-2, 0), (0, 4), (-4, 1)]
    halves = [(1, 0), (0, 0), (0, 0), (0, 1), (1, 0), (-1, 1), (1, -1), (1, 1), (1, 0), (0, -1)]   # (inside the reach)
    pairs = [synth.make_pair(W, H, 4, 900 + k, shift=sh, noise=k % 3, half=hf) for k, (sh, hf) in enumerate(zip(shifts, halves))]
    prevs = np.stack([q[0] for q in pairs]); curs = np.stack([q[1] for q in pairs])
    # two motions in one frame: the upper part moves like pair 1, the lower like pair 5 (waves straddle the seam)
    a, b = synth.make_pair(W, H, 4, 950, shift=(2, -3)), synth.make_pair(W, H, 4, 950, shift=(-1, 2))
    mixed = a[1].copy(); mixed[70:] = b[1][70:]
    prevs = np.concatenate([prevs, a[0][None]]); curs = np.concatenate([curs, mixed[None]])
    rng = np.random.default_rng(8)
    curs = np.concatenate([curs, rng.integers(0, 256, (1, H, W), dtype=np.uint8)])      # unrelated: directions of junk matches
    prevs = np.concatenate([prevs, prevs[:1]])
    for kw in (dict(subpixel=1), dict(subpixel=1, mean_subtract=1), dict(subpixel=1, feature_threshold=0, value_threshold=70000)):
        half_pixel_modes_match_oracle(aof, orc, aof.default_params(W, H, **kw), prevs, curs, gpu_device)
    bright = np.clip(curs.astype(np.int16) + 23, 0, 255).astype(np.uint8)               # equalisation with clamping
    half_pixel_modes_match_oracle(aof, orc, aof.default_params(W, H, subpixel=1, mean_subtract=1), prevs, bright, gpu_device)
    p2 = aof.default_params(W, H, subpixel=1, pyramid_levels=2, mean_subtract=1)
    big = [synth.make_pair(W, H, 12, 970 + k, shift=sh, noise=1, brightness=9) for k, sh in enumerate([(9, -8), (-7, 6), (5, 5), (0, -9)])]
    half_pixel_modes_match_oracle(aof, orc, p2, np.stack([q[0] for q in big]), np.stack([q[1] for q in big]), gpu_device)


def test_half_pixel_ring_that_ends_on_the_last_byte_of_the_last_frame(aof, orc, synth, gpu_device):
    """The bottom-right block of the LAST pair of a batch, matched at (+4, +4) on a frame of (8k + 18) x (8k + 18) pixels: the
    last of the ten bytes of its ring's last row is the last byte of the frame arrays.  A 16-byte row load that reaches
    past that byte does not return those bytes (found by tools/fuzz_gpu.py, seed 11 517 551: one direction in 150 000
    cases): such lanes load their ring byte-exactly.  Twenty textures, each the last pair of its batch, on flat and on
    grouped grids, every search mode and the generic kernel: directions, records and flows against the oracle."""
    import torch
    for (W, H), kw in (((146, 130), dict(subpixel=1)), ((226, 218), dict(subpixel=1)), ((226, 218), dict(subpixel=1, mean_subtract=1))):
        p = aof.default_params(W, H, **kw)
        g = aof.grid(p, 0)
        assert g[0] + 8 * (g[4] - 1) == W - 13 and g[1] + 8 * (g[5] - 1) == H - 13
        po = orc.params_from(p)
        engines = {}
        for mode in (aof.SEARCH_EXHAUSTIVE, aof.SEARCH_PRUNED, "generic"):
            eng = aof.FlowEngine(p, 0)
            if mode == "generic":
                eng.force_generic(True)
            else:
                eng.set_search_mode(mode)
            engines[mode] = eng
        for k in range(20):
            n = 1 + k % 3
            pairs = [synth.make_pair(W, H, 4, 700 + 7 * k + q, shift=(4, 4), noise=k % 4, brightness=(k % 5) * 4) for q in range(n)]
            prevs, curs = np.stack([q[0] for q in pairs]), np.stack([q[1] for q in pairs])
            refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(n)]
            tp, tc = torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device)
            for mode, eng in engines.items():
                sub = torch.full((n, eng.nblocks(0)), 0x77, dtype=torch.uint8, device=gpu_device)
                blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
                torch.cuda.synchronize()
                gb, gf, gs = aof.blocks_view(blocks), aof.flows_view(flows), sub.cpu().numpy()
                for i in range(n):
                    assert gb[i].tobytes() == refs[i]["blocks"].tobytes(), (W, kw, mode, k, i)
                    bad = np.flatnonzero(gs[i] != refs[i]["subdirs"])
                    assert bad.size == 0, (W, kw, mode, k, i, bad[:8], gs[i][bad[:8]], refs[i]["subdirs"][bad[:8]])
                    assert gf[i].tobytes() == refs[i]["flow"].tobytes(), (W, kw, mode, k, i)
        for eng in engines.values():
            eng.close()


def test_pruned_walks_on_grids_that_are_not_dense(aof, orc, synth, gpu_device):
    """The pruned 8x8 search has two walks: lanes that own a column of blocks and keep half of their window for the block
    below (grids whose rows are one tile apart: the dense grid, and sparse grids with a row step of 8), and workgroups that
    walk chunks of 256 blocks (every other grid of more than 256 blocks).  Sparse grids of both kinds, with and without the
    half-pixel step, two levels and equalisation: records, directions and flows of both search modes against the oracle, and
    the adaptive default on a launch large enough to prune."""
    import torch
    cases = [((320, 240), dict(grid_mode=1, num_blocks=20)),                 # 19 x 19 blocks, steps 16 / 12: chunk walk
             ((192, 160), dict(grid_mode=1, num_blocks=20)),                 # 20 x 18, steps 9 / 8: column walk, columns 9 apart
             ((192, 160), dict(grid_mode=1, num_blocks=24)),                 # 22 x 24, steps 8 / 6: chunk walk
             ((192, 160), dict(grid_mode=1, num_blocks=20, subpixel=1, mean_subtract=1)),
             ((320, 240), dict(grid_mode=1, num_blocks=20, subpixel=1)),
             ((192, 160), dict(grid_mode=1, num_blocks=20, pyramid_levels=2, mean_subtract=1))]
    for (W, H), kw in cases:
        p = aof.default_params(W, H, **kw)
        assert aof.check_params(p) == 0 and aof.grid(p, 0)[4] * aof.grid(p, 0)[5] > 256
        reach = 9 if p.pyramid_levels == 2 else 4
        prevs, curs, _ = synth.make_batch(W, H, 5, reach, 8600, noise=2, brightness=7 if p.mean_subtract else 0)
        rng = np.random.default_rng(9)
        curs[4] = rng.integers(0, 256, curs[4].shape, dtype=np.uint8)
        po = orc.params_from(p)
        refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(5)]
        nb = aof.grid(p, 0)[4] * aof.grid(p, 0)[5]
        for mode, reps in ((aof.SEARCH_EXHAUSTIVE, 1), (aof.SEARCH_PRUNED, 1), (aof.SEARCH_ADAPTIVE, (4096 * 256 // nb + 5) // 5 + 1)):
            eng = aof.FlowEngine(p, 0)
            eng.set_search_mode(mode)
            tp, tc = tiled(torch, prevs, reps, gpu_device), tiled(torch, curs, reps, gpu_device)
            sub = torch.full((5 * reps, nb), 0x77, dtype=torch.uint8, device=gpu_device) if p.subpixel else None
            for _ in range(2):
                blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
                torch.cuda.synchronize()
                assert replicas_equal(aof, blocks, flows, refs) < 0, (kw, mode)
                if sub is not None:
                    gs = sub.cpu().numpy()
                    for i in range(gs.shape[0]):
                        assert gs[i].tobytes() == refs[i % 5]["subdirs"].tobytes(), (kw, mode, i)
            if mode == aof.SEARCH_ADAPTIVE:
                assert eng.search_stats()["pruned_launches"] >= 1
            eng.close()


# ---- 16x16 tiles, +-8 (configs[4] geometry): the same exactness bar for k_search_tile16<PRUNE> ----

def pruned16_matches(aof, orc, p, prevs, curs, device):
    po = orc.params_from(p)
    refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(prevs.shape[0])]
    for mode in (aof.SEARCH_EXHAUSTIVE, aof.SEARCH_PRUNED, aof.SEARCH_ADAPTIVE):
        b, f = run(aof, p, prevs, curs, device, mode, hint_fill=0x5A, reps=2)
        for i, r in enumerate(refs):
            assert b[i].tobytes() == r["blocks"].tobytes(), (mode, i)
            assert f[i].tobytes() == r["flow"].tobytes(), (mode, i)


@pytest.mark.parametrize("noise", [0, 3, 40])
def test_pruned16_translations_and_noise(aof, orc, synth, gpu_device, noise):
    p = aof.default_params(320, 256, tile=16, search=8, value_threshold=12000)
    prevs, curs, _ = synth.make_batch(320, 256, 3, 8, 9400 + noise, noise=noise)
    pruned16_matches(aof, orc, p, prevs, curs, gpu_device)


def test_pruned16_c5_full_size(aof, orc, synth, gpu_device):
    p = aof.default_params(1280, 960, tile=16, search=8, value_threshold=12000)
    prevs, curs, shifts = synth.make_batch(1280, 960, 2, 8, 510, noise=2)
    pruned16_matches(aof, orc, p, prevs, curs, gpu_device)


def test_pruned16_ties_flat_and_unrelated_frames(aof, orc, gpu_device):
    rng = np.random.default_rng(16)
    H, W = 160, 192
    img = np.zeros((H, W), np.uint8); img[:, 0::2] = 200          # period 2 in x: ties along dx
    img2 = np.zeros((H, W), np.uint8); img2[0::2, :] = 150        # period 2 in y: every other dy row ties
    img3 = np.zeros((H, W), np.uint8); img3[0::4, 0::4] = 255
    yy, xx = np.mgrid[0:H, 0:W]
    chk = (((xx // 16 + yy // 16) % 2) * 255).astype(np.uint8)
    noise_a = rng.integers(0, 256, (H, W), dtype=np.uint8)
    noise_b = rng.integers(0, 256, (H, W), dtype=np.uint8)
    half = noise_a.copy(); half[:, : W // 2] = 90                  # half flat: gated blocks
    prevs = np.stack([img, img2, img3, chk, chk, noise_a, noise_a, half])
    curs = np.stack([img, img2, img3, chk, 255 - chk, noise_b, noise_a, half])
    for kw in (dict(), dict(feature_threshold=0, value_threshold=70000), dict(mean_subtract=1)):
        pruned16_matches(aof, orc, aof.default_params(W, H, tile=16, search=8, **kw), prevs, curs, gpu_device)


def test_pruned16_two_level_predictor_and_half_pixel(aof, orc, synth, gpu_device):
    for kw in (dict(pyramid_levels=2, mean_subtract=1), dict(subpixel=1), dict(pyramid_levels=2, subpixel=1)):
        p = aof.default_params(320, 256, tile=16, search=8, value_threshold=12000, **kw)
        reach = 17 if kw.get("pyramid_levels") == 2 else 8
        prevs, curs, _ = synth.make_batch(320, 256, 3, reach, 9500, noise=3, brightness=9 if kw.get("mean_subtract") else 0)
        pruned16_matches(aof, orc, p, prevs, curs, gpu_device)


@pytest.mark.parametrize("seed", range(8))
def test_pruned16_fuzz(aof, orc, synth, gpu_device, seed):
    rng = np.random.default_rng(1600 + seed)
    w = int(rng.integers(5, 24)) * 16
    h = int(rng.integers(40, 130)) * 2
    kw = dict(width=w, height=h, tile=16, search=8, mean_subtract=int(rng.integers(0, 2)),
              hist_filter=int(rng.integers(0, 2)), feature_threshold=int(rng.choice([0, 30, 200])),
              value_threshold=int(rng.choice([0, 2000, 12000, 70000])), min_valid=int(rng.choice([0, 10])))
    if rng.random() < 0.4:
        kw["pyramid_levels"] = 2
    p = aof.default_params(**kw)
    if aof.check_params(p) != 0:
        pytest.skip("geometry too small")
    reach = 17 if p.pyramid_levels == 2 else 8
    prevs, curs, _ = synth.make_batch(w, h, 3, reach, 9900 + seed, noise=int(rng.integers(0, 30)),
                                      brightness=int(rng.integers(-25, 26)), contrast=float(rng.choice([1.0, 2.0, 0.2])))
    if seed % 3 == 0:
        curs[0] = rng.integers(0, 256, curs[0].shape, dtype=np.uint8)
    pruned16_matches(aof, orc, p, prevs, curs, gpu_device)


def test_adaptive16_is_the_default_and_judges_every_pair_by_itself(aof, orc, synth, gpu_device):
    """A fresh 16x16 context runs AOF_SEARCH_ADAPTIVE: a probe kernel in front of the search judges every pair
    (aof_ws_layout.hints: 2 = one tile row separates the candidates already, 1 = few candidates survive the two-row
    bounds, 3 / 4 = sensor noise that four / eight rows still see through: run the pruned steps on bounds of that depth;
    0 = nothing separates the candidates, run the exhaustive scan) and the search follows the verdict per pair.  A batch that mixes noise-free translations, a little and a lot of sensor noise, unrelated
    frames and a flat pair must come out with ALL verdicts present, records and
    flows equal to the oracle's for every pair, and the same bytes as the two fixed modes -- for one and two
    levels and with the half-pixel step."""
    import torch
    W, H = 1280, 960
    for kw in (dict(), dict(subpixel=1), dict(pyramid_levels=2, mean_subtract=1)):
        p = aof.default_params(W, H, tile=16, search=8, value_threshold=12000, **kw)
        reach = 17 if kw.get("pyramid_levels") == 2 else 8
        clean_p, clean_c, _ = synth.make_batch(W, H, 2, reach, 7100, brightness=9 if kw.get("mean_subtract") else 0)
        noisy_p, noisy_c, _ = synth.make_batch(W, H, 2, reach, 7200, noise=40)
        some_p, some_c, _ = synth.make_batch(W, H, 1, reach, 7300, noise=3, brightness=9 if kw.get("mean_subtract") else 0)
        rng = np.random.default_rng(72)
        unrelated = rng.integers(0, 256, (H, W), dtype=np.uint8)
        flat = np.full((H, W), 90, np.uint8)
        prevs = np.stack([clean_p[0], noisy_p[0], clean_p[1], noisy_p[1], clean_p[0], flat, some_p[0]])
        curs = np.stack([clean_c[0], noisy_c[0], clean_c[1], noisy_c[1], unrelated, flat, some_c[0]])
        eng = aof.FlowEngine(p, 0)
        assert eng.search_mode == aof.SEARCH_ADAPTIVE and eng.variant == "tile16_lds"
        tp, tc = torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device)
        L = aof.workspace_layout(p, 7)
        ws = torch.full((L.total_bytes,), 0x5A, dtype=torch.uint8, device=gpu_device)
        blocks, flows, _ = eng.flow_batch(tp, tc, workspace=ws)
        torch.cuda.synchronize()
        hints = ws[L.hints:L.hints + 4 * 7].cpu().numpy().view(np.uint32) & 0xFF   # (upper bits: the probe's separation figure)
        assert set(hints.tolist()) <= {0, 1, 2, 3, 4}
        assert hints[0] == 2 and hints[2] == 2, hints          # noise-free translations: pruning pays, on one-row bounds
        assert hints[6] == 1, hints                            # +-3 LSB: pruning pays, on the two-row bounds
        assert hints[1] in (0, 3, 4) and hints[3] in (0, 3, 4), hints     # +-40 LSB: deeper bounds or none
        assert hints[4] == 0 and hints[5] == 0, hints          # unrelated frames, a flat pair: nothing to prune with
        got = dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows))
        po = orc.params_from(p)
        for i in range(7):
            ref = orc.flow_pair(po, prevs[i], curs[i])
            assert got["blocks"][i].tobytes() == ref["blocks"].tobytes(), (kw, i)
            assert got["flows"][i].tobytes() == ref["flow"].tobytes(), (kw, i)
        for mode in (aof.SEARCH_EXHAUSTIVE, aof.SEARCH_PRUNED):
            eng.set_search_mode(mode)
            b2, f2, _ = eng.flow_batch(tp, tc)
            torch.cuda.synchronize()
            assert torch.equal(b2, blocks) and torch.equal(f2, flows), (kw, mode)
        eng.close()


@pytest.mark.parametrize("size", [(32, 5008), (48, 2000), (64, 64), (16 * 30, 48)])
def test_adaptive16_probe_on_narrow_tall_and_tiny_grids(aof, orc, synth, gpu_device, size):
    """The probe samples every 8th block per axis; an axis with fewer blocks is sampled more densely, and a long
    one more sparsely so that the sample fits its table (a 1 x 311 grid once indexed past it).  All three modes
    against the oracle on such geometries, clean and noisy."""
    W, H = size
    p = aof.default_params(W, H, tile=16, search=8, value_threshold=12000, min_valid=0)
    if aof.check_params(p) != 0:
        pytest.skip("geometry too small")
    prevs, curs, _ = synth.make_batch(W, H, 2, 8, 9700 + W, noise=0)
    rng = np.random.default_rng(5)
    curs[1] = np.clip(curs[1].astype(np.int16) + rng.integers(-40, 41, curs[1].shape), 0, 255).astype(np.uint8)
    pruned16_matches(aof, orc, p, prevs, curs, gpu_device)


# ---- 8x8 tiles: AOF_SEARCH_ADAPTIVE, the default -- the context learns from its own launches which kernel to run ----

def test_adaptive8_is_the_default_and_learns_from_its_own_launches(aof, orc, synth, gpu_device):
    """A fresh 8x8 context searches in AOF_SEARCH_ADAPTIVE.  Launches of at least 2 048 chunks of 256 blocks go to the
    pruned kernel (whose waves judge, chunk by chunk, whether rows could be dropped) while the kernel's own reports
    say that pruning pays, to the exhaustive kernel -- with one pruned launch in 16 to look again -- when they say
    it does not.  Whatever it picks, the records are the oracle's, byte for byte."""
    import torch
    W, H, base, reps = 192, 160, 6, 480            # 23 x 19 = 437 blocks per pair, 2 880 pairs: 4 916 chunks
    p = aof.default_params(W, H)
    po = orc.params_from(p)
    prevs, clean, _ = synth.make_batch(W, H, base, 4, 8100, noise=0)
    rng = np.random.default_rng(5)
    noisy = np.clip(clean.astype(np.int16) + rng.integers(-40, 41, clean.shape), 0, 255).astype(np.uint8)
    tp = tiled(torch, prevs, reps, gpu_device)
    batches = {"clean": (tiled(torch, clean, reps, gpu_device), [orc.flow_pair(po, prevs[i], clean[i]) for i in range(base)]),
               "noisy": (tiled(torch, noisy, reps, gpu_device), [orc.flow_pair(po, prevs[i], noisy[i]) for i in range(base)])}
    eng = aof.FlowEngine(p, 0)
    assert eng.search_mode == aof.SEARCH_ADAPTIVE and eng.variant == "lane8"
    assert eng.search_stats() == dict(pruned_launches=0, exhaustive_launches=0, reports_read=0, belief=-1, paying_pct=0)
    ws = torch.zeros(aof.workspace_layout(p, base * reps).total_bytes, dtype=torch.uint8, device=gpu_device)

    def call(kind):
        tc, refs = batches[kind]
        blocks, flows, _ = eng.flow_batch(tp, tc, workspace=ws)
        torch.cuda.synchronize()
        assert replicas_equal(aof, blocks, flows, refs) < 0, (kind, eng.search_stats())
        return eng.search_stats()

    for _ in range(3):
        st = call("clean")
    assert st["pruned_launches"] == 3 and st["exhaustive_launches"] == 0 and st["belief"] == 1, st
    assert st["paying_pct"] >= 60, st
    # noise: the first launch still prunes (and reports that it could not), the next ones run the exhaustive kernel,
    # the 16th looks again
    for _ in range(20):
        st = call("noisy")
    assert st["belief"] == 0 and st["paying_pct"] <= 10, st
    assert st["pruned_launches"] == 3 + 1 + 1 and st["exhaustive_launches"] == 18, st
    # clean again: found at the next look, at most 16 launches later
    for _ in range(17):
        st = call("clean")
    assert st["belief"] == 1 and st["pruned_launches"] >= 3 + 2 + 2, st
    before = st["pruned_launches"]
    st = call("clean")
    assert st["pruned_launches"] == before + 1, st
    # without a host wait between the launches the reports arrive late or not at all: same records
    for kind in ("noisy", "clean", "noisy"):
        tc, refs = batches[kind]
        for _ in range(5):
            blocks, flows, _ = eng.flow_batch(tp, tc, workspace=ws)
        torch.cuda.synchronize()
        assert replicas_equal(aof, blocks, flows, refs) < 0, kind
    eng.close()


def test_adaptive8_leaves_small_launches_to_the_exhaustive_kernel_and_prunes_half_pixel_contexts(aof, orc, synth, gpu_device):
    import torch
    W, H = 192, 160
    prevs, curs, _ = synth.make_batch(W, H, 4, 4, 8200, noise=1)
    for kw, reps, pruned in ((dict(), 1, 0), (dict(subpixel=1), 1, 0), (dict(subpixel=1), 720, 3)):
        p = aof.default_params(W, H, **kw)
        po = orc.params_from(p)
        refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(4)]
        eng = aof.FlowEngine(p, 0)
        assert eng.search_mode == aof.SEARCH_ADAPTIVE
        tp, tc = tiled(torch, prevs, reps, gpu_device), tiled(torch, curs, reps, gpu_device)
        sub = torch.full((4 * reps, eng.nblocks(0)), 0x77, dtype=torch.uint8, device=gpu_device) if p.subpixel else None
        for _ in range(3):
            blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
            torch.cuda.synchronize()
            assert replicas_equal(aof, blocks, flows, refs) < 0
            if p.subpixel:
                got = sub.cpu().numpy()
                for i in range(got.shape[0]):
                    assert got[i].tobytes() == refs[i % 4]["subdirs"].tobytes(), i
        st = eng.search_stats()
        assert st["pruned_launches"] == pruned and st["exhaustive_launches"] == 3 - pruned, st
        eng.close()


@pytest.mark.parametrize("case", ["ties", "flat_and_unrelated", "two_level", "mixed"])
def test_adaptive8_judging_kernel_is_exact_on_hard_inputs(aof, orc, synth, gpu_device, case):
    """The path the pruned kernel takes in the ADAPTIVE mode (first chunk of every wave exhaustively, with the verdict
    about the next one from its SADs) on inputs with ties everywhere, flat halves, unrelated frames, a predictor and
    saturating equalisation: large batches of interleaved replicas, so that consecutive chunks of a wave see different
    kinds of pairs and the verdict flips back and forth."""
    import torch
    rng = np.random.default_rng(77)
    W, H = 192, 160
    kw = dict()
    if case == "ties":
        img = np.zeros((H, W), np.uint8); img[:, 0::2] = 200
        img2 = np.zeros((H, W), np.uint8); img2[0::2, :] = 150
        img3 = np.zeros((H, W), np.uint8); img3[0::4, 0::4] = 255
        yy, xx = np.mgrid[0:H, 0:W]
        chk = (((xx // 8 + yy // 8) % 2) * 255).astype(np.uint8)
        prevs = np.stack([img, img2, img3, chk, chk])
        curs = np.stack([img, img2, img3, chk, 255 - chk])
        kw = dict(feature_threshold=0, value_threshold=70000)
    elif case == "flat_and_unrelated":
        prevs, curs, _ = synth.make_batch(W, H, 5, 4, 8300, noise=2)
        prevs[0, : H // 2] = 90; curs[0, : H // 2] = 90                       # whole waves without a live block
        curs[1] = rng.integers(0, 256, curs[1].shape, dtype=np.uint8)         # nothing can be pruned
        curs[2] = prevs[2]                                                     # identical frames
        prevs[3][:] = 17; curs[3][:] = 17                                      # no block passes the gate
    elif case == "two_level":
        kw = dict(pyramid_levels=2, mean_subtract=1)
        prevs, curs, _ = synth.make_batch(W, H, 5, 9, 8400, noise=2, brightness=18)
        curs[4] = np.clip(curs[4].astype(np.int16) + 90, 0, 255).astype(np.uint8)   # the equalisation saturates
    else:
        prevs, curs, _ = synth.make_batch(W, H, 6, 4, 8500, noise=0)
        for i in (1, 3, 4):
            curs[i] = np.clip(curs[i].astype(np.int16) + rng.integers(-30, 31, curs[i].shape), 0, 255).astype(np.uint8)
    p = aof.default_params(W, H, **kw)
    po = orc.params_from(p)
    base = prevs.shape[0]
    refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(base)]
    reps = (4096 * 256 // 437 + base) // base + 1
    tp, tc = tiled(torch, prevs, reps, gpu_device), tiled(torch, curs, reps, gpu_device)
    eng = aof.FlowEngine(p, 0)
    for _ in range(3):   # (nothing known / what the first launch reported / what the second one did)
        blocks, flows, _ = eng.flow_batch(tp, tc)
        torch.cuda.synchronize()
        bad = replicas_equal(aof, blocks, flows, refs)
        assert bad < 0, (case, bad, eng.search_stats())
    st = eng.search_stats()
    assert st["pruned_launches"] >= 1 and st["reports_read"] >= 1, st
    eng.close()


def test_adaptive8_launch_captured_into_a_graph(aof, orc, synth, gpu_device):
    """A launch of the adaptive 8x8 search captured into a hipGraph holds the kernel chosen at capture time (a fresh context:
    the pruned kernel whose first blocks judge; a context that has learnt: the one that starts pruning) and keeps writing its
    reports into the context's pinned words on every replay.  Replays on new frames -- clean, then noisy -- return the
    oracle's records; eager launches behind them go on learning."""
    import torch
    W, H, base, reps = 192, 160, 4, 720
    p = aof.default_params(W, H)
    po = orc.params_from(p)
    n = base * reps
    for learnt in (False, True):
        eng = aof.FlowEngine(p, 0)
        prevs, curs, _ = synth.make_batch(W, H, base, 4, 8700, noise=1)
        prev, cur = tiled(torch, prevs, reps, gpu_device), tiled(torch, curs, reps, gpu_device)
        blocks = torch.zeros((n, eng.nblocks(0)), dtype=torch.int32, device=gpu_device)
        flows = torch.zeros((n, 16), dtype=torch.uint8, device=gpu_device)
        ws = torch.zeros(aof.workspace_layout(p, n).total_bytes, dtype=torch.uint8, device=gpu_device)
        if learnt:
            for _ in range(2):
                eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
                torch.cuda.synchronize()
            assert eng.search_stats()["belief"] == 1
        side = torch.cuda.Stream(gpu_device)
        side.wait_stream(torch.cuda.current_stream(gpu_device))
        with torch.cuda.stream(side):   # (warm-up on the capture stream, as torch asks for)
            eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
        launches = eng.search_stats()["pruned_launches"]
        rng = np.random.default_rng(3)
        for rep in range(4):
            hp, hc, _ = synth.make_batch(W, H, base, 4, 8710 + rep, noise=0)
            if rep >= 2:
                hc = np.clip(hc.astype(np.int16) + rng.integers(-40, 41, hc.shape), 0, 255).astype(np.uint8)
            prev.copy_(tiled(torch, hp, reps, gpu_device)); cur.copy_(tiled(torch, hc, reps, gpu_device))
            blocks.zero_(); flows.zero_()
            g.replay()
            torch.cuda.synchronize()
            refs = [orc.flow_pair(po, hp[i], hc[i]) for i in range(base)]
            assert replicas_equal(aof, blocks, flows, refs) < 0, (learnt, rep)
        assert eng.search_stats()["pruned_launches"] == launches, "replays are not launches of the library"
        # eager launches on the noisy frames the last replays left: the context reads the replays' reports and its own
        for _ in range(3):
            eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
            torch.cuda.synchronize()
            assert replicas_equal(aof, blocks, flows, refs) < 0
        assert eng.search_stats()["belief"] == 0, eng.search_stats()
        del g
        eng.close()


# ---- round 5: start-row vote, launch-size threshold, belief hint, the column walk that reduces in its launch ----

@pytest.mark.parametrize("shift", [(0, -4), (3, 4), (-4, 0), (1, -3), (-2, 2)])
def test_pruned_walk_votes_for_its_start_row(aof, orc, synth, gpu_device, shift):
    """AOF_SEARCH_PRUNED without a predictor: the first block of every walk votes for the dy row to start in
    (vote_start_row) -- every dy row from the rim to the centre, on clean and noisy frames, one pair with a second
    motion in its lower half (the vote is a majority, the search stays exact) and one whose votes tie."""
    W, H = 320, 200
    p = aof.default_params(W, H)
    prevs, curs = [], []
    for k, noise in enumerate((0, 6, 30)):
        a, b, _ = synth.make_pair(W, H, 4, 9100 + k, shift=shift, noise=noise)
        prevs.append(a); curs.append(b)
    a, b, _ = synth.make_pair(W, H, 4, 9110, shift=shift)
    _, b2, _ = synth.make_pair(W, H, 4, 9110, shift=(-shift[0], -shift[1]))
    b = b.copy(); b[H // 2:] = b2[H // 2:]                       # two motions in one pair
    prevs.append(a); curs.append(b)
    both_modes_match_oracle(aof, orc, p, np.stack(prevs), np.stack(curs), gpu_device)


def test_adaptive8_prunes_from_2048_chunks_on_and_takes_a_hint(aof, orc, synth, gpu_device):
    """kPruneMinChunks = 2 048: a launch of 2 100 chunks prunes (round 4: 4 096), one of 1 900 does not; a fresh context
    that is TOLD its images prune (aof_set_search_belief) starts every walk in the pruned code from its first launch
    on, one told that they do not runs the exhaustive kernel -- the records are the oracle's either way."""
    import torch
    W, H, base = 192, 160, 4                       # 437 blocks per pair
    p = aof.default_params(W, H)
    po = orc.params_from(p)
    prevs, curs, _ = synth.make_batch(W, H, base, 4, 9200, noise=1)
    refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(base)]
    for pairs, pruned in ((1232, 1), (1112, 0)):   # 2 103 / 1 898 chunks of 256 blocks
        eng = aof.FlowEngine(p, 0)
        blocks, flows, _ = eng.flow_batch(tiled(torch, prevs, pairs // base, gpu_device), tiled(torch, curs, pairs // base, gpu_device))
        torch.cuda.synchronize()
        assert replicas_equal(aof, blocks, flows, refs) < 0
        st = eng.search_stats()
        assert st["pruned_launches"] == pruned and st["exhaustive_launches"] == 1 - pruned, (pairs, st)
        eng.close()
    tp, tc = tiled(torch, prevs, 400, gpu_device), tiled(torch, curs, 400, gpu_device)
    for belief, want in ((1, dict(pruned_launches=1, exhaustive_launches=0)), (0, dict(pruned_launches=0, exhaustive_launches=1))):
        eng = aof.FlowEngine(p, 0)
        eng.set_search_belief(belief)
        assert eng.search_stats()["belief"] == belief
        blocks, flows, _ = eng.flow_batch(tp, tc)
        torch.cuda.synchronize()
        assert replicas_equal(aof, blocks, flows, refs) < 0, belief
        st = eng.search_stats()
        assert {k: st[k] for k in want} == want, (belief, st)
        eng.close()
    eng = aof.FlowEngine(p, 0)
    with pytest.raises(aof.AofError):
        eng.set_search_belief(2)
    eng.close()


def test_column_walk_that_reduces_in_its_launch_on_a_large_adaptive_launch(aof, orc, synth, gpu_device):
    """aof_set_reduce_fusion(1) on a context whose launches prune: k_flow_lane8_cols -- walks of several block rows whose
    agreeing votes are added once (WalkVotes), walks whose motion changes half-way (flush), pairs without any vote --
    flow records against K3's and the oracle's, over repeated launches (the vote memory is zero again after each)."""
    import torch
    W, H, base = 256, 208, 6                       # 31 x 25 = 775 blocks per pair
    p = aof.default_params(W, H)
    po = orc.params_from(p)
    prevs, curs, _ = synth.make_batch(W, H, base, 4, 9300, noise=2)
    _, other, _ = synth.make_batch(W, H, base, 4, 9300, noise=2, shift=(-3, 2))
    curs[1, H // 3:] = other[1, H // 3:]           # the motion changes inside every column's walk
    prevs[2] = 7                                   # flat: every block gated, arrivals without votes
    rng = np.random.default_rng(3)
    curs[3] = rng.integers(0, 256, curs[3].shape, dtype=np.uint8)   # unrelated: votes all over the histogram
    refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(base)]
    reps = 120                                     # 720 pairs: 2 180 chunks, the launch prunes
    tp, tc = tiled(torch, prevs, reps, gpu_device), tiled(torch, curs, reps, gpu_device)
    eng = aof.FlowEngine(p, 0)
    eng.set_reduce_fusion(True)
    for launch in range(4):
        blocks, flows, _ = eng.flow_batch(tp, tc)
        torch.cuda.synchronize()
        assert replicas_equal(aof, blocks, flows, refs) < 0, (launch, eng.search_stats())
    assert eng.search_stats()["pruned_launches"] == 4
    eng.close()


@pytest.mark.parametrize("kw", [dict(pyramid_levels=2, mean_subtract=1), dict(pyramid_levels=2), dict(subpixel=1)])
def test_column_walk_under_every_window_alignment(aof, orc, synth, gpu_device, kw):
    """Round 5: a pair whose windows do not start on a dword (predictor px not a multiple of four; origin 5 of the half-pixel
    grid) loads its window rows from the dword below and takes the missing bytes from the next column through a wave shift.
    Every residue of px mod 4 with both signs, windows pushed to the left and right rim (columns that fall outside the frame,
    a row's last column, lane 63 of a wave), the last pair of the batch (whose rows end where the arrays end), one and two
    motions per pair -- pruned and exhaustive against the oracle."""
    W, H = 656, 208                                   # 80 x 25 blocks: more than one wave per block row, lane 63 inside a row
    p = aof.default_params(W, H, **kw)
    two = p.pyramid_levels == 2
    reach = 9 if two else 4
    prevs, curs = [], []
    shifts = list(range(-reach, reach + 1))
    for k, sx in enumerate(shifts):
        sy = (k * 5) % (2 * reach + 1) - reach
        a, b, _ = synth.make_pair(W, H, reach, 9600 + k, shift=(sx, sy), noise=k % 3, brightness=6 if p.mean_subtract else 0)
        prevs.append(a); curs.append(b)
    a, b, _ = synth.make_pair(W, H, reach, 9650, shift=(reach - 2, -reach))
    _, b2, _ = synth.make_pair(W, H, reach, 9650, shift=(-reach, reach - 1))
    b = b.copy(); b[:, W // 2:] = b2[:, W // 2:]       # two motions side by side in the LAST pair
    prevs.append(a); curs.append(b)
    check = half_pixel_modes_match_oracle if p.subpixel else both_modes_match_oracle
    check(aof, orc, p, np.stack(prevs), np.stack(curs), gpu_device)


@pytest.mark.parametrize("kw", [dict(), dict(subpixel=1), dict(pyramid_levels=2, mean_subtract=1)])
def test_perfect_matches_in_different_rows_of_one_wave(aof, orc, synth, gpu_device, kw):
    """Noise-free frames cut from patches that moved differently: the lanes of a wave find SAD 0 in different dy rows,
    some patches carry a period-2 texture (SAD 0 in several rows: the tie goes to the first in scan order), one is
    displaced by half a pixel and one is noise -- the rows behind a perfect match that go unsummed (aof_lane8.hpp,
    pruned_row's `zero`) and the refinement that ends early (aof_refine.hpp, nobody_can_win) must not change a byte."""
    w, h, reach = 640, 160, 9 if kw.get("pyramid_levels") == 2 else 4
    p = aof.default_params(w, h, **kw)
    rng = np.random.default_rng(77)
    prevs, curs = [], []
    for k in range(3):
        prev = np.zeros((h, w), np.uint8)
        cur = np.zeros((h, w), np.uint8)
        for i, x0 in enumerate(range(0, w, 80)):   # eight vertical stripes of ten block columns: several per wave
            s = int(reach)
            shift = [(0, -s), (s, s), (-1, 2), (3, -3), (-s, 0), (2, 1), (0, 0), (1, -s + 1)][(i + k) % 8]
            half = (1, 0) if (i + k) % 8 == 5 and kw.get("subpixel") else (0, 0)
            a, b, _ = synth.make_pair(80, h, reach, 500 + 8 * k + i, shift=shift, half=half)
            if (i + k) % 8 == 3:                   # period 2 in x and y: ties between rows and columns
                a = np.zeros((h, 80), np.uint8)
                a[0::2, 0::2] = 220
                b = a.copy()
            if (i + k) % 8 == 6:                   # unrelated noise: no perfect match in these lanes
                b = rng.integers(0, 256, b.shape, dtype=np.uint8)
            prev[:, x0:x0 + 80], cur[:, x0:x0 + 80] = a, b
        prevs.append(prev)
        curs.append(cur)
    both_modes_match_oracle(aof, orc, p, np.stack(prevs), np.stack(curs), gpu_device)
