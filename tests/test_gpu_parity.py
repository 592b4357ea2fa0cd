"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs.  Integer/byte work: the bar is bit-exact
(block records compared as raw u32, flow records as raw 16 bytes).

"Oracle" = this repo's restatement of the published PX4Flow algorithm + the
build-defined extensions (parity unpinned: upstream PX4 source unavailable)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_gpu(aof, p, prevs, curs, device, force_generic=False, want_ws=False, fused_reduce=False, split_coarse=False):
    import torch
    eng = aof.FlowEngine(p, 0)
    if force_generic:
        eng.force_generic(True)
    if split_coarse:
        eng.set_split_coarse(True)
    if fused_reduce:   # the flat lane8 search reduces in its own launch (grids of more than 256 blocks)
        eng.set_reduce_fusion(True)
    tp = torch.from_numpy(np.ascontiguousarray(prevs)).to(device)
    tc = torch.from_numpy(np.ascontiguousarray(curs)).to(device)
    blocks, flows, ws = eng.flow_batch(tp, tc)
    torch.cuda.synchronize()
    out = dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows), variant=eng.variant, eng=eng)
    if want_ws:
        out["ws"] = ws.cpu().numpy()
    return out


def check_against_oracle(aof, orc, p, prevs, curs, got, subdirs=None):
    po = orc.params_from(p)
    for i in range(prevs.shape[0]):
        ref = orc.flow_pair(po, prevs[i], curs[i])
        gb, rb = got["blocks"][i], ref["blocks"]
        if gb.tobytes() != rb.tobytes():
            bad = np.nonzero(gb.view(np.uint32) != rb.view(np.uint32))[0]
            raise AssertionError(f"pair {i}: {bad.size} block records differ, first {bad[:5]}: "
                                 f"gpu {gb[bad[:5]]} oracle {rb[bad[:5]]}")
        assert got["flows"][i].tobytes() == ref["flow"].tobytes(), (i, got["flows"][i], ref["flow"])
        if subdirs is not None:
            assert np.array_equal(subdirs[i], ref["subdirs"]), i


# ---- BASELINE configs ---------------------------------------------------------

def test_c1_64x64_px4flow_grid(aof, orc, synth, gpu_device):
    """configs[0]: 64x64, 8x8 SAD +-4, the published sparse grid with half-pixel refinement."""
    import torch
    p = aof.px4flow_params(64, 64)
    prevs, curs, shifts = synth.make_batch(64, 64, 12, 4, 0, noise=2)
    for i, half in enumerate([(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1)]):
        prevs[i], curs[i], _ = synth.make_pair(64, 64, 4, i, shift=(1 - i % 3, i % 2), half=half)
    tp, tc = torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device)
    for generic in (False, True):   # lane-per-block kernel + refinement pass, then the generic kernel
        eng = aof.FlowEngine(p, 0)
        assert eng.variant == "lane8"
        eng.force_generic(generic)
        sub = torch.full((12, 25), 99, dtype=torch.uint8, device=gpu_device)
        blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
        torch.cuda.synchronize()
        got = dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows))
        check_against_oracle(aof, orc, p, prevs, curs, got, subdirs=sub.cpu().numpy())
        seen = set(np.unique(sub.cpu().numpy()[:8]))
        assert seen >= {0, 1, 2, 3, 4, 5, 6, 7}, f"half-pixel shifted pairs must exercise every direction: {seen}"


def test_c1_64x64_dense(aof, orc, synth, gpu_device):
    p = aof.default_params(64, 64)
    prevs, curs, shifts = synth.make_batch(64, 64, 16, 4, 100)
    got = run_gpu(aof, p, prevs, curs, gpu_device)
    assert got["variant"] == "lane8"
    check_against_oracle(aof, orc, p, prevs, curs, got)
    # analytic known answer, independent of any implementation
    for i in range(16):
        assert (got["blocks"][i]["dx"] == shifts[i, 0]).all() and (got["blocks"][i]["dy"] == shifts[i, 1]).all()
        assert (got["blocks"][i]["sad"] == 0).all()
        assert got["flows"][i]["flow_x"] == shifts[i, 0] and got["flows"][i]["quality"] == 255


@pytest.mark.parametrize("noise,brightness", [(0, 0), (4, 0), (12, -20)])
def test_c2_vga_8x8_bit_exact(aof, orc, synth, gpu_device, noise, brightness):
    """configs[1]: 640x480, 8x8 SAD +-4, 1 GPU, bit-exact vs CPU."""
    p = aof.default_params(640, 480)
    prevs, curs, shifts = synth.make_batch(640, 480, 5, 4, 200 + noise, noise=noise, brightness=brightness)
    got = run_gpu(aof, p, prevs, curs, gpu_device)
    assert got["variant"] == "lane8"
    assert got["blocks"].shape == (5, 4661)
    check_against_oracle(aof, orc, p, prevs, curs, got)
    fused = run_gpu(aof, p, prevs, curs, gpu_device, fused_reduce=True)   # reduction inside the search launch
    check_against_oracle(aof, orc, p, prevs, curs, fused)


def test_c2_generic_kernel_agrees(aof, orc, synth, gpu_device):
    p = aof.default_params(640, 480)
    prevs, curs, _ = synth.make_batch(640, 480, 2, 4, 300, noise=6)
    a = run_gpu(aof, p, prevs, curs, gpu_device)
    b = run_gpu(aof, p, prevs, curs, gpu_device, force_generic=True)
    assert b["variant"] == "generic"
    assert a["blocks"].tobytes() == b["blocks"].tobytes() and a["flows"].tobytes() == b["flows"].tobytes()
    check_against_oracle(aof, orc, p, prevs, curs, b)


@pytest.mark.parametrize("path", ["fused", "split", "generic"])
def test_c3_vga_two_level_mean_subtracted(aof, orc, synth, gpu_device, path):
    """configs[2]: 640x480, 2-level mean-subtracted pyramid + 4x4 gate / histogram filter.
    fused = the coarse passes in one kernel (k_coarse, level-1 frames in LDS only), split = K1 /
    level-1 search / level-1 reduce as separate kernels, generic = the wave-per-block kernel."""
    p = aof.default_params(640, 480, pyramid_levels=2, mean_subtract=1)
    # two levels reach 2S at level 1 (+ the level-0 refinement): keep |shift| <= 2S+1 = 9
    prevs, curs, shifts = synth.make_batch(640, 480, 4, 9, 401, noise=3, brightness=13)
    got = run_gpu(aof, p, prevs, curs, gpu_device, force_generic=path == "generic", want_ws=True,
                  split_coarse=path == "split")
    check_against_oracle(aof, orc, p, prevs, curs, got)
    f = got["flows"]
    assert (f["flags"] == 3).all()
    big = np.abs(shifts).max(axis=1) > 4
    assert big.any(), "the batch must contain shifts only the pyramid can reach"
    assert np.array_equal(f["flow_x"], shifts[:, 0].astype(np.float32))
    assert np.array_equal(f["flow_y"], shifts[:, 1].astype(np.float32))
    # K1 intermediates against the oracle's building blocks
    L = aof.workspace_layout(p, 4)
    ws = got["ws"]
    sums = ws[L.sums:L.sums + 4 * 16].view(np.uint32).reshape(4, 2, 2)
    l1p = ws[L.l1_prev:L.l1_prev + 4 * 320 * 240].reshape(4, 240, 320)
    l1c = ws[L.l1_cur:L.l1_cur + 4 * 320 * 240].reshape(4, 240, 320)
    for i in range(4):
        rp, rc = orc.pyramid_down(prevs[i]), orc.pyramid_down(curs[i])
        if path != "fused":   # the fused kernel keeps the level-1 frames in LDS
            assert np.array_equal(l1p[i], rp)
            assert np.array_equal(l1c[i], rc)
        assert sums[i, 0, 0] == prevs[i].sum(dtype=np.uint64) and sums[i, 1, 0] == curs[i].sum(dtype=np.uint64)
        assert sums[i, 0, 1] == rp.sum(dtype=np.uint64) and sums[i, 1, 1] == rc.sum(dtype=np.uint64)
    # level-1 records against the oracle
    po = orc.params_from(p)
    nb1 = 39 * 29
    b1 = ws[L.l1_blocks:L.l1_blocks + 4 * nb1 * 4].view(aof.BLOCK_DTYPE).reshape(4, nb1)
    for i in range(4):
        ref = orc.flow_pair(po, prevs[i], curs[i], want_l1=True)
        assert b1[i].tobytes() == ref["blocks_l1"].tobytes()


@pytest.mark.parametrize("kw,w,h", [
    (dict(px4=1, pyramid_levels=2, mean_subtract=1), 128, 128),   # what OpticalFlowOpenCV runs per call
    (dict(px4=1, pyramid_levels=2, mean_subtract=1), 64, 64),
    (dict(px4=1, pyramid_levels=2), 96, 80),
    (dict(px4=1, pyramid_levels=2, mean_subtract=1, hist_filter=0), 112, 66),   # odd level-1 height
    (dict(px4=1, pyramid_levels=2, mean_subtract=1, subpixel=0), 128, 96),
    (dict(pyramid_levels=2, mean_subtract=1), 128, 96),           # dense grid, 15x11 = 165 blocks at level 0
    (dict(pyramid_levels=2, mean_subtract=1, subpixel=1), 144, 112),            # dense, 16x12 = 192 blocks
    (dict(pyramid_levels=2, mean_subtract=1, subpixel=1), 256, 224),            # 140 KB of LDS; level 0 falls back? no: 31x27 blocks
    (dict(px4=1, pyramid_levels=2, mean_subtract=1), 256, 224),   # sparse grid on 143 KB of LDS (dynamic LDS beyond 64 KB)
    (dict(px4=1), 320, 240),                                      # one level, 154 KB of LDS
    (dict(px4=1), 64, 64),                                        # one level: what OpticalFlowPX4 runs per call (configs[0])
    (dict(px4=1, mean_subtract=1), 128, 128),
    (dict(px4=1, subpixel=0, hist_filter=0), 80, 49),             # odd height
    (dict(subpixel=1), 96, 96),                                   # dense, 11x11
])
@pytest.mark.parametrize("n", [1, 7, 128])
def test_small_pairs_in_one_launch(aof, orc, synth, gpu_device, kw, w, h, n):
    """Small pairs (frames fit LDS, grids <= 256 blocks, <= 128 pairs) run sums, pyramid, searches and
    reductions in ONE kernel out of LDS (k_flow_small).  Outputs against the oracle; block records,
    flows and the workspace (sums, level-1 records, directions and flows) against the separate
    kernels, byte for byte."""
    kw = dict(kw)
    mk = aof.px4flow_params if kw.pop("px4", 0) else aof.default_params
    p = mk(w, h, **kw)
    two = p.pyramid_levels == 2
    prevs, curs, shifts = synth.make_batch(w, h, n, 9 if two else 4, 7000 + n, noise=3, brightness=9)
    one = run_gpu(aof, p, prevs, curs, gpu_device, want_ws=True)
    sep = run_gpu(aof, p, prevs, curs, gpu_device, want_ws=True, split_coarse=True)
    g0 = aof.grid(p, 0)
    fits = g0[4] * g0[5] <= 256
    if n == 7:   # one launch: no pyramid kernel, no level-1 kernels, no reduction kernel
        import torch
        eng = aof.FlowEngine(p, 0)
        eng.set_profiling(True)
        eng.flow_batch(torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device))
        torch.cuda.synchronize()
        launches = [len(eng.profile_ms(k)) for k in range(5)]
        assert (launches == [0, 0, 0, 1, 0]) == fits, launches
    assert one["blocks"].tobytes() == sep["blocks"].tobytes()
    assert one["flows"].tobytes() == sep["flows"].tobytes()
    L = aof.workspace_layout(p, n)
    regions = []
    if two:
        g1 = aof.grid(p, 1)
        nb1 = g1[4] * g1[5]
        regions += [(L.l1_blocks, n * nb1 * 4), (L.l1_flows, n * 16)]
        if p.subpixel:
            regions.append((L.l1_subdirs, n * nb1))
    if p.mean_subtract:   # [pair][frame][level]; one level: only the level-0 sums mean anything
        a, b = (x["ws"][L.sums:L.sums + n * 16].view(np.uint32).reshape(n, 2, 2) for x in (one, sep))
        assert np.array_equal(a, b) if two else np.array_equal(a[:, :, 0], b[:, :, 0])
    for off, size in regions:
        assert one["ws"][off:off + size].tobytes() == sep["ws"][off:off + size].tobytes(), (off, size)
    sample = sorted(set([0, n // 2, n - 1]))
    check_against_oracle(aof, orc, p, prevs[sample], curs[sample],
                         dict(blocks=one["blocks"][sample], flows=one["flows"][sample]))


@pytest.mark.parametrize("levels", [1, 2])
@pytest.mark.parametrize("geometry", ["sparse_step8", "dense_vga_chunked"])
def test_every_mode_stays_inside_the_workspace(aof, orc, synth, gpu_device, levels, geometry):
    """Guard bytes behind the workspace, every search mode and kernel choice: a published sparse grid
    whose step comes out as 8 from origin 5 (the shape of a dense grid -- a strip kernel that has since
    been removed once took it and wrote 880 bytes past a layout sized by grid mode), and a grid large
    enough for the two-step reduction, whose per-chunk histograms live in the workspace."""
    import torch
    if geometry == "sparse_step8":
        p = aof.px4flow_params(64, 64, num_blocks=6, pyramid_levels=levels, min_valid=0)
        assert aof.grid(p, 0)[:4] == (5, 5, 8, 8)
        W, H, reach = 64, 64, 4
    else:
        W, H, reach = 1024, 768, 4      # 127 x 95 = 12 065 blocks > 8 192: k_reduce_chunk + k_reduce
        p = aof.default_params(W, H, pyramid_levels=levels, min_valid=0)
        assert aof.workspace_layout(p, 1).total_bytes > aof.workspace_layout(p, 1).l0_hist
    n = 2
    prevs, curs, _ = synth.make_batch(W, H, n, reach, 4242, noise=2)
    tp, tc = torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device)
    L = aof.workspace_layout(p, n)
    po = orc.params_from(p)
    refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(n)]
    for mode in (aof.SEARCH_EXHAUSTIVE, aof.SEARCH_PRUNED, "generic", "split", "fused_reduce"):
        eng = aof.FlowEngine(p, 0)
        if mode == "generic":
            eng.force_generic(True)
        elif mode == "split":
            eng.set_split_coarse(True)
        elif mode == "fused_reduce":
            eng.set_reduce_fusion(True)
        else:
            eng.set_search_mode(mode)
        big = torch.full((L.total_bytes + 8192,), 0xAB, dtype=torch.uint8, device=gpu_device)
        blocks, flows, _ = eng.flow_batch(tp, tc, workspace=big[:L.total_bytes])
        torch.cuda.synchronize()
        assert bool((big[L.total_bytes:] == 0xAB).all()), f"mode {mode} wrote behind the workspace"
        assert np.array_equal(tc.cpu().numpy(), curs) and np.array_equal(tp.cpu().numpy(), prevs)
        gb, gf = aof.blocks_view(blocks), aof.flows_view(flows)
        for i in range(n):
            assert gb[i].tobytes() == refs[i]["blocks"].tobytes() and gf[i].tobytes() == refs[i]["flow"].tobytes(), (mode, i)
        eng.close()


def test_c5_1280x960_16x16_search8(aof, orc, synth, gpu_device):
    """configs[4]: 1280x960, 16x16 SAD +-8 (LDS-tile stress)."""
    p = aof.default_params(1280, 960, tile=16, search=8, value_threshold=12000)
    prevs, curs, shifts = synth.make_batch(1280, 960, 2, 8, 500, noise=3)
    got = run_gpu(aof, p, prevs, curs, gpu_device)
    assert got["blocks"].shape == (2, 4661)
    check_against_oracle(aof, orc, p, prevs, curs, got)
    assert np.array_equal(got["flows"]["flow_x"], shifts[:, 0].astype(np.float32))
    # (a fresh 16x16 context searches in the adaptive mode; the exhaustive scan on its own as well)
    assert got["eng"].search_mode == aof.SEARCH_ADAPTIVE
    import torch
    eng = aof.FlowEngine(p, 0)
    eng.set_search_mode(aof.SEARCH_EXHAUSTIVE)
    blocks, flows, _ = eng.flow_batch(torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device))
    torch.cuda.synchronize()
    check_against_oracle(aof, orc, p, prevs, curs, dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows)))
    eng.close()


def test_c4_batch_properties_at_full_size(aof, orc, synth, gpu_device):
    """configs[3] per-GPU share: 128 independent VGA pairs in one launch.  The
    oracle checks a sample; every pair is checked through the analytic property
    (pure translation -> exact shift, SAD 0, quality 255)."""
    import torch
    p = aof.default_params(640, 480)
    n = 128
    prevs, curs, shifts = synth.make_batch(640, 480, n, 4, 1000)
    got = run_gpu(aof, p, prevs, curs, gpu_device)
    b, f = got["blocks"], got["flows"]
    assert (b["sad"] == 0).all()
    assert (b["dx"] == shifts[:, 0:1]).all() and (b["dy"] == shifts[:, 1:2]).all()
    assert np.array_equal(f["flow_x"], shifts[:, 0].astype(np.float32))
    assert np.array_equal(f["flow_y"], shifts[:, 1].astype(np.float32))
    assert (f["quality"] == 255).all() and (f["count"] == 4661).all()
    sample = [0, 1, 63, 127]
    check_against_oracle(aof, orc, p, prevs[sample], curs[sample],
                         dict(blocks=b[sample], flows=f[sample]))
    # idempotence: a second run over the same resident batch gives the same bytes
    again = run_gpu(aof, p, prevs, curs, gpu_device)
    assert again["blocks"].tobytes() == b.tobytes() and again["flows"].tobytes() == f.tobytes()


def test_c4_full_batch_of_1024_pairs_through_the_shard_and_gather_path(aof, orc, synth, gpu_device):
    """configs[3] at its stated size: 1 024 independent VGA pairs, sharded eight ways exactly as
    the ranks of an 8-GPU node would (batch.shard_range), every shard through the HIP path, the
    16-byte flow records joined in pair order by batch.gather_flows (one process here, so the
    gather is the identity; the world-size-2 gloo test covers the collective itself).  Every
    pair is checked through the analytic property of a pure translation, a sample against
    the oracle."""
    import importlib
    import torch
    batch = importlib.import_module("aero_optical_flow_amd.batch")
    p = aof.default_params(640, 480)
    n_total, world = 1024, 8
    canvas_prev, canvas_cur, shifts = synth.make_batch(640, 480, 64, 4, 7000)   # 64 distinct pairs, tiled
    eng = aof.FlowEngine(p, 0)
    gathered = []
    for rank in range(world):
        b, e = batch.shard_range(n_total, rank, world)
        idx = np.arange(b, e) % 64
        tp = torch.from_numpy(canvas_prev[idx]).to(gpu_device)
        tc = torch.from_numpy(canvas_cur[idx]).to(gpu_device)
        blocks, flows, _ = eng.flow_batch(tp, tc)
        full = batch.gather_flows(flows, e - b)          # world of one: the local records
        torch.cuda.synchronize()
        gathered.append(full.cpu())
        bv = aof.blocks_view(blocks)
        assert (bv["sad"] == 0).all()
        assert (bv["dx"] == shifts[idx, 0:1]).all() and (bv["dy"] == shifts[idx, 1:2]).all()
        if rank in (0, 7):
            sample = [0, 77, 127]
            check_against_oracle(aof, orc, p, canvas_prev[idx[sample]], canvas_cur[idx[sample]],
                                 dict(blocks=bv[sample], flows=aof.flows_view(flows)[sample]))
    f = aof.flows_view(torch.cat(gathered, dim=0))
    want = shifts[np.arange(n_total) % 64]
    assert f.shape[0] == n_total
    assert np.array_equal(f["flow_x"], want[:, 0].astype(np.float32))
    assert np.array_equal(f["flow_y"], want[:, 1].astype(np.float32))
    assert (f["quality"] == 255).all() and (f["count"] == 4661).all()
    # and the whole batch in ONE launch gives the same 16 KB of records
    idx = np.arange(n_total) % 64
    tp = torch.from_numpy(canvas_prev[idx]).to(gpu_device)
    tc = torch.from_numpy(canvas_cur[idx]).to(gpu_device)
    _, flows, _ = eng.flow_batch(tp, tc)
    torch.cuda.synchronize()
    assert aof.flows_view(flows).tobytes() == f.tobytes()


@pytest.mark.parametrize("kw", [dict(pyramid_levels=2, mean_subtract=1), dict(pyramid_levels=2, subpixel=1),
                                dict(pyramid_levels=2, mean_subtract=1, tile=16, search=8, value_threshold=12000),
                                dict(mean_subtract=1, size=(320, 240)),                       # K1 (sums) + flat lane8 + K3
                                dict(pyramid_levels=2, mean_subtract=1, size=(328, 240)),     # K1 / level-1 / level-0 as separate kernels
                                dict(pyramid_levels=2, mean_subtract=1, size=(320, 240))])    # k_coarse + flat lane8
def test_two_level_batch_replays_from_a_graph(aof, orc, synth, gpu_device, kw):
    """A batch call allocates nothing and never synchronises: captured into a hipGraph and replayed on
    NEW frames in the same buffers it returns the oracle's records every time.  (The 16x16 two-level case
    found that a captured hipMemsetAsync does not replay correctly on ROCm 7.2: the pixel sums are
    zeroed by a kernel since.)"""
    import torch
    kw = dict(kw)
    W, H = kw.pop("size", (192, 160) if kw.get("tile") == 16 else (128, 96))
    p = aof.default_params(W, H, **kw)
    n = 10
    hp, hc, _ = synth.make_batch(W, H, n, 9, 5200, noise=3, brightness=5 if kw.get("mean_subtract") else 0)
    prev = torch.from_numpy(hp).to(gpu_device)
    cur = torch.from_numpy(hc).to(gpu_device)
    eng = aof.FlowEngine(p, 0)
    b0, f0, _ = eng.flow_batch(prev, cur)
    torch.cuda.synchronize()
    check_against_oracle(aof, orc, p, hp, hc, dict(blocks=aof.blocks_view(b0), flows=aof.flows_view(f0)))
    blocks = torch.zeros_like(b0)
    flows = torch.zeros_like(f0)
    ws = torch.zeros(aof.workspace_layout(p, n).total_bytes, dtype=torch.uint8, device=gpu_device)
    eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(blocks, b0) and torch.equal(flows, f0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
    for rep in range(3):
        hp, hc, _ = synth.make_batch(W, H, n, 9, 5300 + rep, noise=3, brightness=5 if kw.get("mean_subtract") else 0)
        prev.copy_(torch.from_numpy(hp))
        cur.copy_(torch.from_numpy(hc))
        g.replay()
        torch.cuda.synchronize()
        check_against_oracle(aof, orc, p, hp, hc, dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows)))


@pytest.mark.parametrize("shape", [(160, 128), (640, 480), (320, 66), (96, 96), (672, 464), (48, 32)])
@pytest.mark.parametrize("kw", [dict(), dict(mean_subtract=1), dict(mean_subtract=1, hist_filter=0, min_valid=0)])
def test_fused_coarse_kernel_equals_the_split_kernels(aof, orc, synth, gpu_device, shape, kw):
    """k_coarse (sums + pyramid + level-1 search + predictor in one workgroup per pair, level-1
    frames in LDS) against the separate K1 / K2 / K3 kernels and the oracle: level-1 records,
    predictors, pixel sums and the final records, on textured, flat, saturated and noisy pairs."""
    W, H = shape
    p = aof.default_params(W, H, pyramid_levels=2, **kw)
    n = 7
    prevs, curs, _ = synth.make_batch(W, H, n, 9, 8100 + W, noise=4, brightness=11 if kw else 0)
    prevs[1, : H // 2] = 7                      # flat upper half: gated blocks
    curs[1, : H // 2] = 7
    curs[2] = np.minimum(curs[2].astype(np.int32) * 2, 255).astype(np.uint8)   # saturated: the clamp bites
    prevs[3] = 255 - prevs[3]
    curs[3] = 255 - curs[3]
    curs[4] = prevs[4]                          # identical frames
    prevs[5] = 0                                # black previous frame: everything gated
    outs = {}
    for split in (False, True):
        got = run_gpu(aof, p, prevs, curs, gpu_device, want_ws=True, split_coarse=split)
        L = aof.workspace_layout(p, n)
        nb1 = got["eng"].nblocks(1)
        ws = got["ws"]
        outs[split] = dict(blocks=got["blocks"].tobytes(), flows=got["flows"].tobytes(),
                           b1=ws[L.l1_blocks:L.l1_blocks + n * nb1 * 4].tobytes(),
                           f1=ws[L.l1_flows:L.l1_flows + n * 16].tobytes(),
                           sums=ws[L.sums:L.sums + n * 16].tobytes() if kw else b"")
        if not split:
            check_against_oracle(aof, orc, p, prevs, curs, got)
    for k in outs[False]:
        assert outs[False][k] == outs[True][k], f"{k} differs between the fused and the split coarse passes"


def test_fused_coarse_kernel_walks_many_pairs_per_workgroup(aof, orc, synth, gpu_device):
    """k_coarse is persistent (one workgroup per CU walks pairs i, i + CUs, ...) and requests the
    next pair's first frame rows while it still searches the current one: 1 100 small pairs (four
    to five per workgroup, ragged tail), every one against the split kernels, a sample against
    the oracle."""
    import torch
    W, H = 96, 64
    p = aof.default_params(W, H, pyramid_levels=2, mean_subtract=1, min_valid=0)
    n = 1100
    hp, hc, _ = synth.make_batch(W, H, 50, 6, 9100, noise=3, brightness=7)
    idx = (np.arange(n) * 7) % 50
    prev = torch.from_numpy(hp[idx]).to(gpu_device)
    cur = torch.from_numpy(hc[(idx + (np.arange(n) // 50)) % 50]).to(gpu_device)   # (also mismatched pairs)
    outs = []
    for split in (False, True):
        eng = aof.FlowEngine(p, 0)
        eng.set_split_coarse(split)
        blocks, flows, ws = eng.flow_batch(prev, cur)
        torch.cuda.synchronize()
        L = aof.workspace_layout(p, n)
        w = ws.cpu().numpy()
        outs.append((blocks.cpu().numpy().tobytes(), flows.cpu().numpy().tobytes(),
                     w[L.l1_flows:L.l1_flows + 16 * n].tobytes(), w[L.sums:L.sums + 16 * n].tobytes()))
        if not split:
            sample = [0, 255, 256, 257, 511, 512, 1023, 1024, 1099]
            check_against_oracle(aof, orc, p, prev[sample].cpu().numpy(), cur[sample].cpu().numpy(),
                                 dict(blocks=aof.blocks_view(blocks)[sample], flows=aof.flows_view(flows)[sample]))
    for k, name in enumerate(("records", "flows", "predictors", "sums")):
        assert outs[0][k] == outs[1][k], f"{name} differ between the fused and the split coarse passes"


def test_fused_coarse_kernel_at_vga_beyond_the_first_generation(aof, orc, synth, gpu_device):
    """k_coarse at its BASELINE size (configs[2]: 640x480, 150 KB of level-1 frames in LDS) and past the
    first generation of workgroups: 600 VGA pairs on 256 CUs, which engages the three-group start-up
    stagger (from 512 pairs), the row prefetch across pairs and the two alternating histogram sets.
    Every pair byte-for-byte against the split kernels (records, flows, predictors, pixel sums), a
    sample that straddles the generation boundaries against the oracle."""
    import torch
    W, H = 640, 480
    p = aof.default_params(W, H, pyramid_levels=2, mean_subtract=1)
    n, distinct = 600, 24
    hp, hc, _ = synth.make_batch(W, H, distinct, 9, 9300, noise=3, brightness=9)
    hp[5, : H // 2] = 9                          # flat upper half: gated blocks at both levels
    hc[5, : H // 2] = 9
    hc[6] = np.minimum(hc[6].astype(np.int32) * 2, 255).astype(np.uint8)   # the clamp of the equalisation bites
    idx = (np.arange(n) * 7) % distinct
    jdx = (idx + (np.arange(n) // 200)) % distinct   # pairs 200.. are mismatched frames: no common motion
    prev = torch.from_numpy(hp[idx]).to(gpu_device)
    cur = torch.from_numpy(hc[jdx]).to(gpu_device)
    outs = []
    for split in (False, True):
        eng = aof.FlowEngine(p, 0)
        eng.set_split_coarse(split)
        blocks, flows, ws = eng.flow_batch(prev, cur)
        torch.cuda.synchronize()
        L = aof.workspace_layout(p, n)
        nb1 = eng.nblocks(1)
        outs.append((blocks.cpu().numpy().tobytes(), flows.cpu().numpy().tobytes(),
                     ws[L.l1_blocks:L.l1_blocks + 4 * nb1 * n].cpu().numpy().tobytes(),
                     ws[L.l1_flows:L.l1_flows + 16 * n].cpu().numpy().tobytes(),
                     ws[L.sums:L.sums + 16 * n].cpu().numpy().tobytes()))
        if not split:
            sample = [0, 5, 6, 199, 200, 255, 256, 257, 511, 512, 513, 599]
            check_against_oracle(aof, orc, p, hp[idx[sample]], hc[jdx[sample]],
                                 dict(blocks=aof.blocks_view(blocks)[sample], flows=aof.flows_view(flows)[sample]))
        del blocks, flows, ws
        eng.close()
    for k, name in enumerate(("records", "flows", "level-1 records", "predictors", "sums")):
        assert outs[0][k] == outs[1][k], f"{name} differ between the fused and the split coarse passes"


@pytest.mark.parametrize("kw", [dict(), dict(subpixel=1), dict(pyramid_levels=2, mean_subtract=1),
                                dict(hist_filter=0, min_valid=3), dict(pyramid_levels=2, subpixel=1)])
@pytest.mark.parametrize("shape", [(640, 480), (200, 150)])
@pytest.mark.parametrize("mode", ["default", "pruned"])
def test_reduction_inside_the_search_launch_equals_k3(aof, orc, synth, gpu_device, shape, kw, mode):
    """(mode "pruned": the column walk of the pruned search, k_flow_lane8_cols, which votes block row by block row.)
    The flat lane8 search reduces in its own launch (votes through agent-scope atomics into the
    context's vote memory, the last wave of a pair writes its flow record): flow records and block
    records byte-for-byte against the separate K3 launch and a sample against the oracle -- on common
    motions (two adds per wave), on mismatched frames (votes all over the histogram), gated and
    saturated pairs; launched repeatedly (the vote memory must be zero again after every launch), on
    alternating streams, and replayed from a captured graph."""
    import torch
    W, H = shape
    p = aof.default_params(W, H, **kw)
    n, distinct = 131, 12
    reach = 9 if kw.get("pyramid_levels") == 2 else 4
    hp, hc, _ = synth.make_batch(W, H, distinct, reach, 9400 + W, noise=3, brightness=7 if kw.get("mean_subtract") else 0)
    hp[3, : H // 2] = 9
    hc[3, : H // 2] = 9
    hp[4] = 0                                    # black previous frame: everything gated, no votes at all
    idx = (np.arange(n) * 5) % distinct
    jdx = (idx + (np.arange(n) // 60)) % distinct   # pairs 60.. compare unrelated frames
    prev = torch.from_numpy(hp[idx]).to(gpu_device)
    cur = torch.from_numpy(hc[jdx]).to(gpu_device)
    eng = aof.FlowEngine(p, 0)
    assert eng.variant == "lane8"
    if mode == "pruned":
        eng.set_search_mode(aof.SEARCH_PRUNED)
    eng.set_reduce_fusion(False)
    b_ref, f_ref, _ = eng.flow_batch(prev, cur)
    torch.cuda.synchronize()
    sample = [0, 3, 4, 59, 60, 61, 130]
    check_against_oracle(aof, orc, p, hp[idx[sample]], hc[jdx[sample]],
                         dict(blocks=aof.blocks_view(b_ref)[sample], flows=aof.flows_view(f_ref)[sample]))
    eng.set_reduce_fusion(True)
    side = torch.cuda.Stream(gpu_device)
    for rep in range(4):
        blocks, flows = torch.zeros_like(b_ref), torch.zeros_like(f_ref)
        if rep & 1:   # another stream than the previous launch: the library orders the two
            with torch.cuda.stream(side):
                eng.flow_batch(prev, cur, blocks=blocks, flows=flows)
        else:
            eng.flow_batch(prev, cur, blocks=blocks, flows=flows)
        torch.cuda.synchronize()
        assert torch.equal(blocks, b_ref), f"launch {rep}: block records differ"
        assert torch.equal(flows, f_ref), f"launch {rep}: flow records differ"
    blocks, flows = torch.zeros_like(b_ref), torch.zeros_like(f_ref)
    ws = torch.zeros(aof.workspace_layout(p, n).total_bytes, dtype=torch.uint8, device=gpu_device)
    eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
    for rep in range(3):
        flows.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(flows, f_ref) and torch.equal(blocks, b_ref), f"graph replay {rep}"
    eng.close()


@pytest.mark.gpu
def test_a_fresh_context_launches_k3_and_fusion_is_opt_in(aof, synth, gpu_device):
    """include/aof.h: the reduction inside the search launch is OFF by default.  A fresh context runs K3 as
    its own kernel (counted by the library's profiling events), aof_set_reduce_fusion(1) removes that launch,
    0 brings it back; and once a graph holding an in-launch reduction has been captured, the library's own
    eager launches on that context keep to K3."""
    import torch
    p = aof.default_params(320, 240)
    hp, hc, _ = synth.make_batch(320, 240, 6, 4, 4100)
    prev, cur = torch.from_numpy(hp).to(gpu_device), torch.from_numpy(hc).to(gpu_device)
    eng = aof.FlowEngine(p, 0)
    assert eng.variant == "lane8"

    def k3_launches(call):
        eng.set_profiling(True)   # (resets the ring)
        call()
        torch.cuda.synchronize()
        n = len(eng.profile_ms(aof.K_REDUCE))
        eng.set_profiling(False)
        return n

    _, f_ref, _ = eng.flow_batch(prev, cur)
    torch.cuda.synchronize()
    assert k3_launches(lambda: eng.flow_batch(prev, cur)) == 1, "default: K3 is launched behind the search"
    eng.set_reduce_fusion(True)
    flows = torch.zeros_like(f_ref)
    assert k3_launches(lambda: eng.flow_batch(prev, cur, flows=flows)) == 0
    assert torch.equal(flows, f_ref)
    eng.set_reduce_fusion(False)
    assert k3_launches(lambda: eng.flow_batch(prev, cur)) == 1
    eng.set_reduce_fusion(True)
    blocks = torch.zeros((6, eng.nblocks(0)), dtype=torch.int32, device=gpu_device)
    ws = torch.zeros(aof.workspace_layout(p, 6).total_bytes, dtype=torch.uint8, device=gpu_device)
    eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
    flows.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(flows, f_ref)
    flows2 = torch.zeros_like(f_ref)
    assert k3_launches(lambda: eng.flow_batch(prev, cur, flows=flows2)) == 1, \
        "a captured graph owns the vote records now: eager launches keep to K3"
    assert torch.equal(flows2, f_ref)
    eng.close()


@pytest.mark.gpu
def test_a_finaliser_deadline_is_an_error_not_a_valid_flow(aof, synth, gpu_device):
    """The finaliser waves of the in-launch reduction give up after a deadline.  Forced here (deadline 0 on a
    launch whose last search waves are still running when the first finalisers start): a pair that was given
    up carries quality 0, count 0 and NO valid flag -- never the all-ones record of round 3 --, every other
    pair equals K3's record, and every later call on the context returns -EIO naming the condition (sticky;
    aof_last_error).  A new context works."""
    import torch
    p = aof.default_params(640, 480)
    n = 256
    hp, hc, _ = synth.make_batch(640, 480, 8, 4, 4200)
    idx = np.arange(n) % 8
    prev, cur = torch.from_numpy(hp[idx]).to(gpu_device), torch.from_numpy(hc[idx]).to(gpu_device)
    eng = aof.FlowEngine(p, 0)
    eng.set_search_mode(aof.SEARCH_EXHAUSTIVE)   # (the kernel that reduces in its launch; an ADAPTIVE context may prune instead)
    _, f_ref, _ = eng.flow_batch(prev, cur)
    torch.cuda.synchronize()
    ref = aof.flows_view(f_ref)
    assert (ref["quality"] > 0).all()
    eng.set_reduce_fusion(True)
    with pytest.raises(aof.AofError):
        eng.set_vote_deadline_us(0)       # a deadline no launch can meet is refused ...
    eng.debug_vote_deadline_ticks(0)      # ... the fault injection is a call of its own
    _, flows, _ = eng.flow_batch(prev, cur)
    torch.cuda.synchronize()
    got = aof.flows_view(flows)
    gave_up = [k for k in range(n) if got[k].tobytes() != ref[k].tobytes()]
    assert gave_up, "deadline 0 on a 256-pair launch: some finaliser must have found its pair incomplete"
    for k in gave_up:
        assert got[k].tobytes() == bytes(16), (k, got[k])   # flow 0, count 0, quality 0, flags 0
    with pytest.raises(aof.AofError) as e:
        eng.flow_batch(prev, cur)
    assert e.value.code == -5 and "deadline" in str(e.value) and "pair" in str(e.value)   # -EIO
    with pytest.raises(aof.AofError):   # sticky
        eng.flow_pair_host(hp[0], hc[0])
    eng.close()
    fresh = aof.FlowEngine(p, 0)
    _, f2, _ = fresh.flow_batch(prev, cur)
    torch.cuda.synchronize()
    assert torch.equal(f2, f_ref)
    fresh.close()


# ---- shapes, options and edge cases ---------------------------------------------

SHAPES = [
    dict(width=128, height=96),                                   # dense, rows a multiple of 16 bytes
    dict(width=136, height=72),                                   # width % 16 != 0 -> lane8
    dict(width=188, height=120, pyramid_levels=2, mean_subtract=1),  # lane8 with predictor + equalisation
    dict(width=188, height=120, subpixel=1),                      # lane8 + refine pass
    dict(width=96, height=81),                                    # odd height, 1 level
    dict(width=160, height=128, pyramid_levels=2),                # level 0 under a predictor
    dict(width=160, height=128, pyramid_levels=2, mean_subtract=1, hist_filter=0),
    dict(width=100, height=90, mean_subtract=1),                  # sums via the scalar K1 path
    dict(width=128, height=96, subpixel=1),                       # dense + half-pixel
    dict(width=160, height=128, subpixel=1, pyramid_levels=2, mean_subtract=1),
    dict(width=160, height=130, subpixel=1, pyramid_levels=2),    # shifted path, odd level-1 height
    dict(width=160, height=130, subpixel=1),                      # H = 8k+18: the ring's last row is the frame's last row
    dict(width=128, height=98, subpixel=1, mean_subtract=1),
    dict(width=128, height=128, grid_mode=1, subpixel=1, num_blocks=7),
    dict(width=128, height=96, grid_mode=1, subpixel=1, pyramid_levels=2, mean_subtract=1),
    dict(width=96, height=96, tile=16, search=8, value_threshold=12000),
    dict(width=96, height=96, tile=16, search=5, value_threshold=12000, subpixel=1),
    dict(width=160, height=128, tile=16, search=8, value_threshold=12000, subpixel=1),   # tile16 + refine
    dict(width=160, height=130, tile=16, search=8, value_threshold=12000, subpixel=1, mean_subtract=1),
    dict(width=320, height=256, tile=16, search=8, value_threshold=12000, pyramid_levels=2),          # tile16 under a predictor
    dict(width=320, height=260, tile=16, search=8, value_threshold=12000, pyramid_levels=2, mean_subtract=1, subpixel=1),
    dict(width=152, height=120),                                  # 252 blocks, rows of 152 bytes: one pair per lane8 workgroup
    dict(width=152, height=120, subpixel=1, mean_subtract=1),
    dict(width=80, height=64, search=2),
    dict(width=80, height=64, search=7, min_valid=0),
    dict(width=1280, height=64),                                  # wide rows: strip planning
]


@pytest.mark.parametrize("case", range(len(SHAPES)))
def test_shapes_and_options(aof, orc, synth, gpu_device, case):
    import torch
    kw = SHAPES[case]
    p = aof.default_params(**kw)
    reach = 2 * p.search + 1 if p.pyramid_levels == 2 else p.search
    n = 6
    prevs, curs, _ = synth.make_batch(p.width, p.height, n, reach, 2000 + 10 * case, noise=5,
                                      brightness=7 if p.mean_subtract else 0)
    rng = np.random.default_rng(case)
    curs[n - 1] = rng.integers(0, 256, curs[n - 1].shape, dtype=np.uint8)  # unrelated frame
    prevs[n - 2][: p.height // 2] = 90                                       # half the frame flat
    eng = aof.FlowEngine(p, 0)
    tp, tc = torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device)
    nb = eng.nblocks(0)
    sub = torch.full((n, nb), 99, dtype=torch.uint8, device=gpu_device) if p.subpixel else None
    blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
    torch.cuda.synchronize()
    got = dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows))
    check_against_oracle(aof, orc, p, prevs, curs, got,
                         subdirs=sub.cpu().numpy() if sub is not None else None)


@pytest.mark.parametrize("mode", ["exhaustive", "fused_reduce", "pruned", "generic"])
@pytest.mark.parametrize("kw", [dict(), dict(pyramid_levels=2, mean_subtract=1)])
def test_vga_dense_half_pixel(aof, orc, synth, gpu_device, mode, kw):
    """Dense grid with half-pixel refinement (origin S+1): the integer search on the moved origin
    plus the refinement in the same lane must equal the oracle, every direction exercised."""
    import torch
    p = aof.default_params(640, 480, subpixel=1, **kw)
    n = 10
    prevs, curs, _ = synth.make_batch(640, 480, n, 4, 4100, noise=3, brightness=5 if kw else 0)
    for i, half in enumerate([(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1)]):
        prevs[i], curs[i], _ = synth.make_pair(640, 480, 4, 50 + i, shift=(2 - i % 5, i % 3 - 1), half=half)
    eng = aof.FlowEngine(p, 0)
    assert eng.variant == "lane8"
    if mode == "generic":
        eng.force_generic(True)
    elif mode == "pruned":
        eng.set_search_mode(aof.SEARCH_PRUNED)
        assert eng.variant == "lane8"
    elif mode == "fused_reduce":
        eng.set_reduce_fusion(True)
    tp, tc = torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device)
    sub = torch.full((n, eng.nblocks(0)), 99, dtype=torch.uint8, device=gpu_device)
    blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
    torch.cuda.synchronize()
    got = dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows))
    check_against_oracle(aof, orc, p, prevs, curs, got, subdirs=sub.cpu().numpy())
    seen = set(np.unique(sub.cpu().numpy()[:8]))
    assert seen >= {0, 1, 2, 3, 4, 5, 6, 7}, seen


SMALL = [
    (dict(px4=1), 64, 64),                                      # configs[0]
    (dict(px4=1, mean_subtract=1, hist_filter=0), 64, 64),
    (dict(px4=1, num_blocks=8), 96, 80),
    (dict(subpixel=1), 128, 128),                               # the reference application's image size, dense
    (dict(subpixel=1, mean_subtract=1), 128, 96),
    (dict(), 136, 72),                                          # rows of 136 bytes
    (dict(mean_subtract=1, feature_threshold=0, value_threshold=70000), 100, 64),
]


@pytest.mark.parametrize("case", range(len(SMALL)))
def test_small_frames(aof, orc, synth, gpu_device, case):
    """Small frames (the reference's own sizes) on sparse and dense grids: lane-per-block search,
    refinement pass and wave-per-pair reduction against the oracle; every half-pixel direction."""
    import torch
    kw, w, h = SMALL[case]
    kw = dict(kw)
    p = aof.px4flow_params(w, h, **kw) if kw.pop("px4", 0) else aof.default_params(w, h, **kw)
    n = 23
    prevs, curs, _ = synth.make_batch(w, h, n, 4, 6100 + case, noise=3, brightness=9 if p.mean_subtract else 0)
    for i, half in enumerate([(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1)]):
        prevs[i], curs[i], _ = synth.make_pair(w, h, 4, 70 + i, shift=(2 - i % 5, i % 3 - 1), half=half)
    rng = np.random.default_rng(case)
    curs[n - 1] = rng.integers(0, 256, curs[n - 1].shape, dtype=np.uint8)
    prevs[n - 2][: h // 2] = 90
    eng = aof.FlowEngine(p, 0)
    tp, tc = torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device)
    sub = torch.full((n, eng.nblocks(0)), 99, dtype=torch.uint8, device=gpu_device) if p.subpixel else None
    blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
    torch.cuda.synchronize()
    got = dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows))
    check_against_oracle(aof, orc, p, prevs, curs, got, subdirs=sub.cpu().numpy() if sub is not None else None)
    if p.subpixel:
        assert set(np.unique(sub.cpu().numpy()[:8])) >= {0, 1, 2, 3, 4, 5, 6, 7}


CANARY = [
    dict(width=640, height=480),                                           # flat lane8, K3 / in-launch reduction
    dict(width=640, height=480, subpixel=1),                               # flat lane8 + refinement in the lane
    dict(width=160, height=130, pyramid_levels=2, mean_subtract=1, subpixel=1),
    dict(width=320, height=240, tile=16, search=8, value_threshold=12000, subpixel=1),  # tile16 + K2b
    dict(px4=1, width=64, height=64),                                      # grouped lane8
    dict(width=188, height=120, subpixel=1),                               # flat lane8
    dict(width=100, height=90, search=3, subpixel=1),                      # generic
]


@pytest.mark.parametrize("case", range(len(CANARY)))
def test_no_writes_outside_the_output_buffers(aof, synth, gpu_device, case):
    """Every output (records, directions, flows, workspace) sits between two guard zones inside
    one larger allocation; the guards must still hold their pattern after the launch."""
    import torch
    kw = dict(CANARY[case])
    w, h = kw.pop("width"), kw.pop("height")
    p = aof.px4flow_params(w, h, **kw) if kw.pop("px4", 0) else aof.default_params(w, h, **kw)
    n, guard = 37, 4096
    reach = 2 * p.search + 1 if p.pyramid_levels == 2 else p.search
    m = 6
    prevs, curs, _ = synth.make_batch(w, h, m, reach, 9100 + case, noise=4, brightness=5 if p.mean_subtract else 0)
    reps = (n + m - 1) // m
    tp = torch.from_numpy(prevs).to(gpu_device).repeat(reps, 1, 1)[:n].contiguous()
    tc = torch.from_numpy(curs).to(gpu_device).repeat(reps, 1, 1)[:n].contiguous()
    eng = aof.FlowEngine(p, 0)
    nb = eng.nblocks(0)
    sizes = dict(blocks=4 * n * nb, subdirs=n * nb, flows=16 * n, ws=aof.workspace_layout(p, n).total_bytes)
    total = sum(guard + (v + 255) // 256 * 256 for v in sizes.values()) + guard
    arena = torch.full((total,), 0xA5, dtype=torch.uint8, device=gpu_device)
    off, views, gaps = guard, {}, [(0, guard)]
    for k, v in sizes.items():
        views[k] = arena[off:off + v]
        end = off + (v + 255) // 256 * 256
        gaps.append((off + v, end + guard))
        off = end + guard
    assert views["ws"].data_ptr() % 256 == 0 and views["blocks"].data_ptr() % 4 == 0
    for mode in ("default", "fused_reduce", "pruned"):
        eng.set_reduce_fusion(mode == "fused_reduce")
        eng.set_search_mode(aof.SEARCH_PRUNED if mode == "pruned" else aof.SEARCH_EXHAUSTIVE)
        eng.flow_batch(tp, tc, blocks=views["blocks"].view(torch.int32).view(n, nb),
                       subdirs=views["subdirs"].view(n, nb) if p.subpixel else None,
                       flows=views["flows"].view(n, 16), workspace=views["ws"])
        torch.cuda.synchronize()
        host = arena.cpu().numpy()
        for lo, hi in gaps:
            assert (host[lo:hi] == 0xA5).all(), (mode, lo, hi, np.nonzero(host[lo:hi] != 0xA5)[0][:8])


def test_flat_and_saturated_frames(aof, orc, gpu_device):
    p = aof.default_params(128, 96)
    prevs = np.stack([np.full((96, 128), v, np.uint8) for v in (0, 77, 255)])
    got = run_gpu(aof, p, prevs, prevs.copy(), gpu_device)
    assert (got["blocks"]["sad"] == 0xFFFF).all()
    assert (got["flows"]["count"] == 0).all() and (got["flows"]["flags"] == 0).all()
    check_against_oracle(aof, orc, p, prevs, prevs, got)


def test_first_minimum_wins_on_periodic_texture(aof, orc, gpu_device):
    p = aof.default_params(128, 96)
    img = np.zeros((96, 128), np.uint8)
    img[:, 0::2] = 200            # period 2 in x: five zero-SAD candidates per row of the scan
    img2 = np.zeros((96, 128), np.uint8)
    img2[0::4, :] = 150           # period 4 in y
    prevs = np.stack([img, img2])
    for force in (False, True):
        got = run_gpu(aof, p, prevs, prevs.copy(), gpu_device, force_generic=force)
        assert (got["blocks"][0]["dx"] == -4).all() and (got["blocks"][0]["dy"] == -4).all()
        assert (got["blocks"][1]["dy"] == -4).all() and (got["blocks"][1]["dx"] == -4).all()
        check_against_oracle(aof, orc, p, prevs, prevs, got)


def test_extreme_sads_and_thresholds(aof, orc, gpu_device):
    """Max-contrast checkerboards drive the packed u16 accumulators to 64*255."""
    yy, xx = np.mgrid[0:96, 0:128]
    a = (((xx // 8 + yy // 8) % 2) * 255).astype(np.uint8)
    b = 255 - a
    for kw in (dict(), dict(value_threshold=70000), dict(feature_threshold=0), dict(value_threshold=0)):
        p = aof.default_params(128, 96, **kw)
        prevs, curs = np.stack([a, b, a]), np.stack([b, a, a])
        for force in (False, True):
            got = run_gpu(aof, p, prevs, curs, gpu_device, force_generic=force)
            check_against_oracle(aof, orc, p, prevs, curs, got)
    p = aof.default_params(128, 96, tile=16, search=8, value_threshold=70000, feature_threshold=0)
    got = run_gpu(aof, p, np.stack([a]), np.stack([b]), gpu_device)
    check_against_oracle(aof, orc, p, np.stack([a]), np.stack([b]), got)


def test_mean_equalisation_saturates_like_the_oracle(aof, orc, synth, gpu_device):
    p = aof.default_params(128, 96, mean_subtract=1)
    prev, cur, _ = synth.make_pair(128, 96, 4, 77, contrast=2.5)
    dark = np.clip(cur.astype(np.int32) - 90, 0, 255).astype(np.uint8)   # delta > 0, clips at 255
    bright = np.clip(cur.astype(np.int32) + 90, 0, 255).astype(np.uint8)  # delta < 0, clips at 0
    prevs, curs = np.stack([prev, prev, prev]), np.stack([dark, bright, cur])
    for force in (False, True):
        got = run_gpu(aof, p, prevs, curs, gpu_device, force_generic=force)
        check_against_oracle(aof, orc, p, prevs, curs, got)


def test_predictor_pushes_windows_out_of_frame(aof, orc, synth, gpu_device):
    """2 levels, large shift: edge blocks' displaced windows leave the frame and are skipped."""
    p = aof.default_params(192, 160, pyramid_levels=2)
    for shift in ((9, 8), (-9, -8), (8, -9), (-7, 9), (6, 0)):
        prev, cur, _ = synth.make_pair(192, 160, 12, 5, shift=shift)
        for force in (False, True):
            got = run_gpu(aof, p, prev[None], cur[None], gpu_device, force_generic=force)
            check_against_oracle(aof, orc, p, prev[None], cur[None], got)
            sk = got["blocks"][0]["sad"] == 0xFFFF
            assert sk.any() and not sk.all()


def test_frame_sequence_via_pair_stride(aof, orc, synth, gpu_device):
    """A resident frame SEQUENCE: pair k = (frame k, frame k+1), no copies."""
    import torch
    p = aof.default_params(128, 96)
    frames, steps = synth.make_sequence(128, 96, 9, 4, seed=4)
    t = torch.from_numpy(frames).to(gpu_device)
    eng = aof.FlowEngine(p, 0)
    blocks, flows, _ = eng.flow_batch(t[:-1], t[1:], n_pairs=8, pair_stride=128 * 96)
    torch.cuda.synchronize()
    got = dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows))
    check_against_oracle(aof, orc, p, frames[:-1], frames[1:], got)
    assert np.array_equal(got["flows"]["flow_x"], steps[:, 0].astype(np.float32))


@pytest.mark.parametrize("kw,size,n", [
    (dict(pyramid_levels=2, mean_subtract=1, _px4=1), (128, 128), 200),    # the sequence pipeline's own configuration
    (dict(pyramid_levels=2, mean_subtract=1), (160, 128), 40),             # dense grid: fused coarse kernel / split K1
    (dict(mean_subtract=1), (136, 72), 30),                                # one level, sums only, scalar K1 (width % 16 != 0)
    (dict(pyramid_levels=2, _px4=1), (96, 80), 2),                         # a single pair of a sequence view
    (dict(tile=16, search=8, pyramid_levels=2, mean_subtract=1, value_threshold=12000), (320, 256), 12),
])
def test_frame_sequence_sums_and_filters_every_frame_once(aof, orc, synth, gpu_device, kw, size, n):
    """In the sequence view (d_cur = d_prev + one frame, pair_stride = one frame) K1 runs once per FRAME: frame
    k's sums land in pair k (prev) and pair k-1 (cur), its level-1 image once in the workspace, and the level-1
    search views that sequence twice.  Records, flows, pixel sums and level-1 frames against the oracle and
    against the same pairs run as independent copies."""
    import torch
    kw = dict(kw)
    W, H = size
    px4 = kw.pop("_px4", 0)
    p = aof.px4flow_params(W, H, **kw) if px4 else aof.default_params(W, H, **kw)
    frames, _ = synth.make_sequence(W, H, n + 1, 6, seed=41, max_step=5)
    frames = frames.copy()
    frames[1::3] = np.clip(frames[1::3].astype(np.int16) + 11, 0, 255).astype(np.uint8)   # exposure steps: deltas at work
    t = torch.from_numpy(frames).to(gpu_device)
    L = aof.workspace_layout(p, n)
    for split in (False, True):
        eng = aof.FlowEngine(p, 0)
        if split:
            eng.set_split_coarse(True)
        ws = torch.full((L.total_bytes,), 0xA5, dtype=torch.uint8, device=gpu_device)
        blocks, flows, _ = eng.flow_batch(t[:-1], t[1:], n_pairs=n, pair_stride=W * H, workspace=ws)
        # the same pairs as independent copies (no sequence view: cur is another buffer)
        prev_c, cur_c = t[:-1].clone(), t[1:].clone()
        b2, f2, ws2 = eng.flow_batch(prev_c, cur_c)
        torch.cuda.synchronize()
        assert torch.equal(blocks, b2) and torch.equal(flows, f2), split
        got = dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows))
        sample = sorted(set([0, 1, n // 2, n - 1]))
        check_against_oracle(aof, orc, p, frames[:-1][sample], frames[1:][sample],
                             dict(blocks=got["blocks"][sample], flows=got["flows"][sample]))
        if p.mean_subtract and (split or px4 or p.tile == 16 or p.pyramid_levels == 1):   # (K1 ran: its sums are in the workspace)
            sums = ws[L.sums:L.sums + 16 * n].cpu().numpy().view(np.uint32).reshape(n, 2, 2)
            sums2 = ws2[L.sums:L.sums + 16 * n].cpu().numpy().view(np.uint32).reshape(n, 2, 2)
            assert np.array_equal(sums, sums2)
            assert np.array_equal(sums[:, 0, 0], frames[:-1].reshape(n, -1).sum(axis=1, dtype=np.uint64).astype(np.uint32))
            assert np.array_equal(sums[:, 1, 0], frames[1:].reshape(n, -1).sum(axis=1, dtype=np.uint64).astype(np.uint32))
        eng.close()


def test_padded_pair_stride_and_caller_buffers(aof, orc, synth, gpu_device):
    import torch
    p = aof.default_params(128, 96)
    prevs, curs, _ = synth.make_batch(128, 96, 3, 4, 900, noise=2)
    stride = 128 * 96 + 256
    bp = torch.zeros(3 * stride, dtype=torch.uint8, device=gpu_device)
    bc = torch.zeros(3 * stride, dtype=torch.uint8, device=gpu_device)
    for i in range(3):
        bp[i * stride:i * stride + 128 * 96] = torch.from_numpy(prevs[i].ravel()).to(gpu_device)
        bc[i * stride:i * stride + 128 * 96] = torch.from_numpy(curs[i].ravel()).to(gpu_device)
    eng = aof.FlowEngine(p, 0)
    blocks, flows, _ = eng.flow_batch(bp, bc, n_pairs=3, pair_stride=stride)
    torch.cuda.synchronize()
    check_against_oracle(aof, orc, p, prevs, curs,
                         dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows)))


@pytest.mark.parametrize("kw", [dict(), dict(subpixel=1), dict(px4=1), dict(pyramid_levels=2, mean_subtract=1)])
def test_unaligned_pair_stride_and_base(aof, orc, synth, gpu_device, kw):
    """Frames at odd byte offsets and an odd pair stride: the strip kernels' alignment checks
    must hand the batch to a kernel that reads byte-aligned sources."""
    import torch
    kw = dict(kw)
    p = aof.px4flow_params(128, 96, **kw) if kw.pop("px4", 0) else aof.default_params(128, 96, **kw)
    reach = 9 if p.pyramid_levels == 2 else 4
    prevs, curs, _ = synth.make_batch(128, 96, 5, reach, 950, noise=3, brightness=4 if p.mean_subtract else 0)
    stride, lead = 128 * 96 + 7, 3
    bp = torch.zeros(lead + 5 * stride, dtype=torch.uint8, device=gpu_device)
    bc = torch.zeros(lead + 5 * stride, dtype=torch.uint8, device=gpu_device)
    for i in range(5):
        bp[lead + i * stride:lead + i * stride + 128 * 96] = torch.from_numpy(prevs[i].ravel()).to(gpu_device)
        bc[lead + i * stride:lead + i * stride + 128 * 96] = torch.from_numpy(curs[i].ravel()).to(gpu_device)
    eng = aof.FlowEngine(p, 0)
    sub = torch.full((5, eng.nblocks(0)), 99, dtype=torch.uint8, device=gpu_device) if p.subpixel else None
    blocks, flows, _ = eng.flow_batch(bp[lead:], bc[lead:], n_pairs=5, pair_stride=stride, subdirs=sub)
    torch.cuda.synchronize()
    check_against_oracle(aof, orc, p, prevs, curs, dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows)),
                         subdirs=sub.cpu().numpy() if sub is not None else None)


def test_very_large_batch_of_small_frames(aof, orc, synth, gpu_device):
    """100 000 pairs of configs[0] in one launch: grouped lane8 workgroups (ragged last one) and
    the generic kernel's y/z grid split agree on every record; a sample is checked against the
    oracle."""
    import torch
    p = aof.px4flow_params(64, 64)
    n, m = 100_003, 40
    prevs, curs, _ = synth.make_batch(64, 64, m, 4, 990, noise=3)
    reps = (n + m - 1) // m
    tp = torch.from_numpy(prevs).to(gpu_device).repeat(reps, 1, 1)[:n].contiguous()
    tc = torch.from_numpy(curs).to(gpu_device).repeat(reps, 1, 1)[:n].contiguous()
    outs = []
    for generic in (False, True):
        eng = aof.FlowEngine(p, 0)
        eng.force_generic(generic)
        sub = torch.full((n, 25), 99, dtype=torch.uint8, device=gpu_device)
        blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
        torch.cuda.synchronize()
        outs.append((blocks.cpu().numpy(), flows.cpu().numpy(), sub.cpu().numpy()))
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        assert a.tobytes() == b.tobytes()
    got = dict(blocks=aof.blocks_view(torch.from_numpy(outs[0][0][-m:])), flows=aof.flows_view(torch.from_numpy(outs[0][1][-m:])))
    idx = [(n - m + i) % m for i in range(m)]
    check_against_oracle(aof, orc, p, prevs[idx], curs[idx], got, subdirs=outs[0][2][-m:])


@pytest.mark.parametrize("kw", [dict(), dict(subpixel=1, hist_filter=0), dict(pyramid_levels=2, mean_subtract=1)])
def test_large_grid_two_step_reduction(aof, orc, synth, gpu_device, kw):
    """1024x768 has 11 844 blocks per pair: the reduction runs as per-chunk vote histograms plus
    the per-pair sum; records and flow records must still equal the oracle's."""
    import torch
    p = aof.default_params(1024, 768, **kw)
    reach = 9 if p.pyramid_levels == 2 else 4
    prevs, curs, _ = synth.make_batch(1024, 768, 3, reach, 1234, noise=5, brightness=4 if p.mean_subtract else 0)
    prevs[2][:300] = 77                                   # a flat band: many skipped blocks
    eng = aof.FlowEngine(p, 0)
    assert eng.nblocks(0) > 8192
    tp, tc = torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device)
    sub = torch.full((3, eng.nblocks(0)), 99, dtype=torch.uint8, device=gpu_device) if p.subpixel else None
    blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
    torch.cuda.synchronize()
    check_against_oracle(aof, orc, p, prevs, curs, dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows)),
                         subdirs=sub.cpu().numpy() if sub is not None else None)


def test_empty_batch_and_errors(aof, gpu_device):
    import torch
    p = aof.default_params(128, 96)
    eng = aof.FlowEngine(p, 0)
    z = torch.empty((0, 96, 128), dtype=torch.uint8, device=gpu_device)
    blocks, flows, _ = eng.flow_batch(z, z, n_pairs=0, pair_stride=128 * 96)
    assert blocks.shape[0] == 0
    t = torch.zeros((2, 96, 128), dtype=torch.uint8, device=gpu_device)
    small = torch.zeros(256, dtype=torch.uint8, device=gpu_device)
    with pytest.raises(aof.AofError) as e:
        eng.flow_batch(t, t, workspace=small)
    assert e.value.code == -28  # -ENOSPC
    with pytest.raises(aof.AofError):
        aof.FlowEngine(p, 99)  # no such device


def test_host_pair_and_streaming_entry_points(aof, orc, synth, gpu_device):
    p = aof.px4flow_params(64, 64)
    eng = aof.FlowEngine(p, 0)
    po = orc.params_from(p)
    prev, cur, _ = synth.make_pair(64, 64, 4, 31, noise=3)
    blocks, subdirs, flow = eng.flow_pair_host(prev, cur)
    ref = orc.flow_pair(po, prev, cur)
    assert blocks.tobytes() == ref["blocks"].tobytes() and np.array_equal(subdirs, ref["subdirs"])
    assert flow.tobytes() == ref["flow"].tobytes()
    frames, _ = synth.make_sequence(64, 64, 6, 4, seed=9, max_step=3)
    assert eng.stream_push(frames[0]) is None
    for k in range(1, 6):
        f = eng.stream_push(frames[k])
        assert f.tobytes() == orc.flow_pair(po, frames[k - 1], frames[k])["flow"].tobytes()
    eng.stream_reset()
    assert eng.stream_push(frames[3]) is None


def facade_oracle_params(aof, orc, cls, w, h):
    """Oracle parameters of what the facade class ships: OpticalFlowPX4 = the published
    single-level PX4Flow configuration; OpticalFlowOpenCV = the same grid on two levels with
    mean equalisation wherever the geometry carries a half-resolution grid."""
    if cls == "OpticalFlowOpenCV":
        two = aof.px4flow_params(w, h, pyramid_levels=2, mean_subtract=1)
        if aof.check_params(two) == 0:
            return orc.px4flow_params(w, h, pyramid_levels=2, mean_subtract=1), 2
    return orc.px4flow_params(w, h), 1


def same_outputs(got, exp, k):
    assert got[0] == exp[0], (k, got, exp)
    if exp[0] >= 0:
        assert got[1] == exp[1], (k, got, exp)
        assert np.float32(got[2]).tobytes() == np.float32(exp[2]).tobytes(), (k, got, exp)
        assert np.float32(got[3]).tobytes() == np.float32(exp[3]).tobytes(), (k, got, exp)


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("cls", ["OpticalFlowPX4", "OpticalFlowOpenCV"])
def test_facade_calcflow_matches_oracle(aof, orc, synth, gpu_device, cls, resident):
    """The C++ facade classes (calcFlow contract of mainloop.cpp:322-331) through
    the HIP engine, against the oracle of the same semantics -- bit-exact floats; one launch per
    call (replayed hipGraph) and served by the resident kernel (setResidentKernel)."""
    fx, fy = 216.6677, 216.2457  # /root/reference/src/main.cpp:60-61
    for rate, size in ((15, (64, 64)), (0, (64, 64)), (40, (128, 128)), (15, (128, 96)), (15, (75, 64))):
        flow = getattr(aof, cls)(fx, fy, rate, size[0], size[1])
        assert flow.lastError() == "ok"
        assert flow.setResidentKernel(resident)
        assert (flow.getImageWidth(), flow.getImageHeight()) == size
        po, levels = facade_oracle_params(aof, orc, cls, *size)
        assert flow.getPyramidLevels() == levels
        if cls == "OpticalFlowOpenCV":
            assert levels == (1 if size[0] % 2 else 2)
        o = orc.Px4(po, fx, fy, rate)
        frames, _ = synth.make_sequence(size[0], size[1], 30, 4, seed=rate, max_step=2)
        t, published = 0, 0
        for k in range(30):
            got = flow.calcFlow(frames[k], t)
            exp = o.calc_flow(frames[k], t)
            same_outputs(got, exp, (rate, k))
            if exp[0] >= 0 and k > 0:
                published += 1
            t += 13333
        assert published >= {15: 4, 0: 29, 40: 14}[rate]
        flow.close()


@pytest.mark.parametrize("cls", ["OpticalFlowPX4", "OpticalFlowOpenCV"])
def test_facade_timestamp_wrap_and_frames_without_quality(aof, orc, synth, gpu_device, cls):
    """Edge paths of the calcFlow contract through the real classes: timestamps that cross
    2^32 us inside an integration window (mainloop.cpp:305-315 hands over a u32 that wraps),
    and frames of quality 0 inside a window (flat images: nothing passes the gradient gate),
    which must neither add flow nor count towards the mean quality."""
    fx, fy = 216.6677, 216.2457
    W = H = 128
    frames, _ = synth.make_sequence(W, H, 40, 4, seed=77, max_step=2)
    frames = frames.copy()
    for k in (7, 8, 19, 31):
        frames[k] = 97                       # flat: quality 0 for the pairs whose PREVIOUS frame it is
    for rate in (15, 0):
        flow = getattr(aof, cls)(fx, fy, rate, W, H)
        po, _ = facade_oracle_params(aof, orc, cls, W, H)
        o = orc.Px4(po, fx, fy, rate)
        # time starts at 0 with the first frame (mainloop.cpp:305-315) and wraps after 71 minutes:
        # three 20-minute gaps, a last one that ends 30 ms short of 2^32 us, then 75 Hz frames, so
        # that the wrap falls between two frames of one integration window
        times = [0, 1200000011, 2400000022, 3600000033, (1 << 32) - 30000]
        while len(times) < 40:
            times.append(times[-1] + 13333)
        saw_wrap, zero_quality_published, published = False, 0, 0
        for k in range(40):
            got = flow.calcFlow(frames[k], times[k])
            exp = o.calc_flow(frames[k], times[k] & 0xFFFFFFFF)
            same_outputs(got, exp, (rate, k))
            if got[0] >= 0 and k > 0:
                published += 1
                assert 0 < got[1] < (1 << 31), "dt_us stays a positive interval across the wrap"
                if k > 4:
                    assert got[1] < 200000
                zero_quality_published += got[0] == 0
            saw_wrap |= k > 0 and (times[k] >> 32) != (times[k - 1] >> 32)
        assert saw_wrap and published >= (6 if rate else 39)
        if rate == 0:
            assert zero_quality_published >= 4   # every pair whose PREVIOUS frame is flat: nothing passes the gate
        flow.close()


def test_opencv_facade_reaches_beyond_the_single_level_search(aof, orc, synth, gpu_device):
    """The class mainloop.cpp:423 creates runs two levels: per-frame shifts of 6..9 px (beyond
    the +-4.5 px of one level; the half-resolution match must lie within its own +-4 px) come out exactly; OpticalFlowPX4 (one level, the published
    configuration) cannot follow them, and setSearchPyramid() switches either class over."""
    fx, fy = 216.6677, 216.2457
    W = H = 128
    for shift in ((9, -7), (-8, 6), (6, 9), (-9, -9)):
        steps = np.tile(np.array(shift), (5, 1))
        pos = np.vstack([[0, 0], np.cumsum(steps, axis=0)])
        span = int(np.abs(pos).max()) + 1
        c = synth.canvas(W, H, span, 991 + shift[0])
        frames = np.stack([c[span - py:span - py + H, span - px:span - px + W] for px, py in pos])
        cv = aof.OpticalFlowOpenCV(fx, fy, 0, W, H)        # rate 0: every frame publishes
        px4 = aof.OpticalFlowPX4(fx, fy, 0, W, H)
        o = orc.Px4(orc.px4flow_params(W, H, pyramid_levels=2, mean_subtract=1), fx, fy, 0)
        assert cv.getPyramidLevels() == 2 and px4.getPyramidLevels() == 1
        for k in range(6):
            got, exp = cv.calcFlow(frames[k], 13333 * k), o.calc_flow(frames[k], 13333 * k)
            same_outputs(got, exp, (shift, k))
            one = px4.calcFlow(frames[k], 13333 * k)
            if k > 0:
                assert got[0] >= 150      # (tiles whose displaced window leaves the frame drop out)
                assert np.float32(got[2]) == np.arctan2(np.float32(shift[0]), np.float32(fx)).astype(np.float32)
                assert np.float32(got[3]) == np.arctan2(np.float32(shift[1]), np.float32(fy)).astype(np.float32)
                assert abs(np.tan(one[2]) * fx) <= 4.5 + 1e-3, "one level saturates at the search limit"
        # switching the single-level class to two levels makes it follow as well (fresh state)
        assert px4.setSearchPyramid(2, True) and px4.getPyramidLevels() == 2
        o2 = orc.Px4(orc.px4flow_params(W, H, pyramid_levels=2, mean_subtract=1), fx, fy, 0)
        for k in range(3):   # (time goes on for the live object; the fresh oracle only pins quality and flow)
            got, exp = px4.calcFlow(frames[k], 13333 * (6 + k)), o2.calc_flow(frames[k], 13333 * k)
            assert got[0] == exp[0], (shift, "switched", k, got, exp)
            assert np.float32(got[2]).tobytes() == np.float32(exp[2]).tobytes()
            assert np.float32(got[3]).tobytes() == np.float32(exp[3]).tobytes()
        assert not aof.OpticalFlowPX4(fx, fy, 0, 75, 64).setSearchPyramid(2, True)   # odd width: refused, engine kept
        cv.close(); px4.close()


def test_replay_harness_links_and_matches_oracle(aof, orc, synth, gpu_device, tmp_path):
    """replay_mainloop.cpp replays mainloop.cpp:295-331 (crop, contiguous copy, calcFlow,
    negative gate) against <flow_opencv.hpp> with the reference's argument types."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(aof.FACADE_PATH), "replay", "replay_mainloop")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    cam_w, cam_h, crop = 320, 240, 128  # the reference's defaults, main.cpp:54-58
    frames, _ = synth.make_sequence(cam_w, cam_h, 24, 4, seed=5, max_step=3)
    raw = tmp_path / "frames.raw"
    raw.write_bytes(frames.tobytes())
    out = subprocess.run([exe, str(raw), str(cam_w), str(cam_h), str(crop), str(crop), "75"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if not l.startswith("#")]
    assert "engine: ok" in out.stdout and "DEFAULT_OUTPUT_RATE=15" in out.stdout
    o = orc.Px4(facade_oracle_params(aof, orc, "OpticalFlowOpenCV", crop, crop)[0], 216.6677, 216.2457, 15)
    x0, y0 = cam_w // 2 - crop // 2, cam_h // 2 - crop // 2
    n_pub = 0
    for k in range(24):
        t = int(1000000 + int(k * 1.0e6 / 75)) - 1000000 if k else 0
        q, dt, ax, ay = o.calc_flow(frames[k, y0:y0 + crop, x0:x0 + crop], t)
        if q < 0:
            assert lines[k] == f"{k} skip"
        else:
            n_pub += 1
            head, wire = lines[k].split(" mavlink=")
            assert head == (f"{k} quality={q} integration_time_us={dt} "
                            f"integrated_x={ax:.9g} integrated_y={ay:.9g}"), (lines[k], q, dt, ax, ay)
            exp = aof.pack_optical_flow_rad(5000000, t, dt, ax, ay, quality=q, seq=n_pub & 0xFF)
            assert wire == exp.hex(), "OPTICAL_FLOW_RAD frame of the published flow"
    assert n_pub >= 4


def test_facade_track_features_matches_oracle(aof, orc, synth, gpu_device):
    """OpticalFlowPX4::trackFeatures: the sparse grid's tiles as tracked features."""
    flow = aof.OpticalFlowPX4(216.0, 216.0, 15, 64, 64)
    po = orc.px4flow_params(64, 64)
    g = orc.grid(po, 0)
    hx = [1, 1, 0, -1, -1, -1, 0, 1, 0]
    hy = [0, 1, 1, 1, 0, -1, -1, -1, 0]
    for idx, kw in enumerate([dict(shift=(2, -1)), dict(shift=(0, 3), half=(1, 0)), dict(noise=9),
                              dict(shift=(-3, -3), half=(-1, 1))]):
        prev, cur, _ = synth.make_pair(64, 64, 4, 60 + idx, **kw)
        if idx == 2:
            prev[:32] = 80  # flat upper half: gradient gate rejects those tiles
        feats = flow.trackFeatures(prev, cur)
        ref = orc.flow_pair(po, prev, cur)
        assert feats.shape == (25, 6)
        for k in range(25):
            b, sd = ref["blocks"][k], int(ref["subdirs"][k])
            px, py = g.x0 + (k % g.nx) * g.step_x + 4.0, g.y0 + (k // g.nx) * g.step_y + 4.0
            searched = b["sad"] != 0xFFFF
            exp = [px, py, px + (b["dx"] + 0.5 * hx[sd] if searched else 0), py + (b["dy"] + 0.5 * hy[sd] if searched else 0),
                   float(b["sad"]) if searched else -1.0, 1.0 if (searched and b["sad"] < 3000) else 0.0]
            assert list(feats[k]) == [np.float32(v) for v in exp], (idx, k, feats[k], exp)
    # the streaming state of calcFlow is untouched by trackFeatures
    frames, _ = synth.make_sequence(64, 64, 3, 4, seed=3, max_step=2)
    o = orc.Px4(po, 216.0, 216.0, 0)
    o.calc_flow(frames[0], 0)
    flow2 = aof.OpticalFlowPX4(216.0, 216.0, 0, 64, 64)
    flow2.calcFlow(frames[0], 0)
    flow2.trackFeatures(frames[2], frames[0])
    assert flow2.calcFlow(frames[1], 13333) == o.calc_flow(frames[1], 13333)


@pytest.mark.parametrize("kw", [dict(), dict(pyramid_levels=2, mean_subtract=1)])
def test_batch_path_is_graph_capturable(aof, orc, synth, gpu_device, kw):
    """aof_flow_batch_device only enqueues (no allocation, no sync): the whole launch
    sequence can be captured into a hipGraph and replayed on new frame contents."""
    import torch
    p = aof.default_params(128, 96, **kw)
    eng = aof.FlowEngine(p, 0)
    n = 6
    prev = torch.zeros((n, 96, 128), dtype=torch.uint8, device=gpu_device)
    cur = torch.zeros_like(prev)
    blocks = torch.zeros((n, eng.nblocks(0)), dtype=torch.int32, device=gpu_device)
    flows = torch.zeros((n, 16), dtype=torch.uint8, device=gpu_device)
    ws = torch.zeros(aof.workspace_layout(p, n).total_bytes, dtype=torch.uint8, device=gpu_device)
    eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)  # warm-up outside capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.flow_batch(prev, cur, blocks=blocks, flows=flows, workspace=ws)
    for rep in range(8):  # (the per-pair predictor and sums go through the scalar cache: stale lines would show here)
        hp, hc, _ = synth.make_batch(128, 96, n, 9 if kw else 4, 3000 + 10 * rep, noise=4,
                                     brightness=6 if kw else 0)
        prev.copy_(torch.from_numpy(hp))
        cur.copy_(torch.from_numpy(hc))
        g.replay()
        torch.cuda.synchronize()
        check_against_oracle(aof, orc, p, hp, hc, dict(blocks=aof.blocks_view(blocks), flows=aof.flows_view(flows)))


@pytest.mark.parametrize("kw", [dict(), dict(pyramid_levels=2, mean_subtract=1, grid_mode=0, subpixel=0)])
def test_streaming_graph_and_eager_paths_agree(aof, orc, synth, gpu_device, kw):
    """aof_stream_push_host replays a captured hipGraph per frame; switching the capture
    off must give the same records, and both must match the oracle."""
    p = aof.px4flow_params(64, 64, **kw) if not kw else aof.default_params(128, 96, **kw)
    frames, _ = synth.make_sequence(p.width, p.height, 9, 4, seed=21, max_step=3)
    po = orc.params_from(p)
    outs = []
    for graph in (True, False):
        eng = aof.FlowEngine(p, 0)
        eng.set_stream_graph(graph)
        assert eng.stream_push(frames[0]) is None
        got = [eng.stream_push(frames[k]).tobytes() for k in range(1, 9)]
        assert eng.stream_graph_active() == graph
        outs.append(got)
        eng.close()
    assert outs[0] == outs[1]
    for k in range(1, 9):
        assert outs[0][k - 1] == orc.flow_pair(po, frames[k - 1], frames[k])["flow"].tobytes()


@pytest.mark.parametrize("kw", [dict(), dict(pyramid_levels=2, mean_subtract=1)])
def test_per_call_graph_polls_a_tagged_record(aof, orc, synth, gpu_device, kw):
    """Small frames: the replayed graph is ONE kernel that publishes the 16-byte record in pinned memory with
    a tag in the top byte of `count`, and the host polls for the tag instead of waiting for the stream.
    600 calls take the tag through its 256 values twice; `count` must come back without it, calls through
    the other host entry point (which share the stream and the pinned record) may come in between, and a
    context destroyed right after a call must not trip over a launch the runtime has not retired yet."""
    p = aof.px4flow_params(64, 64, **kw)
    frames, _ = synth.make_sequence(64, 64, 7, 4, seed=77, max_step=3)
    po = orc.params_from(p)
    want = {}
    eng = aof.FlowEngine(p, 0)
    assert eng.stream_push(frames[0]) is None
    prev = 0
    for call in range(600):
        k = (prev + 1 + call % 3) % 7
        if (prev, k) not in want:
            want[(prev, k)] = orc.flow_pair(po, frames[prev], frames[k])["flow"]
        got = eng.stream_push(frames[k])
        assert got.tobytes() == want[(prev, k)].tobytes(), (call, prev, k)
        assert int(got["count"]) < (1 << 24)
        prev = k
        if call % 97 == 5:
            _, _, pair = eng.flow_pair_host(frames[1], frames[2])
            assert pair.tobytes() == orc.flow_pair(po, frames[1], frames[2])["flow"].tobytes()
    assert eng.stream_graph_active()
    eng.close()
    for _ in range(3):   # destroyed with the last launch barely retired
        e2 = aof.FlowEngine(p, 0)
        e2.stream_push(frames[0])
        e2.stream_push(frames[1])
        e2.close()


@pytest.mark.parametrize("kw", [dict(px4=1), dict(px4=1, pyramid_levels=2, mean_subtract=1),
                                dict(px4=1, size=(128, 128), pyramid_levels=2, mean_subtract=1),
                                dict(size=(128, 96), mean_subtract=1), dict(px4=1, size=(96, 80), subpixel=0, hist_filter=0)])
@pytest.mark.timeout(120)
def test_resident_kernel_serves_the_streaming_entry_point(aof, orc, synth, gpu_device, kw, capfd):
    """aof_set_stream_resident: ONE workgroup stays on the device and takes the frames through a mailbox
    in pinned memory.  Every record must equal the oracle's (and hence the launch-per-call paths'):
    consecutive frames (the older frame stays in LDS, only the new one crosses PCIe), a stream reset,
    a frame that takes the graph path in between (the LDS copy must not be trusted afterwards), a
    kernel switch (the resident kernel must step aside), a pause longer than its idle deadline (it
    leaves by itself and the next call starts it again), and destruction while it is on the device."""
    import time
    kw = dict(kw)
    W, H = kw.pop("size", (64, 64))
    p = aof.px4flow_params(W, H, **kw) if kw.pop("px4", 0) else aof.default_params(W, H, **kw)
    frames, _ = synth.make_sequence(W, H, 40, 4, seed=51, max_step=3)
    po = orc.params_from(p)
    eng = aof.FlowEngine(p, 0)
    eng.set_stream_resident(True)
    prev = None

    def push(k, expect_resident=None):
        nonlocal prev
        got = eng.stream_push(frames[k])
        if prev is None:
            assert got is None
        else:
            assert got.tobytes() == orc.flow_pair(po, frames[prev], frames[k])["flow"].tobytes(), (k, prev)
        prev = k
        if expect_resident is not None and eng.stream_resident_running() != expect_resident:
            # An unanswered request is a failure of this test, with the library's own account of it: the
            # stderr line and aof_stream_stats (launch call time, launch -> first poll latency, which of
            # "never started" / "started but silent" / "left" it was).
            raise AssertionError((k, expect_resident, eng.stream_stats(), capfd.readouterr().err))

    push(0)
    for k in range(1, 8):
        push(k, True)
    push(5, True)                      # any frame may follow any frame
    eng.stream_reset()
    prev = None
    push(9)                            # first frame again: no request
    push(10, True)                     # its predecessor was not posted through a request: both frames are fetched
    push(11, True)
    eng.set_stream_resident(False)     # one frame through the launch-per-call path ...
    assert not eng.stream_resident_running()
    push(12, False)
    eng.set_stream_resident(True)      # ... whose pinned frame the resident kernel never saw
    push(13, True)
    push(14, True)
    eng.force_generic(True)            # the one-workgroup kernel does not serve this selection
    assert not eng.stream_resident_running()
    push(15, False)
    eng.force_generic(False)
    push(16, True)
    time.sleep(0.12)                   # beyond the 50 ms idle deadline
    assert not eng.stream_resident_running(), "the kernel must leave by itself when no frame comes"
    push(17, True)
    for k in range(18, 40):            # 22 more calls; the 200 ms lifetime does not matter to results
        push(k)
    assert eng.stream_resident_running()
    st = eng.stream_stats()
    assert st["resident_fallbacks"] == 0 and st["resident_lost"] == 0 and st["tagged_slow"] == 0, st
    assert st["resident_launches"] >= 4, st   # first start, after the two stops, after the idle exit (+ lifetime restarts)
    assert st["resident_served"] == 36 and st["calls"] == 38, st   # two frames took the launch-per-call paths
    eng.close()                        # stops the kernel first
    assert "aof:" not in capfd.readouterr().err


def test_a_resident_kernel_that_does_not_leave_costs_one_sample_and_never_a_hang(aof, orc, synth, gpu_device, capfd):
    """Fault injection (aof_debug_resident_fault): the resident kernel ignores the request to leave and the library
    waits 5 ms instead of a second.  Asked to step aside it is declared lost: one line on stderr, its box, stream and
    pinned buffers abandoned.  The context then continues on fresh buffers -- the next frame starts a new sequence
    (no flow, like the first frame), every later record equals the oracle's --, the counters say so, and aof_destroy
    returns at once (it leaks instead of freeing under a kernel that may still run).  The zombie leaves by itself on
    its 50 ms idle deadline; a new context afterwards works as if nothing had happened."""
    import time
    p = aof.px4flow_params(64, 64)
    frames, _ = synth.make_sequence(64, 64, 16, 4, seed=61, max_step=3)
    po = orc.params_from(p)
    eng = aof.FlowEngine(p, 0)
    eng.set_stream_resident(True)
    eng.debug_resident_fault(True, 5000)
    assert eng.stream_push(frames[0]) is None
    for k in range(1, 5):
        assert eng.stream_push(frames[k]).tobytes() == orc.flow_pair(po, frames[k - 1], frames[k])["flow"].tobytes()
    assert eng.stream_resident_running()
    t0 = time.perf_counter()
    eng.set_stream_resident(False)                       # the deaf kernel does not leave within 5 ms
    assert time.perf_counter() - t0 < 0.5
    err = capfd.readouterr().err
    assert "did not leave" in err and "abandoned" in err
    st = eng.stream_stats()
    assert st["resident_lost"] == 1 and st["resident_fallbacks"] == 0
    assert eng.stream_push(frames[5]) is None            # fresh buffers: a new sequence starts
    for k in range(6, 12):
        assert eng.stream_push(frames[k]).tobytes() == orc.flow_pair(po, frames[k - 1], frames[k])["flow"].tobytes()
    _, _, pair = eng.flow_pair_host(frames[2], frames[3])   # the other host entry point shares the rebuilt state
    assert pair.tobytes() == orc.flow_pair(po, frames[2], frames[3])["flow"].tobytes()
    eng.set_stream_resident(True)                        # the mode can be switched on again: a new box, a new (deaf) kernel
    eng.debug_resident_fault(False, 0)
    assert eng.stream_push(frames[12]).tobytes() == orc.flow_pair(po, frames[11], frames[12])["flow"].tobytes()
    t0 = time.perf_counter()
    eng.close()
    assert time.perf_counter() - t0 < 1.5
    assert "leaked, not freed" in capfd.readouterr().err
    time.sleep(0.25)                                     # (every zombie has gone on its deadlines by now)
    e2 = aof.FlowEngine(p, 0)
    e2.set_stream_resident(True)
    assert e2.stream_push(frames[0]) is None
    assert e2.stream_push(frames[1]).tobytes() == orc.flow_pair(po, frames[0], frames[1])["flow"].tobytes()
    e2.close()
    assert "aof:" not in capfd.readouterr().err


def test_streaming_path_follows_kernel_switches_mid_sequence(aof, orc, synth, gpu_device):
    """The streaming entry point replays captured hipGraphs: switching the kernel selection
    (aof_set_force_generic, aof_set_search_mode, aof_set_split_coarse) between two frames must
    drop them, so that the next frame really runs the newly selected kernels -- and of course all
    of them give the oracle's record."""
    p = aof.default_params(160, 128, pyramid_levels=2, mean_subtract=1)
    frames, _ = synth.make_sequence(p.width, p.height, 14, 6, seed=33, max_step=5)
    po = orc.params_from(p)
    eng = aof.FlowEngine(p, 0)
    assert eng.stream_push(frames[0]) is None
    switches = {3: lambda: eng.force_generic(True), 5: lambda: eng.force_generic(False),
                7: lambda: eng.set_search_mode(aof.SEARCH_PRUNED), 9: lambda: eng.set_split_coarse(True),
                11: lambda: (eng.set_split_coarse(False), eng.set_search_mode(aof.SEARCH_EXHAUSTIVE))}
    for k in range(1, 14):
        if k in switches:
            was_active = eng.stream_graph_active()
            switches[k]()
            assert was_active and not eng.stream_graph_active(), "a kernel switch must drop the captured graphs"
            assert eng.variant == ("generic" if 3 <= k < 5 else "lane8")
        got = eng.stream_push(frames[k])
        assert got.tobytes() == orc.flow_pair(po, frames[k - 1], frames[k])["flow"].tobytes(), k
    eng.close()


def test_c_abi_argument_handling(aof, orc, synth, gpu_device):
    """Direct calls into the C ABI: optional outputs, error codes, many small pairs."""
    import ctypes as C
    import torch
    p = aof.default_params(64, 48)
    eng = aof.FlowEngine(p, 0)
    n = 3000                                         # many tiny pairs in one launch
    prevs, curs, shifts = synth.make_batch(64, 48, 8, 4, 4100)
    tp = torch.from_numpy(np.tile(prevs, (n // 8, 1, 1))).to(gpu_device)
    tc = torch.from_numpy(np.tile(curs, (n // 8, 1, 1))).to(gpu_device)
    L = aof.workspace_layout(p, n)
    ws = torch.zeros(L.total_bytes + 256, dtype=torch.uint8, device=gpu_device)
    flows = torch.zeros((n, 16), dtype=torch.uint8, device=gpu_device)
    stream = torch.cuda.current_stream().cuda_stream
    call = aof.lib.aof_flow_batch_device
    # d_blocks == NULL: records land in the workspace
    rc = call(eng._ctx, tp.data_ptr(), tc.data_ptr(), 64 * 48, n, None, None, flows.data_ptr(),
              ws.data_ptr(), L.total_bytes, stream)
    assert rc == 0
    torch.cuda.synchronize()
    f = aof.flows_view(flows)
    assert np.array_equal(f["flow_x"], np.tile(shifts[:, 0], n // 8).astype(np.float32))
    nb = eng.nblocks(0)
    rec = ws.cpu().numpy()[L.l0_blocks:L.l0_blocks + n * nb * 4].view(aof.BLOCK_DTYPE).reshape(n, nb)
    ref = orc.flow_pair(orc.params_from(p), prevs[5], curs[5])
    assert rec[5].tobytes() == ref["blocks"].tobytes() and rec[2997].tobytes() == ref["blocks"].tobytes()
    # errors: misaligned workspace, missing flows, short pair stride
    assert call(eng._ctx, tp.data_ptr(), tc.data_ptr(), 64 * 48, n, None, None, flows.data_ptr(),
                ws.data_ptr() + 16, L.total_bytes, stream) == -22
    assert b"aligned" in aof.lib.aof_last_error(eng._ctx)
    assert call(eng._ctx, tp.data_ptr(), tc.data_ptr(), 64 * 48, n, None, None, None,
                ws.data_ptr(), L.total_bytes, stream) == -22
    assert call(eng._ctx, tp.data_ptr(), tc.data_ptr(), 100, n, None, None, flows.data_ptr(),
                ws.data_ptr(), L.total_bytes, stream) == -22
    assert call(eng._ctx, tp.data_ptr(), tc.data_ptr(), 64 * 48, n, None, None, flows.data_ptr(),
                ws.data_ptr(), 1024, stream) == -28
    assert call(None, tp.data_ptr(), tc.data_ptr(), 64 * 48, n, None, None, flows.data_ptr(),
                ws.data_ptr(), L.total_bytes, stream) == -22
    # profiling ring: counts launches of the timed kernels only
    eng.set_profiling(True)
    for _ in range(3):
        eng.flow_batch(tp[:16], tc[:16])
    assert len(eng.profile_ms(aof.K_SEARCH)) == 3 and len(eng.profile_ms(aof.K_PYRAMID)) == 0
    assert all(0 < ms < 50 for ms in eng.profile_ms(aof.K_SEARCH))
    eng.set_profiling(False)
    par = aof.Params()
    assert aof.lib.aof_get_params(eng._ctx, C.byref(par)) == 0 and par.width == 64 and par.search == 4


def test_maximum_frame_size_and_sum_range(aof, orc, gpu_device):
    """Largest frame the ABI accepts (2^24 pixels): the u32 pixel sums of K1 peak at
    255 * 2^24 < 2^32, and the block grid has 261 121 tiles."""
    import torch
    w = h = 4096
    assert aof.check_params(aof.default_params(w, h)) == 0
    assert aof.check_params(aof.default_params(w, h + 2)) != 0
    p = aof.default_params(w, h, pyramid_levels=2, mean_subtract=1)
    rng = np.random.default_rng(1)
    prev = np.full((h, w), 255, np.uint8)
    cur = np.full((h, w), 254, np.uint8)
    # one textured patch so that some blocks are searched, shifted by (3, -2)
    patch = rng.integers(0, 256, (200, 300), dtype=np.uint8)
    prev[1000:1200, 2000:2300] = patch
    cur[998:1198, 2003:2303] = np.clip(patch.astype(np.int32) - 1, 0, 255).astype(np.uint8)
    tp, tc = torch.from_numpy(prev[None]).to(gpu_device), torch.from_numpy(cur[None]).to(gpu_device)
    eng = aof.FlowEngine(p, 0)
    blocks, flows, ws = eng.flow_batch(tp, tc)
    torch.cuda.synchronize()
    L = aof.workspace_layout(p, 1)
    sums = ws.cpu().numpy()[L.sums:L.sums + 16].view(np.uint32).reshape(2, 2)
    assert sums[0, 0] == prev.sum(dtype=np.uint64) and sums[1, 0] == cur.sum(dtype=np.uint64)
    assert sums[0, 0] > 0xFE000000, "the test must sit at the top of the u32 range"
    ref = orc.flow_pair(orc.params_from(p), prev, cur)
    gb, gf = aof.blocks_view(blocks)[0], aof.flows_view(flows)[0]
    assert gb.tobytes() == ref["blocks"].tobytes() and gf.tobytes() == ref["flow"].tobytes()
    assert gf["flow_x"] == 3 and gf["flow_y"] == -2 and gf["count"] > 100


def test_two_contexts_on_two_streams(aof, orc, synth, gpu_device):
    """Two engines with different configurations enqueue on two HIP streams at once
    (a multi-camera server); nothing in a context is shared, results stay bit-exact."""
    import torch
    pa = aof.default_params(640, 480)
    pb = aof.default_params(320, 240, pyramid_levels=2, mean_subtract=1)
    ea, eb = aof.FlowEngine(pa, 0), aof.FlowEngine(pb, 0)
    eb.set_search_mode(aof.SEARCH_PRUNED)
    a_prev, a_cur, _ = synth.make_batch(640, 480, 6, 4, 6100, noise=3)
    b_prev, b_cur, _ = synth.make_batch(320, 240, 9, 9, 6200, noise=3, brightness=7)
    ta = [torch.from_numpy(x).to(gpu_device) for x in (a_prev, a_cur)]
    tb = [torch.from_numpy(x).to(gpu_device) for x in (b_prev, b_cur)]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):
        with torch.cuda.stream(sa):
            ra = ea.flow_batch(ta[0], ta[1])
        with torch.cuda.stream(sb):
            rb = eb.flow_batch(tb[0], tb[1])
        outs.append((ra, rb))
    torch.cuda.synchronize()
    (ba, fa, _), (bb, fb, _) = outs[-1]
    check_against_oracle(aof, orc, pa, a_prev, a_cur, dict(blocks=aof.blocks_view(ba), flows=aof.flows_view(fa)))
    check_against_oracle(aof, orc, pb, b_prev, b_cur, dict(blocks=aof.blocks_view(bb), flows=aof.flows_view(fb)))


@pytest.mark.gpu
def test_wide_16x16_frames_keep_the_lds_tile_when_only_the_pruned_tables_do_not_fit(aof, orc, synth, gpu_device):
    """at widths of about 3 000 pixels the pruned steps' tables (68 B per block column) push the 16x16 kernel's
    LDS tile past 156 KB while the exhaustive tile still fits; the default (ADAPTIVE) context then runs the exhaustive
    LDS kernel instead of dropping to the generic one."""
    W, H = 3072, 64
    p = aof.default_params(W, H, tile=16, search=8, value_threshold=12000, min_valid=0)
    prevs, curs, _ = synth.make_batch(W, H, 2, 8, 9400, noise=2)
    eng = aof.FlowEngine(p, 0)
    assert eng.search_mode == aof.SEARCH_ADAPTIVE and eng.variant == "tile16_lds"
    blocks, subdirs, flow = eng.flow_pair_host(prevs[0], curs[0])
    ref = orc.flow_pair(orc.params_from(p), prevs[0], curs[0])
    assert blocks.tobytes() == ref["blocks"].tobytes() and flow.tobytes() == ref["flow"].tobytes()
    eng.force_generic(True)
    b2, _, f2 = eng.flow_pair_host(prevs[0], curs[0])
    assert b2.tobytes() == ref["blocks"].tobytes() and f2.tobytes() == ref["flow"].tobytes()
    eng.close()


@pytest.mark.gpu
def test_a_faulted_context_launches_nothing_through_the_sequence_pipeline(aof, synth, gpu_device):
    """aof_sequence_device checks the context's sticky state and the calling thread's device BEFORE its first
    launch (the ingest kernel), and says why through aof_last_error."""
    import torch
    p = aof.default_params(640, 480)
    hp, hc, _ = synth.make_batch(640, 480, 8, 4, 4300)
    idx = np.arange(256) % 8
    eng = aof.FlowEngine(p, 0)
    eng.set_search_mode(aof.SEARCH_EXHAUSTIVE)
    eng.set_reduce_fusion(True)
    eng.debug_vote_deadline_ticks(0)
    eng.flow_batch(torch.from_numpy(hp[idx]).to(gpu_device), torch.from_numpy(hc[idx]).to(gpu_device))
    torch.cuda.synchronize()
    frames = torch.full((4, 480, 640), 0x5A, dtype=torch.uint8, device=gpu_device)
    times = torch.arange(4, dtype=torch.int64, device=gpu_device) * 40000
    sp = aof.sequence_params(640, 480, 640, 480, 216.6677, 216.2457, 15, 5_000_000, 1, 100, 0)
    L = aof.sequence_layout(p, sp, 4)
    ws = torch.full((L.total_bytes,), 0xC3, dtype=torch.uint8, device=gpu_device)
    with pytest.raises(aof.AofError) as e:
        eng.sequence(sp, frames, times, workspace=ws)
    assert e.value.code == -5 and "deadline" in str(e.value)
    torch.cuda.synchronize()
    assert bool((ws == 0xC3).all()), "the faulted context must not have launched anything into the workspace"
    eng.close()
