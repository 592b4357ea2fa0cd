"""Seeded differential fuzz on the GPU: random geometry / options / image statistics,
HIP path (fast kernels where they apply, and the generic kernel forced) against the oracle.
Bit-exact or fail."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_case(rng):
    tile = int(rng.choice([8, 8, 8, 16]))
    search = int(rng.choice([4, 4, 4, 8 if tile == 16 else 4, rng.integers(1, 9)]))
    levels = int(rng.choice([1, 1, 2]))
    grid_mode = int(rng.choice([0, 0, 0, 1]))
    subpixel = int(rng.choice([0, 0, 1])) if grid_mode == 0 else 1
    margin = search + 1
    min_dim = (tile + 2 * margin + 8) * (2 if levels == 2 else 1)
    w = int(rng.integers(min_dim, min_dim + 160))
    h = int(rng.integers(min_dim, min_dim + 120))
    if rng.random() < 0.6:
        w = (w + 15) // 16 * 16          # let the LDS-tiled kernels qualify most of the time
    if levels == 2:
        w += w & 1
        h += h & 1
    kw = dict(width=w, height=h, tile=tile, search=search, pyramid_levels=levels, grid_mode=grid_mode,
              subpixel=subpixel, mean_subtract=int(rng.integers(0, 2)), hist_filter=int(rng.integers(0, 2)),
              feature_threshold=int(rng.choice([0, 30, 30, 200])),
              value_threshold=int(rng.choice([0, 500, 3000, 3000, 70000])) * (4 if tile == 16 else 1),
              min_valid=int(rng.choice([0, 10, 10, 50])), num_blocks=int(rng.integers(2, 9)))
    return kw


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_against_oracle(aof, orc, synth, gpu_device, seed):
    import torch
    rng = np.random.default_rng(9000 + seed)
    kw = random_case(rng)
    p = aof.default_params(**kw)
    if aof.check_params(p) != 0:
        pytest.skip("geometry too small for this option mix")
    reach = 2 * p.search + 1 if p.pyramid_levels == 2 else p.search
    n = 3
    prevs, curs, _ = synth.make_batch(p.width, p.height, n, reach, 7000 + 13 * seed,
                                      noise=int(rng.integers(0, 12)), brightness=int(rng.integers(-30, 31)),
                                      contrast=float(rng.choice([1.0, 1.0, 2.0, 0.3])))
    style = seed % 4
    if style == 1:
        curs[0] = rng.integers(0, 256, curs[0].shape, dtype=np.uint8)      # unrelated frame
    elif style == 2:
        prevs[1][:, : p.width // 2] = int(rng.integers(0, 256))            # flat half
    elif style == 3:
        curs[2] = prevs[2]                                                   # identical frames
    po = orc.params_from(p)
    refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(n)]
    tp, tc = torch.from_numpy(prevs).to(gpu_device), torch.from_numpy(curs).to(gpu_device)
    for force in (False, True):
        eng = aof.FlowEngine(p, 0)
        if force:
            eng.force_generic(True)
        nb = eng.nblocks(0)
        sub = torch.full((n, nb), 99, dtype=torch.uint8, device=gpu_device) if p.subpixel else None
        blocks, flows, _ = eng.flow_batch(tp, tc, subdirs=sub)
        torch.cuda.synchronize()
        gb, gf = aof.blocks_view(blocks), aof.flows_view(flows)
        for i in range(n):
            assert gb[i].tobytes() == refs[i]["blocks"].tobytes(), (kw, force, i, eng.variant)
            assert gf[i].tobytes() == refs[i]["flow"].tobytes(), (kw, force, i, gf[i], refs[i]["flow"])
            if sub is not None:
                assert np.array_equal(sub[i].cpu().numpy(), refs[i]["subdirs"]), (kw, force, i)
        eng.close()
