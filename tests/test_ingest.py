"""Frame ingest: centre crop + 10-bin auto-exposure histogram.

Unlike the flow engine, the reference code for these steps IS in the mount
(/root/reference/src/mainloop.cpp:295-298 crop, :203-220 masked calcHist + MSV), so the
oracle follows those lines; cv::calcHist's binning is restated from OpenCV's uniform
8-bit lookup (floor(v * histSize/(high-low)) in double, half-open range)."""
import math

import numpy as np
import pytest


def np_ingest(cam, crop_w, crop_h):
    h, w = cam.shape
    x0, y0 = w // 2 - crop_w // 2, h // 2 - crop_h // 2
    crop = cam[y0:y0 + crop_h, x0:x0 + crop_w].copy()
    mx0, my0 = max(crop_w // 2 - 64, 0), max(crop_h // 2 - 64, 0)
    mx1, my1 = min(crop_w // 2 + 64, crop_w), min(crop_h // 2 + 64, crop_h)
    m = crop[my0:my1, mx0:mx1].astype(np.float64)
    idx = np.floor(m * (10 / 255.0)).astype(np.int64)
    hist = np.bincount(idx[idx < 10].ravel(), minlength=10).astype(np.uint32)
    return crop, hist


def test_exposure_bin_matches_calchist_formula(orc, aof):
    a = 10 / 255.0
    for v in range(256):
        exp = math.floor(v * a)
        exp = exp if exp < 10 else -1
        assert orc.exposure_bin(v) == exp
        assert aof.lib.aof_exposure_bin(v) == exp, v   # integer formula in the product
    assert orc.exposure_bin(255) == -1 and orc.exposure_bin(254) == 9 and orc.exposure_bin(0) == 0
    assert [orc.exposure_bin(v) for v in (25, 26, 51, 102, 153, 204)] == [0, 1, 2, 4, 6, 8]


def test_oracle_ingest_against_numpy(orc):
    rng = np.random.default_rng(3)
    for (cw, ch), (w, h) in [((128, 128), (320, 240)), ((128, 128), (640, 480)), ((64, 64), (160, 120)),
                             ((100, 90), (322, 242)), ((640, 480), (640, 480)), ((200, 130), (320, 240))]:
        cam = rng.integers(0, 256, (h, w), dtype=np.uint8)
        cam[h // 2 - 5:h // 2 + 5, w // 2 - 5:w // 2 + 5] = 255   # out-of-range values inside the mask
        crop, hist = orc.ingest(cam, cw, ch)
        ecrop, ehist = np_ingest(cam, cw, ch)
        assert np.array_equal(crop, ecrop) and np.array_equal(hist, ehist)
        mask_px = min(cw, 128) * min(ch, 128)
        assert hist.sum() == mask_px - np.count_nonzero(ecrop[max(ch // 2 - 64, 0):ch // 2 + 64, max(cw // 2 - 64, 0):cw // 2 + 64] == 255)


def test_msv_hand_and_product(orc, aof):
    hist = np.zeros(10, np.uint32)
    hist[4] = 16384            # every mask pixel in bin 4 -> MSV 5.0 = EXPOSURE_MSV_TARGET (mainloop.cpp:53)
    assert orc.exposure_msv(hist) == 5.0 and aof.exposure_msv(hist) == 5.0
    hist = np.arange(10, dtype=np.uint32) * 100 + 7
    exp = np.float32(0)
    for i in range(10):
        exp = np.float32(exp + np.float32(i + 1) * np.float32(hist[i]) / np.float32(16384.0))
    assert np.float32(orc.exposure_msv(hist)) == exp
    assert np.float32(aof.exposure_msv(hist)) == exp


def test_ingest_rejects_bad_geometry(aof):
    p = aof.IngestParams(64, 64, 128, 128)
    assert aof.lib.aof_ingest_batch_device(p, None, 0, 1, None, 0, None, None) == -22


@pytest.mark.gpu
@pytest.mark.parametrize("cam,crop", [((320, 240), (128, 128)), ((640, 480), (128, 128)),
                                      ((160, 120), (64, 64)), ((322, 242), (100, 90)),
                                      ((640, 480), (640, 480)), ((1280, 960), (256, 192)),
                                      ((320, 240), (144, 136)), ((640, 480), (400, 128))])   # mask edges inside 16-byte pieces; 12.5 pieces per lane
def test_gpu_ingest_parity(aof, orc, gpu_device, cam, crop):
    import torch
    rng = np.random.default_rng(cam[0] + crop[0])
    n = 5
    frames = rng.integers(0, 256, (n, cam[1], cam[0]), dtype=np.uint8)
    frames[1] = 255                                   # everything out of range
    frames[2] = 37                                    # a single bin: worst-case vote contention
    frames[3, cam[1] // 2 - 20:cam[1] // 2 + 20] = 255
    t = torch.from_numpy(frames).to(gpu_device)
    cropped, hist = aof.ingest_batch(t, crop[0], crop[1])
    torch.cuda.synchronize()
    cropped, hist = cropped.cpu().numpy(), hist.cpu().numpy().view(np.uint32)
    for i in range(n):
        ecrop, ehist = orc.ingest(frames[i], crop[0], crop[1])
        assert np.array_equal(cropped[i], ecrop), i
        assert np.array_equal(hist[i], ehist), (i, hist[i], ehist)
        assert np.float32(aof.exposure_msv(hist[i])) == np.float32(orc.exposure_msv(ehist))
    # crop-only and histogram-only forms
    c2, h2 = aof.ingest_batch(t, crop[0], crop[1], want_hist=False)
    assert h2 is None and np.array_equal(c2.cpu().numpy(), cropped)


@pytest.mark.gpu
def test_gpu_ingest_feeds_the_flow_engine(aof, orc, synth, gpu_device):
    """Sensor frames -> device crop -> flow, the order of mainloop.cpp:295-322, all resident."""
    import torch
    cam_w, cam_h, crop = 320, 240, 128
    frames, steps = synth.make_sequence(cam_w, cam_h, 7, 4, seed=8, max_step=3)
    t = torch.from_numpy(frames).to(gpu_device)
    cropped, hist = aof.ingest_batch(t, crop, crop)
    p = aof.px4flow_params(crop, crop)
    eng = aof.FlowEngine(p, 0)
    blocks, flows, _ = eng.flow_batch(cropped[:-1], cropped[1:], n_pairs=6, pair_stride=crop * crop)
    torch.cuda.synchronize()
    f = aof.flows_view(flows)
    po = orc.params_from(p)
    x0, y0 = cam_w // 2 - crop // 2, cam_h // 2 - crop // 2
    for k in range(6):
        ref = orc.flow_pair(po, frames[k, y0:y0 + crop, x0:x0 + crop], frames[k + 1, y0:y0 + crop, x0:x0 + crop])
        assert f[k].tobytes() == ref["flow"].tobytes()
    assert np.array_equal(f["flow_x"], steps[:, 0].astype(np.float32))
