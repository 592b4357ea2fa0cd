"""aof_sequence_device: a recorded frame sequence through ingest -> sequence-mode flow -> rate limiter ->
de-rotation -> OPTICAL_FLOW_RAD frames in ONE enqueue, against the reference's own per-frame loop
(/root/reference/src/mainloop.cpp:295-373) replayed two independent ways over the same frames:
  A  the C++ facade driven frame by frame (crop on the host, OpticalFlowPX4 / OpticalFlowOpenCV::calcFlow with
     the 32-bit time stamp, the negative-return gate, fillOpticalFlowRad + packOpticalFlowRad) -- the GPU's
     per-call path;
  B  the CPU oracle's calcFlow chain and the independent MAVLink serializer of tests/test_mavlink.py.
The bar: records and wire frames byte-identical to both ("parity unpinned": the oracle is this repo's
restatement; upstream PX4 / mavlink_c sources are absent from the reference mount)."""
import numpy as np
import pytest

from test_mavlink import py_frame

pytestmark = pytest.mark.gpu

FX, FY = 216.6677, 216.2457   # /root/reference/src/main.cpp:60-61


def make_inputs(synth, cam_w, cam_h, n, seed, times):
    frames, _ = synth.make_sequence(cam_w, cam_h, n, 4, seed=seed, max_step=3)
    rng = np.random.default_rng(seed)
    gyro = np.zeros((n, 4), np.float32)
    gyro[:, :3] = rng.normal(0, 0.004, (n, 3)).astype(np.float32)
    dt = np.diff(np.concatenate([[0], times])).astype(np.float64) * 1e-6
    gyro[:, 3] = np.clip(dt, 0, 1).astype(np.float32)
    return frames, gyro


def crop_of(frames, cw, ch):
    n, H, W = frames.shape
    x0, y0 = W // 2 - cw // 2, H // 2 - ch // 2            # mainloop.cpp:295-297
    return np.ascontiguousarray(frames[:, y0:y0 + ch, x0:x0 + cw])


def replay(calc_flow, cropped, times, gyro, offset, first_seq, pack):
    """mainloop.cpp:322-373 around a calcFlow implementation: the negative-return gate, the gyro taken and
    zeroed with every published flow, the field mapping and the frame."""
    recs, wire = [], []
    g = np.zeros(3, np.float64)
    seq = first_seq
    for k in range(len(times)):
        g += gyro[k, :3].astype(np.float64)                 # integrated since the last message (:383-405)
        q, dt, ax, ay = calc_flow(cropped[k], int(times[k]) & 0xFFFFFFFF)
        if q < 0:                                           # :327-331
            continue
        taken, g = g.copy(), np.zeros(3, np.float64)        # :333-334
        recs.append((k, q, dt, np.float32(ax), np.float32(ay), np.float32(taken[0]), np.float32(taken[1]), np.float32(taken[2])))
        if offset:                                          # :353-357
            wire.append(pack(offset, int(times[k]), dt, float(np.float32(ax)), float(np.float32(ay)),
                             tuple(float(v) for v in taken), q, seq & 0xFF))
            seq += 1
    return recs, wire


CASES = [
    # (facade class, crop, camera, frames, output rate, time stamps)
    dict(cls="px4", crop=(64, 64), cam=(160, 120), n=90, rate=15, seed=3),
    dict(cls="opencv", crop=(128, 128), cam=(320, 240), n=75, rate=15, seed=4),
    dict(cls="px4", crop=(64, 64), cam=(160, 120), n=40, rate=0, seed=5),                 # no limit: every frame publishes
    dict(cls="px4", crop=(64, 64), cam=(64, 64), n=60, rate=30, seed=6, wrap=True),       # crop == sensor; u32 wrap of the time stamps
    dict(cls="px4", crop=(64, 64), cam=(160, 120), n=33, rate=10, seed=7, offset=0),      # vehicle time unknown: nothing is sent
    dict(cls="opencv", crop=(128, 128), cam=(320, 240), n=50, rate=200, seed=8, dark=True),   # rate above the frame rate; frames without valid flow
    # more than 128 pairs: the flow runs its separate kernels, and the ingest kernel leaves K1's outputs (pixel sums,
    # one level-1 frame per frame) itself -- crop window at an odd byte offset of the sensor rows, and crop == sensor
    dict(cls="opencv", crop=(128, 128), cam=(322, 250), n=260, rate=15, seed=9, dark=True),
    dict(cls="opencv", crop=(128, 128), cam=(128, 128), n=200, rate=30, seed=10),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['cls']}-{c['crop'][0]}-rate{c['rate']}-n{c['n']}")
def test_sequence_pipeline_equals_the_facade_frame_by_frame_and_the_oracle_chain(aof, orc, synth, gpu_device, case):
    import torch
    cw, ch = case["crop"]
    cam_w, cam_h = case["cam"]
    n, rate = case["n"], case["rate"]
    rng = np.random.default_rng(100 + case["seed"])
    times = np.cumsum(np.concatenate([[0], rng.integers(9000, 18000, n - 1)])).astype(np.int64)   # ~75 fps with jitter
    if case.get("wrap"):
        times[n // 2:] += (1 << 32) - int(times[n // 2]) - 20000      # the 32-bit time stamp wraps mid-sequence (mainloop.cpp:313-315)
    frames, gyro = make_inputs(synth, cam_w, cam_h, n, case["seed"], times)
    if case.get("dark"):
        frames[10:14] = 0                                             # flat frames: no block passes the gate, quality 0
    offset = case.get("offset", 5_000_000)
    first_seq = 250                                                   # (wraps through 255 -> 0)
    cropped = crop_of(frames, cw, ch)

    # ---- the device pipeline, one call ----
    if case["cls"] == "px4":
        p = aof.px4flow_params(cw, ch)
    else:   # what OpticalFlowOpenCV's constructor selects at this size (optical_flow.cpp: two levels + equalisation)
        p = aof.px4flow_params(cw, ch, pyramid_levels=2, mean_subtract=1)
    eng = aof.FlowEngine(p, 0)
    sp = aof.sequence_params(cam_w, cam_h, cw, ch, FX, FY, rate, offset, 1, 100, first_seq, derotate=(4.5, 0.01))
    ws, L = eng.sequence(sp, torch.from_numpy(frames).to(gpu_device), torch.from_numpy(times).to(gpu_device),
                         torch.from_numpy(gyro).to(gpu_device))
    torch.cuda.synchronize()
    out = eng.sequence_outputs(sp, ws, L, n)
    assert out["status"] == 0
    assert np.array_equal(out["cropped"], cropped)

    # ---- A: the C++ facade, frame by frame ----
    if case["cls"] == "px4":
        fac = aof.OpticalFlowPX4(FX, FY, rate, cw, ch)
    else:
        fac = aof.OpticalFlowOpenCV(FX, FY, rate, cw, ch)
        assert fac.getPyramidLevels() == 2
    recs_a, wire_a = replay(lambda img, t: fac.calcFlow(img, t), cropped, times, gyro, offset, first_seq, aof.pack_optical_flow_rad)
    fac.close()

    # ---- B: the oracle's calcFlow chain + the independent serializer ----
    o = orc.Px4(orc.params_from(p), FX, FY, rate)
    recs_b, wire_b = replay(lambda img, t: o.calc_flow(img, t), cropped, times, gyro, offset, first_seq, py_frame)

    got = out["records"]
    assert len(got) == len(recs_a) == len(recs_b) and len(got) >= 2
    for m, (ra, rb) in enumerate(zip(recs_a, recs_b)):
        g = got[m]
        tup = (int(g["frame"]), int(g["quality"]), int(g["dt_us"]), g["flow_x"], g["flow_y"], g["gyro_x"], g["gyro_y"], g["gyro_z"])
        for ref, name in ((ra, "facade"), (rb, "oracle")):
            assert tup[:3] == ref[:3], (name, m, tup, ref)
            assert b"".join(np.float32(v).tobytes() for v in tup[3:]) == b"".join(np.float32(v).tobytes() for v in ref[3:]), (name, m, tup, ref)
    assert out["frames_sent"] == len(wire_a) == len(wire_b)
    if offset:
        assert out["mavlink"] == wire_a, "device frames differ from the C++ facade's"
        assert out["mavlink"] == wire_b, "device frames differ from the oracle chain's"
        assert [f[4] for f in out["mavlink"]] == [(first_seq + m) & 0xFF for m in range(len(wire_a))]
    else:
        assert all(len(f) == 0 for f in out["mavlink"])
    if rate > 0 and not case.get("wrap"):
        published = got["frame"][1:]
        assert len(published) < n - 1 or rate >= 75, "the limiter must hold frames back"

    # ---- the intermediates: per-pair flows, exposure histograms, de-rotated flows vs the oracle ----
    po = orc.params_from(p)
    for k in range(0, n - 1, max(1, (n - 1) // 12)):
        assert out["flows"][k].tobytes() == orc.flow_pair(po, cropped[k], cropped[k + 1])["flow"].tobytes(), k
        _, eh = orc.ingest(frames[k], cw, ch)
        assert np.array_equal(out["exposure"][k], eh), k
        f = out["flows"][k]
        want = orc.derotate(float(f["flow_x"]), float(f["flow_y"]), float(gyro[k + 1, 0]), float(gyro[k + 1, 1]),
                            float(gyro[k + 1, 3]), FX, FY, 4.5, 0.01)
        assert np.asarray(want, np.float32).tobytes() == out["derotated"][k].tobytes(), k
    eng.close()


def test_sequence_pipeline_replays_from_a_graph_and_handles_short_and_stalled_sequences(aof, orc, synth, gpu_device):
    """The call only enqueues: captured into a hipGraph and replayed on new frame contents it gives the new
    sequence's records.  One frame gives one (zero) record, two frames at most two; time stamps that never
    advance raise AOF_SEQ_STATUS_STALLED and publish nothing behind frame 0."""
    import torch
    cw = ch = 64
    p = aof.px4flow_params(cw, ch)
    eng = aof.FlowEngine(p, 0)
    sp = aof.sequence_params(160, 120, cw, ch, FX, FY, 15, 7_000_000, 1, 100, 0)
    n = 48
    times = (np.arange(n) * 13333).astype(np.int64)
    fa, ga = make_inputs(synth, 160, 120, n, 21, times)
    fb, gb = make_inputs(synth, 160, 120, n, 22, times)
    cam = torch.from_numpy(fa).to(gpu_device)
    tt = torch.from_numpy(times).to(gpu_device)
    gy = torch.from_numpy(ga).to(gpu_device)
    ws, L = eng.sequence(sp, cam, tt, gy)
    torch.cuda.synchronize()
    first = eng.sequence_outputs(sp, ws, L, n)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.sequence(sp, cam, tt, gy, workspace=ws)
    results = []
    for frames, gyro in ((fb, gb), (fa, ga)):
        cam.copy_(torch.from_numpy(frames))
        gy.copy_(torch.from_numpy(gyro))
        ws.zero_()
        g.replay()
        torch.cuda.synchronize()
        results.append(eng.sequence_outputs(sp, ws, L, n))
    assert results[1]["mavlink"] == first["mavlink"] and results[1]["records"].tobytes() == first["records"].tobytes()
    assert results[0]["mavlink"] != first["mavlink"]
    fac = aof.OpticalFlowPX4(FX, FY, 15, cw, ch)
    _, wire = replay(lambda img, t: fac.calcFlow(img, t), crop_of(fb, cw, ch), times, gb, 7_000_000, 0, aof.pack_optical_flow_rad)
    fac.close()
    assert results[0]["mavlink"] == wire

    for short in (1, 2):
        ws2, L2 = eng.sequence(sp, torch.from_numpy(fa[:short]).to(gpu_device), tt[:short].clone(), gy[:short].clone())
        torch.cuda.synchronize()
        o = eng.sequence_outputs(sp, ws2, L2, short)
        assert len(o["records"]) == 1 and o["records"][0]["quality"] == 0 and o["records"][0]["dt_us"] == 0   # 13.3 ms < 1/15 s
        assert o["mavlink"][0] == py_frame(7_000_000, 0, 0, 0.0, 0.0, tuple(float(v) for v in ga[0, :3]), 0, 0)

    stalled = 6000
    cam_s = torch.zeros((stalled, 64, 64), dtype=torch.uint8, device=gpu_device)
    sp_s = aof.sequence_params(64, 64, cw, ch, FX, FY, 15, 7_000_000, 1, 100, 0)
    ws3, L3 = eng.sequence(sp_s, cam_s, torch.full((stalled,), 5, dtype=torch.int64, device=gpu_device))
    torch.cuda.synchronize()
    o = eng.sequence_outputs(sp_s, ws3, L3, stalled)
    assert o["status"] & aof.SEQ_STATUS_STALLED and len(o["records"]) == 1
    eng.close()


def test_sequence_pipeline_scales_to_a_long_recording(aof, orc, synth, gpu_device):
    """20 000 frames (4.4 minutes at 75 fps) in one call: every record equals the C++ facade's, and the
    limiter's chain (pointer doubling over 15 rounds) numbers the messages without a gap."""
    import torch
    cw = ch = 64
    n = 20000
    base, _ = synth.make_sequence(cw, ch, 64, 4, seed=31, max_step=3)
    frames = base[np.arange(n) % 64]
    frames[5000:5040] = 0                                     # a stretch without texture: quality 0 frames inside segments
    rng = np.random.default_rng(31)
    times = np.cumsum(np.concatenate([[0], rng.integers(12000, 15000, n - 1)])).astype(np.int64)
    gyro = np.zeros((n, 4), np.float32)
    gyro[:, :3] = rng.normal(0, 0.003, (n, 3)).astype(np.float32)
    p = aof.px4flow_params(cw, ch)
    eng = aof.FlowEngine(p, 0)
    sp = aof.sequence_params(cw, ch, cw, ch, FX, FY, 15, 9_000_000, 1, 100, 17)
    ws, L = eng.sequence(sp, torch.from_numpy(frames).to(gpu_device), torch.from_numpy(times).to(gpu_device),
                         torch.from_numpy(gyro).to(gpu_device))
    torch.cuda.synchronize()
    out = eng.sequence_outputs(sp, ws, L, n)
    fac = aof.OpticalFlowPX4(FX, FY, 15, cw, ch)
    recs, wire = replay(lambda img, t: fac.calcFlow(img, t), frames, times, gyro, 9_000_000, 17, aof.pack_optical_flow_rad)
    fac.close()
    assert out["status"] == 0 and len(out["records"]) == len(recs) > 3000
    assert out["mavlink"] == wire
    assert np.all(np.diff(out["records"]["frame"].astype(np.int64)) > 0)
    eng.close()


def test_sequence_entry_point_argument_handling(aof, gpu_device):
    """aof_sequence_device straight through the C ABI: what it refuses (and with which code), and that refusing
    leaves the context usable."""
    import ctypes as C
    import torch
    p = aof.px4flow_params(64, 64)
    eng = aof.FlowEngine(p, 0)
    sp = aof.sequence_params(160, 120, 64, 64, FX, FY, 15, 1_000_000, 1, 100, 0)
    n = 12
    L = aof.sequence_layout(p, sp, n)
    cam = torch.zeros((n, 120, 160), dtype=torch.uint8, device=gpu_device)
    tt = torch.arange(n, dtype=torch.int64, device=gpu_device) * 13333
    gy = torch.zeros((n, 4), dtype=torch.float32, device=gpu_device)
    ws = torch.zeros(L.total_bytes + 256, dtype=torch.uint8, device=gpu_device)
    stream = torch.cuda.current_stream().cuda_stream
    call = aof.lib.aof_sequence_device
    args = lambda **kw: [kw.get("ctx", eng._ctx), C.byref(kw.get("sp", sp)), kw.get("cam", cam.data_ptr()), 160 * 120, kw.get("n", n),
                         kw.get("tt", tt.data_ptr()), kw.get("gy", gy.data_ptr()), kw.get("ws", ws.data_ptr()),
                         kw.get("bytes", L.total_bytes), stream]
    assert call(*args()) == 0
    assert call(*args(ctx=None)) == -22
    assert call(*args(cam=None)) == -22 and call(*args(tt=None)) == -22 and call(*args(ws=None)) == -22
    assert call(*args(ws=ws.data_ptr() + 16)) == -22                     # 256-byte alignment
    assert call(*args(bytes=L.total_bytes - 1)) == -28                   # -ENOSPC
    assert call(*args(n=-1)) == -22
    assert call(*args(n=0)) == 0                                         # nothing to do
    assert call(*args(sp=aof.sequence_params(160, 120, 32, 32, FX, FY, 15, 1, 1, 100, 0))) == -22   # crop != the context's frame
    assert call(*args(sp=aof.sequence_params(160, 120, 64, 64, 0.0, FY, 15, 1, 1, 100, 0))) == -22  # focal length must be positive
    derot = aof.sequence_params(160, 120, 64, 64, FX, FY, 15, 1, 1, 100, 0, derotate=(4.5, 0.01))
    Ld = aof.sequence_layout(p, derot, n)
    wsd = torch.zeros(Ld.total_bytes, dtype=torch.uint8, device=gpu_device)
    assert call(*args(sp=derot, gy=None, ws=wsd.data_ptr(), bytes=Ld.total_bytes)) == -22          # de-rotation needs the gyro
    assert call(*args(sp=derot, ws=wsd.data_ptr(), bytes=Ld.total_bytes)) == 0
    assert call(*args(gy=None)) == 0                                     # no gyro: zeros in the messages
    torch.cuda.synchronize()
    out = eng.sequence_outputs(sp, ws[:L.total_bytes], L, n)
    assert len(out["records"]) >= 1 and out["records"][0]["frame"] == 0 and out["status"] == 0
    eng.close()
