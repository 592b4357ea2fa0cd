"""Gyro de-rotation of flow records: the published PX4Flow
compensation, host oracle vs the device kernel, bit-exact floats."""
import numpy as np
import pytest


def test_oracle_derotate_hand(orc):
    f = np.float32
    # below the rate threshold: untouched
    assert orc.derotate(1.5, -2.0, 0.0001, 0.0001, 0.0133, 216.0, 216.0, 4.5, 0.01) == (1.5, -2.0)
    # y gyro adds to x flow, x gyro subtracts from y flow
    x, y = orc.derotate(1.5, -2.0, 0.004, 0.002, 0.0133, 200.0, 100.0, 4.5, 0.01)
    assert f(x) == f(f(1.5) + f(f(0.002) * f(200.0))) and f(y) == f(f(-2.0) - f(f(0.004) * f(100.0)))
    # clamped to the measurable range
    assert orc.derotate(4.0, -4.0, 0.01, 0.01, 0.0133, 216.0, 216.0, 4.5, 0.01) == (4.5, -4.5)


@pytest.mark.gpu
def test_gpu_derotate_parity(aof, orc, synth, gpu_device):
    import torch
    n = 4096
    rng = np.random.default_rng(12)
    flows = np.zeros(n, aof.FLOW_DTYPE)
    flows["flow_x"] = rng.uniform(-4.5, 4.5, n).astype(np.float32)
    flows["flow_y"] = rng.uniform(-4.5, 4.5, n).astype(np.float32)
    gyro = np.zeros(n, aof.GYRO_DTYPE)
    gyro["integ_x"] = rng.normal(0, 0.004, n).astype(np.float32)
    gyro["integ_y"] = rng.normal(0, 0.004, n).astype(np.float32)
    gyro["dt_s"] = rng.uniform(0.01, 0.02, n).astype(np.float32)
    gyro["integ_x"][:50] = 0
    tf = torch.from_numpy(flows.view(np.uint8).reshape(n, 16)).to(gpu_device)
    tg = torch.from_numpy(gyro.view(np.float32).reshape(n, 4)).to(gpu_device)
    out = aof.derotate_batch(tf, tg, 216.6677, 216.2457, 4.5, 0.05).cpu().numpy()
    clamped = untouched = 0
    for i in range(n):
        ex, ey = orc.derotate(float(flows["flow_x"][i]), float(flows["flow_y"][i]), float(gyro["integ_x"][i]),
                              float(gyro["integ_y"][i]), float(gyro["dt_s"][i]), 216.6677, 216.2457, 4.5, 0.05)
        assert np.float32(ex).tobytes() == out[i, 0].tobytes() and np.float32(ey).tobytes() == out[i, 1].tobytes(), i
        clamped += abs(ex) == 4.5
        untouched += ex == flows["flow_x"][i]
    assert clamped > 0 and untouched > 0, "both branches exercised"
