"""OPTICAL_FLOW_RAD producer: field mapping of
/root/reference/src/mainloop.cpp:359-371 and the MAVLink 2 frame of
src/mavlink_tcp.cpp:142-162.  modules/mavlink_c is absent from the reference mount, so
the wire format is checked against an independent restatement of the public MAVLink 2
serialization rules written here (struct packing + CRC-16/MCRF4XX), not against
reference bytes."""
import os
import struct

import numpy as np
import pytest


def x25(data, crc=0xFFFF):
    for b in data:
        tmp = (b ^ (crc & 0xFF)) & 0xFF
        tmp = (tmp ^ (tmp << 4)) & 0xFF
        crc = ((crc >> 8) ^ (tmp << 8) ^ (tmp << 3) ^ (tmp >> 4)) & 0xFFFF
    return crc


def py_frame(offset_ts, img_time_us, dt_us, fx, fy, gyro, quality, seq):
    payload = struct.pack("<QIfffffIfhBB", offset_ts + img_time_us, dt_us & 0xFFFFFFFF, fx, fy,
                          np.float32(-gyro[1]), np.float32(gyro[0]), np.float32(gyro[2]), 0, -1.0, 0, 0,
                          quality & 0xFF)
    assert len(payload) == 44
    while len(payload) > 1 and payload[-1] == 0:
        payload = payload[:-1]
    hdr = bytes([len(payload), 0, 0, seq, 1, 100, 106, 0, 0])
    crc = x25(bytes([138]), x25(hdr + payload))
    return b"\xfd" + hdr + payload + struct.pack("<H", crc)


@pytest.fixture(scope="module")
def facade(aof):
    if not os.path.exists(aof.FACADE_PATH):
        import __graft_entry__ as ge
        ge.build()
    return aof


def test_crc_check_value(facade):
    data = b"123456789"
    assert x25(data) == 0x6F91, "CRC-16/MCRF4XX check value"
    buf = np.frombuffer(data, np.uint8).copy()
    assert facade.facade_lib().aof_facade_mavlink_crc(buf.ctypes.data, 9) == 0x6F91


def test_frames_match_independent_serializer(facade):
    rng = np.random.default_rng(4)
    for k in range(200):
        q = int(rng.integers(1, 256)) if k % 5 else 0       # quality 0 exercises payload truncation
        args = (int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 2 ** 32)), int(rng.integers(0, 200000)),
                float(np.float32(rng.normal(0, 0.02))), float(np.float32(rng.normal(0, 0.02))),
                tuple(float(v) for v in rng.normal(0, 0.01, 3)), q, k & 0xFF)
        got = facade.pack_optical_flow_rad(args[0], args[1], args[2], args[3], args[4], args[5], args[6], args[7])
        assert got == py_frame(*args), k
        assert got[0] == 0xFD and got[7:10] == bytes([106, 0, 0]) and got[5:7] == bytes([1, 100])
        assert len(got) == 12 + got[1] and got[1] == (44 if q else 40)


def test_field_mapping_of_the_reference(facade):
    """mainloop.cpp:364-365 swaps the gyro axes; :367-370 are constants."""
    f = facade.pack_optical_flow_rad(1000, 234, 66665, 0.25, -0.5, gyro=(0.1, 0.2, 0.3), quality=200, seq=7)
    p = f[10:10 + f[1]].ljust(44, b"\0")
    t, integ, x, y, gx, gy, gz, tdd, dist, temp, sid, q = struct.unpack("<QIfffffIfhBB", p)
    assert (t, integ, x, y) == (1234, 66665, 0.25, -0.5)
    assert gx == np.float32(-0.2) and gy == np.float32(0.1) and gz == np.float32(0.3)
    assert (tdd, dist, temp, sid, q) == (0, -1.0, 0, 0, 200)
