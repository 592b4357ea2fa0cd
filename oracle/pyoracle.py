"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: the oracle restates the published PX4Flow algorithm plus the
build-defined extensions of DESIGN.md "Spec"; the reference's own engine source
and tests are absent (see oracle/aof_oracle.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

BLOCK_DTYPE = np.dtype([("dx", "i1"), ("dy", "i1"), ("sad", "<u2")])
FLOW_DTYPE = np.dtype([("flow_x", "<f4"), ("flow_y", "<f4"), ("count", "<u4"), ("quality", "u1"),
                       ("flags", "u1"), ("pred_x", "i1"), ("pred_y", "i1")])
PARAM_FIELDS = ("width", "height", "tile", "search", "grid_mode", "num_blocks",
                "feature_threshold", "value_threshold", "subpixel", "hist_filter",
                "pyramid_levels", "mean_subtract", "min_valid")


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in PARAM_FIELDS]


class Grid(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("x0", "y0", "step_x", "step_y", "nx", "ny")]


class Px4State(C.Structure):
    _fields_ = [("params", Params), ("focal_x", C.c_float), ("focal_y", C.c_float),
                ("output_rate", C.c_int), ("initialized", C.c_int), ("img_old", C.c_void_p),
                ("time_last_pub", C.c_uint32), ("sum_flow_x", C.c_float), ("sum_flow_y", C.c_float),
                ("sum_flow_quality", C.c_int), ("valid_frame_count", C.c_int)]


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    VP, P = C.c_void_p, C.POINTER
    lib.orc_params_default.argtypes = [P(Params), C.c_int, C.c_int]
    lib.orc_params_check.argtypes = [P(Params)]
    lib.orc_grid_for_level.argtypes = [P(Params), C.c_int, P(Grid)]
    lib.orc_hist_size.argtypes = [P(Params), C.c_int]
    lib.orc_compute_diff.restype = C.c_uint32
    lib.orc_compute_diff.argtypes = [VP, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_set_fast_sad.argtypes = [C.c_int]
    lib.orc_fast_sad_available.restype = C.c_int
    lib.orc_sad.restype = C.c_uint32
    lib.orc_sad.argtypes = [VP, C.c_int, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_subpixel.argtypes = [VP, C.c_int, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]
    lib.orc_frame_mean.restype = C.c_uint32
    lib.orc_frame_mean.argtypes = [VP, C.c_int64]
    lib.orc_pyramid_down.argtypes = [VP, C.c_int, C.c_int, VP]
    lib.orc_equalise.argtypes = [VP, C.c_int64, C.c_int, VP]
    lib.orc_reduce.argtypes = [P(Params), VP, VP, C.c_int, C.c_int, VP, P(C.c_int32), P(C.c_int32)]
    lib.orc_flow_pair.argtypes = [P(Params), VP, VP, VP, VP, VP, VP, VP]
    lib.orc_flow_batch.argtypes = [P(Params), VP, VP, C.c_int64, C.c_int64, VP, VP, C.c_int]
    lib.orc_crop_rect.argtypes = [C.c_int] * 4 + [P(C.c_int), P(C.c_int)]
    lib.orc_exposure_bin.argtypes = [C.c_int]
    lib.orc_ingest.argtypes = [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]
    lib.orc_exposure_msv.restype = C.c_float
    lib.orc_exposure_msv.argtypes = [VP]
    lib.orc_derotate.argtypes = [C.c_float] * 9 + [P(C.c_float), P(C.c_float)]
    lib.orc_angle.restype = C.c_float
    lib.orc_angle.argtypes = [C.c_float, C.c_float]
    lib.orc_px4_init.argtypes = [P(Px4State), P(Params), C.c_float, C.c_float, C.c_int]
    lib.orc_px4_free.argtypes = [P(Px4State)]
    lib.orc_px4_calc_flow.argtypes = [P(Px4State), VP, C.c_uint32, P(C.c_int), P(C.c_float),
                                      P(C.c_float)]
    return lib


lib = _load()


def params_from(obj) -> Params:
    """Copy the 13 int fields from any object/dict with the same names (e.g. the
    product's aof Params) -- the oracle keeps its own struct definition."""
    p = Params()
    for n in PARAM_FIELDS:
        setattr(p, n, int(obj[n] if isinstance(obj, dict) else getattr(obj, n)))
    return p


def default_params(width, height, **kw) -> Params:
    p = Params()
    lib.orc_params_default(C.byref(p), width, height)
    for k, v in kw.items():
        if k not in PARAM_FIELDS:
            raise AttributeError(k)
        setattr(p, k, int(v))
    return p


def px4flow_params(width, height, search=4, feature_threshold=30, value_threshold=3000, **kw):
    return default_params(width, height, search=search, grid_mode=1, num_blocks=5,
                          feature_threshold=feature_threshold, value_threshold=value_threshold,
                          subpixel=1, **kw)


def grid(p: Params, level=0) -> Grid:
    g = Grid()
    rc = lib.orc_grid_for_level(C.byref(p), level, C.byref(g))
    if rc:
        raise ValueError("bad grid")
    return g


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def flow_pair(p: Params, prev, cur, want_l1=False):
    """Returns dict(blocks, subdirs, flow[, blocks_l1, subdirs_l1])."""
    prev, cur = _u8(prev), _u8(cur)
    assert prev.shape == cur.shape == (p.height, p.width)
    g0 = grid(p, 0)
    nb0 = g0.nx * g0.ny
    blocks = np.zeros(nb0, dtype=BLOCK_DTYPE)
    subdirs = np.zeros(nb0, dtype=np.uint8)
    flow = np.zeros(1, dtype=FLOW_DTYPE)
    out = {}
    b1 = s1 = None
    if p.pyramid_levels == 2:
        g1 = grid(p, 1)
        b1 = np.zeros(g1.nx * g1.ny, dtype=BLOCK_DTYPE)
        s1 = np.full(g1.nx * g1.ny, 8, dtype=np.uint8)
    rc = lib.orc_flow_pair(C.byref(p), prev.ctypes.data, cur.ctypes.data, blocks.ctypes.data,
                           subdirs.ctypes.data, b1.ctypes.data if b1 is not None else None,
                           s1.ctypes.data if s1 is not None else None, flow.ctypes.data)
    if rc:
        raise ValueError(f"orc_flow_pair rc={rc}")
    out.update(blocks=blocks, subdirs=subdirs, flow=flow[0])
    if want_l1 and b1 is not None:
        out.update(blocks_l1=b1, subdirs_l1=s1)
    return out


def flow_batch(p: Params, prevs, curs, threads=0):
    prevs, curs = _u8(prevs), _u8(curs)
    n = prevs.shape[0]
    g0 = grid(p, 0)
    blocks = np.zeros((n, g0.nx * g0.ny), dtype=BLOCK_DTYPE)
    flows = np.zeros(n, dtype=FLOW_DTYPE)
    used = lib.orc_flow_batch(C.byref(p), prevs.ctypes.data, curs.ctypes.data,
                              p.width * p.height, n, blocks.ctypes.data, flows.ctypes.data, threads)
    if used < 0:
        raise ValueError("orc_flow_batch failed")
    return blocks, flows, used


def set_fast_sad(on):
    """bench.py's cpu_baseline leg only: SAD through the host's SAD instruction (SSE2 psadbw)."""
    lib.orc_set_fast_sad(int(bool(on)))


def angle(flow_px, focal_px):
    """Pixel flow -> angular flow (rad): the oracle's restatement of the spec's fixed-operation atan2."""
    return float(lib.orc_angle(flow_px, focal_px))


def fast_sad_available():
    return bool(lib.orc_fast_sad_available())


def compute_diff(img, x, y, tile=8):
    img = _u8(img)
    return int(lib.orc_compute_diff(img.ctypes.data, x, y, img.shape[1], tile))


def sad(a, ax, ay, b, bx, by, tile=8):
    a, b = _u8(a), _u8(b)
    return int(lib.orc_sad(a.ctypes.data, ax, ay, b.ctypes.data, bx, by, a.shape[1], tile))


def subpixel(a, ax, ay, b, bx, by, tile=8):
    a, b = _u8(a), _u8(b)
    acc = np.zeros(8, dtype=np.uint32)
    lib.orc_subpixel(a.ctypes.data, ax, ay, b.ctypes.data, bx, by, a.shape[1], tile, acc.ctypes.data)
    return acc


def frame_mean(img):
    img = _u8(img)
    return int(lib.orc_frame_mean(img.ctypes.data, img.size))


def pyramid_down(img):
    img = _u8(img)
    h, w = img.shape
    out = np.zeros((h // 2, w // 2), dtype=np.uint8)
    lib.orc_pyramid_down(img.ctypes.data, w, h, out.ctypes.data)
    return out


def equalise(img, delta):
    img = _u8(img)
    out = np.zeros_like(img)
    lib.orc_equalise(img.ctypes.data, img.size, int(delta), out.ctypes.data)
    return out


def reduce(p: Params, blocks, subdirs, rng):
    blocks = np.ascontiguousarray(blocks, dtype=BLOCK_DTYPE)
    flow = np.zeros(1, dtype=FLOW_DTYPE)
    px, py = C.c_int32(), C.c_int32()
    sd = None if subdirs is None else _u8(subdirs)
    lib.orc_reduce(C.byref(p), blocks.ctypes.data, sd.ctypes.data if sd is not None else None,
                   blocks.size, rng, flow.ctypes.data, C.byref(px), C.byref(py))
    return flow[0], px.value, py.value


def ingest(cam, crop_w, crop_h):
    """(cropped frame, 10-bin exposure histogram) of one camera frame."""
    cam = _u8(cam)
    h, w = cam.shape
    crop = np.zeros((crop_h, crop_w), dtype=np.uint8)
    hist = np.zeros(10, dtype=np.uint32)
    rc = lib.orc_ingest(cam.ctypes.data, w, h, crop_w, crop_h, crop.ctypes.data, hist.ctypes.data)
    if rc:
        raise ValueError(rc)
    return crop, hist


def exposure_msv(hist):
    hist = np.ascontiguousarray(hist, dtype=np.uint32)
    return float(lib.orc_exposure_msv(hist.ctypes.data))


def exposure_bin(v):
    return int(lib.orc_exposure_bin(int(v)))


def derotate(flow_x, flow_y, gx, gy, dt_s, focal_x, focal_y, max_flow, rate_threshold):
    ox, oy = C.c_float(), C.c_float()
    lib.orc_derotate(flow_x, flow_y, gx, gy, dt_s, focal_x, focal_y, max_flow, rate_threshold,
                     C.byref(ox), C.byref(oy))
    return ox.value, oy.value


class Px4:
    """Oracle of the facade semantics (calcFlow)."""

    def __init__(self, p: Params, fx, fy, output_rate):
        self.s = Px4State()
        rc = lib.orc_px4_init(C.byref(self.s), C.byref(p), fx, fy, output_rate)
        if rc:
            raise ValueError(rc)
        self.p = p

    def calc_flow(self, img, t_us):
        img = _u8(img)
        dt, fx, fy = C.c_int(0), C.c_float(0), C.c_float(0)
        q = lib.orc_px4_calc_flow(C.byref(self.s), img.ctypes.data, t_us & 0xFFFFFFFF,
                                  C.byref(dt), C.byref(fx), C.byref(fy))
        return q, dt.value, fx.value, fy.value

    def __del__(self):
        lib.orc_px4_free(C.byref(self.s))
