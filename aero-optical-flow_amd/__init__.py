"""Python harness over the C ABI of the MI355X flow engine (include/aof.h).

The product is ``csrc/libaof.so`` (hand-written gfx950 kernels behind a plain C
ABI) and the C++ facade in ``facade/``; this module only binds the C ABI with
``ctypes`` so that tests and ``bench.py`` can drive it with device memory owned
by PyTorch.  It mirrors the reference's operator surface for the path --
``OpticalFlowPX4.calcFlow`` as called from
``/root/reference/src/mainloop.cpp:322`` -- in :class:`OpticalFlowPX4`.

There is no CPU fallback anywhere in this package: if ``libaof.so`` is missing
the import raises, and every compute entry point needs a gfx950 device.

The directory name contains a hyphen, so import it through
``__graft_entry__.load_package()`` (registers it as ``aero_optical_flow_amd``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# AOF_LIB: load another build of the same ABI (A/B timing of two kernel builds on one box)
LIB_PATH = os.environ.get("AOF_LIB") or os.path.join(_HERE, "csrc", "libaof.so")

GRID_DENSE, GRID_PX4FLOW = 0, 1
SEARCH_EXHAUSTIVE, SEARCH_PRUNED, SEARCH_ADAPTIVE = 0, 1, 2
SAD_SKIPPED = 0xFFFF
FLAG_FLOW_VALID, FLAG_PRED_VALID = 1, 2
K_PYRAMID, K_SEARCH_L1, K_REDUCE_L1, K_SEARCH, K_REDUCE = range(5)

BLOCK_DTYPE = np.dtype([("dx", "i1"), ("dy", "i1"), ("sad", "<u2")])
FLOW_DTYPE = np.dtype([("flow_x", "<f4"), ("flow_y", "<f4"), ("count", "<u4"), ("quality", "u1"),
                       ("flags", "u1"), ("pred_x", "i1"), ("pred_y", "i1")])
assert BLOCK_DTYPE.itemsize == 4 and FLOW_DTYPE.itemsize == 16


class Params(C.Structure):
    """``aof_params`` (include/aof.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "tile", "search", "grid_mode", "num_blocks", "feature_threshold",
        "value_threshold", "subpixel", "hist_filter", "pyramid_levels", "mean_subtract",
        "min_valid")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class IngestParams(C.Structure):
    """``aof_ingest_params`` (include/aof.h)."""
    _fields_ = [(n, C.c_int32) for n in ("camera_width", "camera_height", "crop_width", "crop_height")]


class DerotateParams(C.Structure):
    """``aof_derotate_params`` (include/aof.h)."""
    _fields_ = [(n, C.c_float) for n in ("focal_x", "focal_y", "max_flow", "rate_threshold")]


GYRO_DTYPE = np.dtype([("integ_x", "<f4"), ("integ_y", "<f4"), ("integ_z", "<f4"), ("dt_s", "<f4")])


class SequenceParams(C.Structure):
    """``aof_sequence_params`` (include/aof.h)."""
    _fields_ = [("ingest", IngestParams), ("focal_x", C.c_float), ("focal_y", C.c_float),
                ("output_rate", C.c_int32), ("offset_timestamp_usec", C.c_uint64),
                ("system_id", C.c_uint8), ("component_id", C.c_uint8), ("first_seq", C.c_uint8),
                ("derotate", C.c_uint8), ("derotate_params", DerotateParams)]


class SeqLayout(C.Structure):
    """``aof_seq_layout`` (include/aof.h)."""
    _fields_ = [(n, C.c_size_t) for n in ("total_bytes", "cropped", "exposure", "flows", "derotated", "count",
                                          "records", "frames", "frame_len", "scratch")]


SEQ_RECORD_DTYPE = np.dtype([("frame", "<u4"), ("quality", "<i4"), ("dt_us", "<i4"), ("flow_x", "<f4"),
                             ("flow_y", "<f4"), ("gyro_x", "<f4"), ("gyro_y", "<f4"), ("gyro_z", "<f4")])
SEQ_FRAME_BYTES = 56
SEQ_STATUS_STALLED = 1
assert SEQ_RECORD_DTYPE.itemsize == 32


class WsLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "total_bytes", "sums", "l1_prev", "l1_cur", "l1_blocks", "l1_subdirs", "l1_flows",
        "l0_blocks", "l0_subdirs", "l0_hist", "l1_hist", "hints")]


class StreamStats(C.Structure):
    """``aof_stream_stats`` (include/aof.h)."""
    _fields_ = [("calls", C.c_uint64), ("resident_served", C.c_uint64), ("resident_launches", C.c_uint32),
                ("resident_fallbacks", C.c_uint32), ("resident_lost", C.c_uint32), ("tagged_slow", C.c_uint32),
                ("launch_call_us_max", C.c_float), ("start_latency_us_max", C.c_float),
                ("last_report", C.c_char * 320)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["last_report"] = d["last_report"].decode(errors="replace")
        return d


class SearchStats(C.Structure):
    """``aof_search_stats`` (include/aof.h): what the ADAPTIVE search mode of an 8x8 context has done so far."""
    _fields_ = [("pruned_launches", C.c_uint64), ("exhaustive_launches", C.c_uint64), ("reports_read", C.c_uint64),
                ("belief", C.c_int32), ("paying_pct", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class AofError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"aof error {code}: {text}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:  # share torch's HIP runtime instance when torch is in the process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the facade
        pass
    lib = C.CDLL(LIB_PATH)
    P, VP, I64 = C.POINTER, C.c_void_p, C.c_int64
    sig = {
        "aof_version": (C.c_int, []),
        "aof_strerror": (C.c_char_p, [C.c_int]),
        "aof_params_default": (C.c_int, [P(Params), C.c_int, C.c_int]),
        "aof_params_px4flow": (C.c_int, [P(Params), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "aof_params_check": (C.c_int, [P(Params)]),
        "aof_grid": (C.c_int, [P(Params), C.c_int] + [P(C.c_int32)] * 6),
        "aof_workspace_layout": (C.c_int, [P(Params), I64, P(WsLayout)]),
        "aof_create": (C.c_int, [P(Params), C.c_int, P(VP)]),
        "aof_destroy": (None, [VP]),
        "aof_last_error": (C.c_char_p, [VP]),
        "aof_get_params": (C.c_int, [VP, P(Params)]),
        "aof_search_variant": (C.c_char_p, [VP]),
        "aof_set_force_generic": (C.c_int, [VP, C.c_int]),
        "aof_set_search_mode": (C.c_int, [VP, C.c_int]),
        "aof_get_search_mode": (C.c_int, [VP]),
        "aof_get_search_stats": (C.c_int, [VP, VP]),
        "aof_set_search_belief": (C.c_int, [VP, C.c_int]),
        "aof_set_split_coarse": (C.c_int, [VP, C.c_int]),
        "aof_set_reduce_fusion": (C.c_int, [VP, C.c_int]),
        "aof_flow_batch_device": (C.c_int, [VP, VP, VP, I64, I64, VP, VP, VP, VP, C.c_size_t, VP]),
        "aof_flow_pair_host": (C.c_int, [VP, VP, VP, VP, VP, VP]),
        "aof_stream_push_host": (C.c_int, [VP, VP, VP]),
        "aof_stream_reset": (C.c_int, [VP]),
        "aof_set_stream_graph": (C.c_int, [VP, C.c_int]),
        "aof_set_stream_resident": (C.c_int, [VP, C.c_int]),
        "aof_stream_get_stats": (C.c_int, [VP, P(StreamStats)]),
        "aof_debug_resident_fault": (C.c_int, [VP, C.c_int, C.c_uint32]),
        "aof_set_vote_deadline_us": (C.c_int, [VP, C.c_uint32]),
        "aof_debug_vote_deadline_ticks": (C.c_int, [VP, C.c_uint32]),
        "aof_ingest_batch_device": (C.c_int, [P(IngestParams), VP, I64, I64, VP, I64, VP, VP]),
        "aof_sequence_layout": (C.c_int, [P(Params), P(SequenceParams), I64, P(SeqLayout)]),
        "aof_sequence_device": (C.c_int, [VP, P(SequenceParams), VP, I64, I64, VP, VP, VP, C.c_size_t, VP]),
        "aof_flow_angle": (C.c_float, [C.c_float, C.c_float]),
        "aof_derotate_batch_device": (C.c_int, [P(DerotateParams), VP, VP, I64, VP, VP]),
        "aof_exposure_msv": (C.c_float, [VP]),
        "aof_exposure_bin": (C.c_int, [C.c_int]),
        "aof_set_profiling": (C.c_int, [VP, C.c_int]),
        "aof_set_profiling_mask": (C.c_int, [VP, C.c_uint32]),
        "aof_kernel_ms": (C.c_int, [VP, C.c_int, P(C.c_float)]),
        "aof_profile_count": (C.c_int, [VP, C.c_int]),
        "aof_profile_ms": (C.c_int, [VP, C.c_int, C.c_int, P(C.c_float)]),
    }
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if os.environ.get("AOF_LIB"):   # A/B timing against an older build of the ABI
                continue
            raise
        fn.restype, fn.argtypes = res, args
    return lib, tuple(sig)


lib, EXPORTS = _load()


def default_params(width, height, **overrides) -> Params:
    p = Params()
    lib.aof_params_default(C.byref(p), width, height)
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, int(v))
    return p


def px4flow_params(width, height, search=4, feature_threshold=30, value_threshold=3000,
                   **overrides) -> Params:
    p = Params()
    lib.aof_params_px4flow(C.byref(p), width, height, search, feature_threshold, value_threshold)
    for k, v in overrides.items():
        setattr(p, k, int(v))
    return p


def check_params(p: Params) -> int:
    return lib.aof_params_check(C.byref(p))


def grid(p: Params, level=0):
    """(x0, y0, step_x, step_y, nx, ny) of a pyramid level."""
    v = [C.c_int32() for _ in range(6)]
    rc = lib.aof_grid(C.byref(p), level, *[C.byref(x) for x in v])
    if rc:
        raise AofError(rc, lib.aof_strerror(rc).decode())
    return tuple(x.value for x in v)


def workspace_layout(p: Params, n_pairs: int) -> WsLayout:
    L = WsLayout()
    rc = lib.aof_workspace_layout(C.byref(p), n_pairs, C.byref(L))
    if rc:
        raise AofError(rc, lib.aof_strerror(rc).decode())
    return L


def algorithmic_bytes(p: Params) -> int:
    """Compulsory HBM bytes per frame pair:
    read both frames once, write 4 B per block and the 16 B result."""
    _, _, _, _, nx, ny = grid(p, 0)
    return 2 * p.width * p.height + 4 * nx * ny + 16


def abs_diffs(p: Params) -> int:
    _, _, _, _, nx, ny = grid(p, 0)
    return nx * ny * (2 * p.search + 1) ** 2 * p.tile ** 2


def ingest_batch(camera, crop_w, crop_h, cropped=None, hist=None, want_hist=True):
    """Frame ingest on the device (centre crop + 10-bin exposure histogram; the caller-side
    steps of /root/reference/src/mainloop.cpp:295-298,203-214).  camera: uint8 CUDA tensor
    [n, cam_h, cam_w].  Returns (cropped [n, crop_h, crop_w], hist [n, 10] int32-as-uint32)."""
    import torch
    n, cam_h, cam_w = camera.shape
    p = IngestParams(cam_w, cam_h, crop_w, crop_h)
    if cropped is None:
        cropped = torch.empty((n, crop_h, crop_w), dtype=torch.uint8, device=camera.device)
    if hist is None and want_hist:
        hist = torch.empty((n, 10), dtype=torch.int32, device=camera.device)
    stream = torch.cuda.current_stream(camera.device).cuda_stream
    rc = lib.aof_ingest_batch_device(C.byref(p), camera.data_ptr(), camera.stride(0), n,
                                     cropped.data_ptr(), cropped.stride(0) if n else crop_w * crop_h,
                                     hist.data_ptr() if hist is not None else None, stream)
    if rc:
        raise AofError(rc, lib.aof_strerror(rc).decode())
    return cropped, hist


def derotate_batch(flows, gyro, focal_x, focal_y, max_flow, rate_threshold):
    """Published PX4Flow gyro compensation of a batch of flow records on the device.
    flows: uint8 CUDA tensor [n, 16] (aof_flow records); gyro: float32 CUDA tensor [n, 4]
    (integ_x, integ_y, integ_z, dt_s).  Returns float32 [n, 2]."""
    import torch
    n = flows.shape[0]
    out = torch.empty((n, 2), dtype=torch.float32, device=flows.device)
    p = DerotateParams(focal_x, focal_y, max_flow, rate_threshold)
    rc = lib.aof_derotate_batch_device(C.byref(p), flows.data_ptr(), gyro.data_ptr(), n, out.data_ptr(),
                                       torch.cuda.current_stream(flows.device).cuda_stream)
    if rc:
        raise AofError(rc, lib.aof_strerror(rc).decode())
    return out


def flow_angle(flow_px: float, focal_px: float) -> float:
    """Pixel flow -> angular flow (rad): the fixed-operation atan2 of include/aof_math.h."""
    return float(lib.aof_flow_angle(flow_px, focal_px))


def sequence_params(cam_w, cam_h, crop_w, crop_h, focal_x=216.6677, focal_y=216.2457, output_rate=15,
                    offset_timestamp_usec=0, system_id=1, component_id=100, first_seq=0, derotate=None) -> SequenceParams:
    """``aof_sequence_params``; derotate: None or (max_flow, rate_threshold) of the gyro compensation."""
    sp = SequenceParams()
    sp.ingest = IngestParams(cam_w, cam_h, crop_w, crop_h)
    sp.focal_x, sp.focal_y, sp.output_rate = focal_x, focal_y, output_rate
    sp.offset_timestamp_usec = offset_timestamp_usec
    sp.system_id, sp.component_id, sp.first_seq = system_id, component_id, first_seq & 0xFF
    sp.derotate = 0 if derotate is None else 1
    if derotate is not None:
        sp.derotate_params = DerotateParams(focal_x, focal_y, derotate[0], derotate[1])
    return sp


def sequence_layout(p: Params, sp: SequenceParams, n_frames: int) -> SeqLayout:
    L = SeqLayout()
    rc = lib.aof_sequence_layout(C.byref(p), C.byref(sp), n_frames, C.byref(L))
    if rc:
        raise AofError(rc, lib.aof_strerror(rc).decode())
    return L


def exposure_msv(hist) -> float:
    h = np.ascontiguousarray(np.asarray(hist), dtype=np.uint32)
    assert h.shape == (10,)
    return float(lib.aof_exposure_msv(h.ctypes.data))


class FlowEngine:
    """One ``aof_ctx``: the batched, device-resident hot path."""

    def __init__(self, params: Params, device: int = 0):
        self._ctx = C.c_void_p()
        rc = lib.aof_create(C.byref(params), device, C.byref(self._ctx))
        if rc:
            self._ctx = None
            raise AofError(rc, lib.aof_strerror(rc).decode())
        self.params = params
        self.device = device
        self._ws = None

    def close(self):
        if getattr(self, "_ctx", None) and lib is not None:  # lib is None at interpreter shutdown
            lib.aof_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise AofError(rc, lib.aof_last_error(self._ctx).decode())
        return rc

    @property
    def variant(self) -> str:
        return lib.aof_search_variant(self._ctx).decode()

    def force_generic(self, on=True):
        self._check(lib.aof_set_force_generic(self._ctx, int(on)))

    def set_search_mode(self, mode):
        """SEARCH_ADAPTIVE (the default of every context: exact pruning where it pays -- 16x16 tiles by a probe per
        pair, 8x8 tiles by what the context's previous launches reported), SEARCH_EXHAUSTIVE (every candidate summed
        completely: the data-independent rate) or SEARCH_PRUNED (exact pruning always: rate depends on the images).
        All three write the same records."""
        self._check(lib.aof_set_search_mode(self._ctx, int(mode)))

    @property
    def search_mode(self) -> int:
        return lib.aof_get_search_mode(self._ctx)

    def search_stats(self) -> dict:
        """Launch counters and current verdict of the ADAPTIVE 8x8 search (``aof_search_stats``)."""
        st = SearchStats()
        self._check(lib.aof_get_search_stats(self._ctx, C.byref(st)))
        return st.as_dict()

    def set_search_belief(self, belief):
        """Tell an ADAPTIVE 8x8 context what to assume about its images before it has launched anything:
        1 pruning pays, 0 it does not, -1 forget (``aof_set_search_belief``)."""
        self._check(lib.aof_set_search_belief(self._ctx, int(belief)))

    def set_split_coarse(self, on=True):
        """Two-level batches: run K1 / level-1 search / level-1 reduce as separate kernels (fills the
        workspace's level-1 frames) instead of the fused coarse kernel."""
        self._check(lib.aof_set_split_coarse(self._ctx, int(on)))

    def set_reduce_fusion(self, on=True):
        """8x8 tiles on large grids: on=True reduces inside the search launch (opt-in); the default
        (on=False) launches K3 as a separate kernel behind the search."""
        if not hasattr(lib, "aof_set_reduce_fusion"):
            return   # (AOF_LIB pointing at an older build)
        self._check(lib.aof_set_reduce_fusion(self._ctx, int(on)))

    def set_profiling(self, on=True, kernels=None):
        """Time every kernel (kernels=None) or only the given kernel ids with HIP events."""
        if on and kernels is not None:
            mask = 0
            for k in kernels:
                mask |= 1 << k
            self._check(lib.aof_set_profiling_mask(self._ctx, mask))
        else:
            self._check(lib.aof_set_profiling(self._ctx, int(on)))

    def kernel_ms(self, kernel_id) -> float:
        ms = C.c_float()
        self._check(lib.aof_kernel_ms(self._ctx, kernel_id, C.byref(ms)))
        return ms.value

    def profile_ms(self, kernel_id):
        """Durations (ms) of the launches of one kernel timed since profiling was enabled
        (at most the last AOF_PROFILE_RING); synchronises on their events."""
        n = self._check(lib.aof_profile_count(self._ctx, kernel_id))
        out = []
        ms = C.c_float()
        for i in range(n):
            self._check(lib.aof_profile_ms(self._ctx, kernel_id, i, C.byref(ms)))
            out.append(ms.value)
        return out

    def grid(self, level=0):
        return grid(self.params, level)

    def nblocks(self, level=0):
        g = self.grid(level)
        return g[4] * g[5]

    # -- device-resident batch ------------------------------------------------
    def flow_batch(self, prev, cur, blocks=None, subdirs=None, flows=None, workspace=None,
                   pair_stride=None, n_pairs=None):
        """prev/cur: uint8 CUDA tensors [n, H, W] (or any layout described by
        pair_stride).  Returns (blocks [n, nb] int32-viewed records, flows [n, 16] bytes,
        workspace).  Everything is enqueued on torch's current stream."""
        import torch
        p = self.params
        if n_pairs is None:
            n_pairs = prev.shape[0]
        if pair_stride is None:
            pair_stride = prev.stride(0) if prev.dim() == 3 else p.width * p.height
        dev = prev.device
        nb = self.nblocks(0)
        if blocks is None:
            blocks = torch.empty((n_pairs, nb), dtype=torch.int32, device=dev)
        if flows is None:
            flows = torch.empty((n_pairs, 16), dtype=torch.uint8, device=dev)
        if subdirs is None and p.subpixel:
            subdirs = torch.empty((n_pairs, nb), dtype=torch.uint8, device=dev)
        L = workspace_layout(p, n_pairs)
        if workspace is None:
            if self._ws is None or self._ws.numel() < L.total_bytes or self._ws.device != dev:
                self._ws = torch.empty(L.total_bytes, dtype=torch.uint8, device=dev)
            workspace = self._ws
        stream = torch.cuda.current_stream(dev).cuda_stream
        self._check(lib.aof_flow_batch_device(
            self._ctx, prev.data_ptr(), cur.data_ptr(), pair_stride, n_pairs, blocks.data_ptr(),
            subdirs.data_ptr() if subdirs is not None else None, flows.data_ptr(),
            workspace.data_ptr(), workspace.numel(), stream))
        return blocks, flows, workspace

    def bind_batch(self, prev, cur, blocks, flows, workspace, subdirs=None, stream=None):
        """flow_batch with every argument resolved once: returns a callable that enqueues the same
        batch on torch's current stream (or on `stream`, a hipStream_t value) with ONE ctypes call (a step of 128 VGA pairs takes 35 us on
        the device; flow_batch's own argument handling takes about as long on the host)."""
        import torch
        p = self.params
        n_pairs = prev.shape[0]
        stride = prev.stride(0) if prev.dim() == 3 else p.width * p.height
        assert workspace.numel() >= workspace_layout(p, n_pairs).total_bytes
        args = (self._ctx, prev.data_ptr(), cur.data_ptr(), stride, n_pairs, blocks.data_ptr(),
                subdirs.data_ptr() if subdirs is not None else None, flows.data_ptr(),
                workspace.data_ptr(), workspace.numel())
        keep = (prev, cur, blocks, flows, workspace, subdirs)
        fn, dev, current_stream = lib.aof_flow_batch_device, prev.device, torch.cuda.current_stream

        if stream is not None:   # a fixed hipStream_t (0 = the default stream)
            def enqueue():
                rc = fn(*args, stream)
                if rc < 0:
                    self._check(rc)
        else:
            def enqueue():
                rc = fn(*args, current_stream(dev).cuda_stream)
                if rc < 0:
                    self._check(rc)
        enqueue.keep = keep
        return enqueue

    # -- a recorded frame sequence as one device pipeline -------------------------------
    def sequence(self, sp: SequenceParams, camera, time_us, gyro=None, workspace=None):
        """aof_sequence_device: camera uint8 CUDA tensor [n, cam_h, cam_w]; time_us int64 CUDA tensor [n]
        (microseconds relative to the first frame); gyro float32 CUDA tensor [n, 4] (integ_x, integ_y,
        integ_z, dt_s of the interval that ends at frame k) or None.  Everything is enqueued on torch's
        current stream.  Returns (workspace, layout): read the outputs with sequence_outputs()."""
        import torch
        n = camera.shape[0]
        L = sequence_layout(self.params, sp, n)
        if workspace is None:
            workspace = torch.empty(L.total_bytes, dtype=torch.uint8, device=camera.device)
        assert time_us.dtype == torch.int64 and time_us.numel() == n
        stream = torch.cuda.current_stream(camera.device).cuda_stream
        self._check(lib.aof_sequence_device(self._ctx, C.byref(sp), camera.data_ptr(), camera.stride(0) if n else 0, n,
                                            time_us.data_ptr(), gyro.data_ptr() if gyro is not None else None,
                                            workspace.data_ptr(), workspace.numel(), stream))
        return workspace, L

    def sequence_outputs(self, sp: SequenceParams, workspace, L: SeqLayout, n: int):
        """Host views of what a (completed) sequence() call left in its workspace."""
        p = self.params
        ws = workspace.cpu().numpy()
        pairs = max(n - 1, 0)
        count = ws[L.count:L.count + 16].view(np.uint32)
        out = {
            "records": ws[L.records:L.records + 32 * int(count[0])].view(SEQ_RECORD_DTYPE),
            "frames_sent": int(count[1]), "status": int(count[2]),
            "cropped": ws[L.cropped:L.cropped + n * p.width * p.height].reshape(n, p.height, p.width),
            "exposure": ws[L.exposure:L.exposure + 40 * n].view(np.uint32).reshape(n, 10),
            "flows": ws[L.flows:L.flows + 16 * pairs].view(FLOW_DTYPE),
            "derotated": ws[L.derotated:L.derotated + 8 * pairs].view(np.float32).reshape(pairs, 2) if sp.derotate else None,
        }
        lens = ws[L.frame_len:L.frame_len + int(count[0])]
        out["mavlink"] = [bytes(ws[L.frames + SEQ_FRAME_BYTES * m:L.frames + SEQ_FRAME_BYTES * m + int(lens[m])])
                          for m in range(int(count[0]))]
        return out

    # -- host buffers -----------------------------------------------------------
    def flow_pair_host(self, prev: np.ndarray, cur: np.ndarray):
        p = self.params
        prev = np.ascontiguousarray(prev, dtype=np.uint8)
        cur = np.ascontiguousarray(cur, dtype=np.uint8)
        assert prev.shape == cur.shape == (p.height, p.width)
        nb = self.nblocks(0)
        blocks = np.zeros(nb, dtype=BLOCK_DTYPE)
        subdirs = np.zeros(nb, dtype=np.uint8)
        flow = np.zeros(1, dtype=FLOW_DTYPE)
        self._check(lib.aof_flow_pair_host(self._ctx, prev.ctypes.data, cur.ctypes.data,
                                           blocks.ctypes.data, subdirs.ctypes.data,
                                           flow.ctypes.data))
        return blocks, subdirs, flow[0]

    def stream_push(self, frame: np.ndarray):
        """Returns None for the first frame, else the flow record vs the previous frame."""
        p = self.params
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        assert frame.shape == (p.height, p.width)
        flow = np.zeros(1, dtype=FLOW_DTYPE)
        rc = self._check(lib.aof_stream_push_host(self._ctx, frame.ctypes.data, flow.ctypes.data))
        return None if rc == 1 else flow[0]

    def set_stream_graph(self, on=True):
        """Streaming entry point: replay a captured hipGraph per frame (default) or launch eagerly."""
        self._check(lib.aof_set_stream_graph(self._ctx, int(on)))

    def set_stream_resident(self, on=True):
        """Streaming entry point for small frames: a resident one-workgroup kernel serves the calls
        through a mailbox in pinned memory instead of one launch per frame."""
        self._check(lib.aof_set_stream_resident(self._ctx, int(on)))

    def stream_resident_running(self) -> bool:
        return lib.aof_set_stream_resident(self._ctx, -1) == 1

    def stream_stats(self) -> dict:
        """Counters of the streaming entry point (``aof_stream_stats``)."""
        st = StreamStats()
        self._check(lib.aof_stream_get_stats(self._ctx, C.byref(st)))
        return st.as_dict()

    def debug_resident_fault(self, deaf=True, stop_wait_us=0):
        """Fault injection: resident kernels ignore the request to leave; the library's wait for them is shortened."""
        self._check(lib.aof_debug_resident_fault(self._ctx, int(deaf), int(stop_wait_us)))

    def set_vote_deadline_us(self, microseconds):
        """Deadline of the finaliser waves of the in-launch reduction (at least 100 us)."""
        self._check(lib.aof_set_vote_deadline_us(self._ctx, int(microseconds)))

    def debug_vote_deadline_ticks(self, ticks):
        """Fault injection: that deadline in 10 ns ticks, unchecked (0: every finaliser gives up at once)."""
        self._check(lib.aof_debug_vote_deadline_ticks(self._ctx, int(ticks)))

    def stream_graph_active(self) -> bool:
        return lib.aof_set_stream_graph(self._ctx, -1) == 1

    def stream_reset(self):
        self._check(lib.aof_stream_reset(self._ctx))


def blocks_view(t) -> np.ndarray:
    """int32 tensor/array of packed records -> structured numpy view."""
    a = t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)
    return np.ascontiguousarray(a).view(BLOCK_DTYPE).reshape(a.shape)


def flows_view(t) -> np.ndarray:
    a = t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)
    return np.ascontiguousarray(a).view(FLOW_DTYPE).reshape(a.shape[0])


# ---- the C++ facade (facade/libOpticalFlow.so) through its plain-C handles -------

FACADE_PATH = os.path.join(_HERE, "facade", "libOpticalFlow.so")
_facade = None


def facade_lib():
    """facade/libOpticalFlow.so: the drop-in classes OpticalFlowPX4 / OpticalFlowOpenCV."""
    global _facade
    if _facade is None:
        if not os.path.exists(FACADE_PATH):
            raise ImportError(f"{FACADE_PATH} is missing: run __graft_entry__.build()")
        f = C.CDLL(FACADE_PATH)
        f.aof_facade_px4_create.restype = C.c_void_p
        f.aof_facade_px4_create.argtypes = [C.c_float, C.c_float] + [C.c_int] * 6
        f.aof_facade_opencv_create.restype = C.c_void_p
        f.aof_facade_opencv_create.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        f.aof_facade_destroy.argtypes = [C.c_void_p]
        f.aof_facade_calc_flow.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int),
                                           C.POINTER(C.c_float), C.POINTER(C.c_float)]
        f.aof_facade_px4_track_features.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        f.aof_facade_pack_optical_flow_rad.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_float, C.c_float,
                                                       C.c_double, C.c_double, C.c_double, C.c_int, C.c_uint8,
                                                       C.c_void_p]
        f.aof_facade_mavlink_crc.restype = C.c_uint
        f.aof_facade_mavlink_crc.argtypes = [C.c_void_p, C.c_int]
        f.aof_facade_image_width.argtypes = [C.c_void_p]
        f.aof_facade_set_search_pyramid.argtypes = [C.c_void_p, C.c_int, C.c_int]
        f.aof_facade_pyramid_levels.argtypes = [C.c_void_p]
        f.aof_facade_set_resident.argtypes = [C.c_void_p, C.c_int]
        f.aof_facade_image_height.argtypes = [C.c_void_p]
        f.aof_facade_last_error.restype = C.c_char_p
        f.aof_facade_last_error.argtypes = [C.c_void_p]
        _facade = f
    return _facade


def pack_optical_flow_rad(offset_ts, img_time_us, dt_us, flow_x, flow_y, gyro=(0.0, 0.0, 0.0), quality=0,
                          seq=0) -> bytes:
    """The OPTICAL_FLOW_RAD MAVLink 2 frame the reference would send for these calcFlow
    outputs (/root/reference/src/mainloop.cpp:359-373, src/mavlink_tcp.cpp:142-162)."""
    buf = (C.c_uint8 * 56)()
    n = facade_lib().aof_facade_pack_optical_flow_rad(offset_ts, img_time_us, dt_us, flow_x, flow_y, gyro[0],
                                                      gyro[1], gyro[2], quality, seq, buf)
    return bytes(buf[:n])


DEFAULT_OUTPUT_RATE = 15
DEFAULT_IMAGE_WIDTH = 64
DEFAULT_IMAGE_HEIGHT = 64
DEFAULT_SEARCH_SIZE = 4
DEFAULT_FLOW_FEATURE_THRESHOLD = 30
DEFAULT_FLOW_VALUE_THRESHOLD = 3000


class _FacadeFlow:
    """Handle on one C++ facade object.  ``calcFlow`` keeps the reference's
    convention (/root/reference/src/mainloop.cpp:322-331): the return value is
    negative while the engine integrates towards its output rate, else the
    quality 0..255 with dt_us and the angular flow (rad)."""
    _h = None

    def getImageWidth(self):
        return facade_lib().aof_facade_image_width(self._h)

    def getImageHeight(self):
        return facade_lib().aof_facade_image_height(self._h)

    def lastError(self):
        return facade_lib().aof_facade_last_error(self._h).decode()

    def setSearchPyramid(self, levels, mean_subtract):
        return bool(facade_lib().aof_facade_set_search_pyramid(self._h, int(levels), int(bool(mean_subtract))))

    def getPyramidLevels(self):
        return facade_lib().aof_facade_pyramid_levels(self._h)

    def setResidentKernel(self, on=True):
        """Serve calcFlow() from a kernel that stays on the device (mailbox in pinned memory)."""
        return bool(facade_lib().aof_facade_set_resident(self._h, int(bool(on))))

    def calcFlow(self, img, img_time_us):
        """Returns (quality, dt_us, flow_x_rad, flow_y_rad); the C++ out-parameters keep
        their previous values (0 here) when the call does not publish."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        assert img.size == self.getImageWidth() * self.getImageHeight()
        dt, fx, fy = C.c_int(0), C.c_float(0), C.c_float(0)
        q = facade_lib().aof_facade_calc_flow(self._h, img.ctypes.data, int(img_time_us) & 0xFFFFFFFF,
                                              C.byref(dt), C.byref(fx), C.byref(fy))
        return q, dt.value, fx.value, fy.value

    def close(self):
        if self._h:
            facade_lib().aof_facade_destroy(self._h)
            self._h = None

    __del__ = close


class OpticalFlowPX4(_FacadeFlow):
    """facade/include/flow_px4.hpp"""

    def __init__(self, f_length_x, f_length_y, output_rate=DEFAULT_OUTPUT_RATE,
                 img_width=DEFAULT_IMAGE_WIDTH, img_height=DEFAULT_IMAGE_HEIGHT,
                 search_size=DEFAULT_SEARCH_SIZE,
                 flow_feature_threshold=DEFAULT_FLOW_FEATURE_THRESHOLD,
                 flow_value_threshold=DEFAULT_FLOW_VALUE_THRESHOLD):
        self._h = facade_lib().aof_facade_px4_create(f_length_x, f_length_y, output_rate, img_width,
                                                      img_height, search_size,
                                                      flow_feature_threshold, flow_value_threshold)


    def trackFeatures(self, img_prev, img_current, capacity=4096):
        """Rows of (prev_x, prev_y, cur_x, cur_y, sad, accepted) per grid tile."""
        a = np.ascontiguousarray(img_prev, dtype=np.uint8)
        b = np.ascontiguousarray(img_current, dtype=np.uint8)
        out = np.zeros((capacity, 6), dtype=np.float32)
        n = facade_lib().aof_facade_px4_track_features(self._h, a.ctypes.data, b.ctypes.data,
                                                       out.ctypes.data, capacity)
        if n < 0:
            raise AofError(n, self.lastError())
        return out[:min(n, capacity)]


class OpticalFlowOpenCV(_FacadeFlow):
    """facade/include/flow_opencv.hpp -- the class /root/reference/src/mainloop.cpp:423 creates."""

    def __init__(self, f_length_x, f_length_y, output_rate=DEFAULT_OUTPUT_RATE,
                 img_width=DEFAULT_IMAGE_WIDTH, img_height=DEFAULT_IMAGE_HEIGHT):
        self._h = facade_lib().aof_facade_opencv_create(f_length_x, f_length_y, output_rate,
                                                         img_width, img_height)
