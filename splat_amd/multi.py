"""Multi-GPU through the C ABI (include/splat_hip.h, "Multi-GPU"): tile-row slabs + one RCCL gather of slab
rows per frame.  `MultiRenderer` = form (B), one process with a host thread and a context per device;
`slab_partition_native` / `Renderer.comm_*` (renderer.py) = form (A), one process per GPU."""
import ctypes as C

import numpy as np

from . import _lib
from .renderer import SplatError, _fp


def slab_partition_native(row_loads, n_ranks, row_overhead=0.0, n_rows=None):
    """splat_slab_partition: the partition every native caller derives.  row_loads=None: equal split of n_rows."""
    L = _lib.lib()
    out = (C.c_int32 * (2 * n_ranks))()
    if row_loads is None:
        rc = L.splat_slab_partition(None, int(n_rows), int(n_ranks), float(row_overhead), out)
    else:
        loads = np.ascontiguousarray(row_loads, np.uint64)
        rc = L.splat_slab_partition(loads.ctypes.data_as(C.POINTER(C.c_uint64)), len(loads), int(n_ranks),
                                    float(row_overhead), out)
    if rc != 0:
        raise SplatError(rc, "splat_slab_partition")
    return [(out[2 * i], out[2 * i + 1]) for i in range(n_ranks)]


class MultiRenderer:
    """devices: HIP ordinals, rank 0 first (the root).  A device listed twice shares that GPU between two slabs
    (copy transport; for testing the decomposition on one GPU)."""

    def __init__(self, devices, **conventions):
        self._L = _lib.lib()
        cfg = _lib.Config()
        self._L.splat_default_config(C.byref(cfg))
        for k, v in conventions.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown convention %r" % k)
            setattr(cfg, k, v)
        self.devices = [int(d) for d in devices]
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        h = C.c_void_p()
        rc = self._L.splat_multi_create(C.byref(cfg), arr, len(self.devices), C.byref(h))
        if rc != 0:
            raise SplatError(rc, (self._L.splat_multi_last_error(None) or b"").decode())
        self._h = h
        self.n = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.splat_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise SplatError(rc, (self._L.splat_multi_last_error(self._h) or b"").decode())

    def upload(self, g):
        self._check(self._L.splat_multi_upload_scene(self._h, len(g), _fp(g.positions), _fp(g.cov3d), _fp(g.opacities),
                                                     _fp(g.sh)))
        self.n = len(g)

    def set_frame_overlap(self, n):
        """2: every device renders into two slab images in turn, consecutive frames composite side by side (splat_multi_set_frame_overlap)"""
        self._check(self._L.splat_multi_set_frame_overlap(self._h, int(n)))

    def balance(self, cam_c):
        self._check(self._L.splat_multi_balance(self._h, C.byref(cam_c)))
        return self.slabs()

    def slabs(self):
        out = (C.c_int32 * (2 * len(self.devices)))()
        self._check(self._L.splat_multi_get_slabs(self._h, out))
        return [(out[2 * i], out[2 * i + 1]) for i in range(len(self.devices))]

    def render(self, cam_c, argb, want_stats=True):
        """render_to_buffer across the devices: blends onto argb (uint32 [h,w], host) in place."""
        assert argb.dtype == np.uint32 and argb.flags.c_contiguous and argb.shape == (int(cam_c.h), int(cam_c.w))
        st = _lib.Stats()
        self._check(self._L.splat_multi_render(self._h, C.byref(cam_c), argb.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               C.byref(st) if want_stats else None))
        return st

    def render_frame(self, cam_c):
        """viewer-loop frame (clear + render + gather), asynchronous"""
        self._check(self._L.splat_multi_render_frame(self._h, C.byref(cam_c)))

    def sync(self):
        self._check(self._L.splat_multi_sync(self._h))

    def download(self, h, w):
        out = np.zeros((int(h), int(w)), np.uint32)
        self._check(self._L.splat_multi_download(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), int(w), int(h)))
        return out

    def rank_timing(self, rank, reset=True):
        """per-kernel device time accumulated by one rank's context (see Renderer.timing)"""
        ctx = self._L.splat_multi_ctx(self._h, int(rank))
        ms = (C.c_double * 6)()
        frames = C.c_uint64()
        rc = self._L.splat_get_timing(C.c_void_p(ctx), ms, C.byref(frames), 1 if reset else 0)
        if rc != 0:
            raise SplatError(rc, (self._L.splat_last_error(C.c_void_p(ctx)) or b"").decode())
        names = ("preprocess", "scan", "emit", "sort", "composite", "status")
        return {k: ms[i] for i, k in enumerate(names)}, frames.value

    def rank_frames_dropped(self, rank):
        """frames one rank's device skipped since creation (splat_frames_dropped of its context)"""
        ctx = self._L.splat_multi_ctx(self._h, int(rank))
        return int(self._L.splat_frames_dropped(C.c_void_p(ctx)))
