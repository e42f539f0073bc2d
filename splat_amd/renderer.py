"""Thin object wrapper over the C ABI context (include/splat_hip.h)."""
import ctypes as C

import numpy as np

from . import _lib

f32 = np.float32


class SplatError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("splat error %d: %s" % (code, msg))
        self.code = code


def _fp(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Renderer:
    """One context = one GPU = one stream.  `conventions` overrides splat_default_config fields
    (y_up, sample_half, zclip, zmin, zmax)."""

    def __init__(self, device=0, pair_capacity=0, **conventions):
        self._L = _lib.lib()
        cfg = _lib.Config()
        self._L.splat_default_config(C.byref(cfg))
        cfg.device = int(device)
        cfg.pair_capacity = int(pair_capacity)
        for k, v in conventions.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown convention %r" % k)
            setattr(cfg, k, v)
        self.config = cfg
        h = C.c_void_p()
        rc = self._L.splat_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise SplatError(rc, (self._L.splat_last_error(None) or b"").decode())
        self._h = h
        self.n = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.splat_destroy(self._h)
            self._h = None
            for p in getattr(self, "_pinned", []):
                self._L.splat_host_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise SplatError(rc, (self._L.splat_last_error(self._h) or b"").decode())

    # ---- scene ------------------------------------------------------------------------
    def upload(self, gaussians):
        g = gaussians
        self._check(self._L.splat_upload_scene(self._h, len(g), _fp(g.positions), _fp(g.cov3d), _fp(g.opacities),
                                               _fp(g.sh)))
        self.n = len(g)

    def compute_cov3d(self, scales, rotations):
        scales = np.ascontiguousarray(scales, f32)
        rotations = np.ascontiguousarray(rotations, f32)
        n = scales.shape[0]
        out = np.zeros((n, 9), f32)
        self._check(self._L.splat_compute_cov3d(self._h, n, _fp(scales), _fp(rotations), _fp(out)))
        return out

    def set_slab(self, tile_row0=0, tile_row1=-1):
        self._check(self._L.splat_set_slab(self._h, int(tile_row0), int(tile_row1)))

    def tile_row_loads(self, cam_c):
        """(Gaussian, tile) pairs per tile row of the full frame: the load estimate for slab balancing."""
        n_rows = (int(cam_c.h) + _lib.TILE - 1) // _lib.TILE
        out = np.zeros(n_rows, np.uint64)
        self._check(self._L.splat_tile_row_loads(self._h, C.byref(cam_c), out.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                 n_rows))
        return out

    def set_frame_overlap(self, n):
        """2: asynchronous frames to DIFFERENT device images (a swap chain) may composite side by side (include/splat_hip.h,
        splat_set_frame_overlap); 1 (default): one compositor at a time, in call order."""
        self._check(self._L.splat_set_frame_overlap(self._h, int(n)))

    def set_option(self, option, value):
        """splat_set_option: a tuning option (_lib.OPT_*) from code instead of from the environment; never changes a pixel"""
        self._check(self._L.splat_set_option(self._h, int(option), float(value)))

    def get_option(self, option):
        v = C.c_double()
        self._check(self._L.splat_get_option(self._h, int(option), C.byref(v)))
        return v.value

    def set_stream(self, stream_ptr):
        self._check(self._L.splat_set_stream(self._h, C.c_void_p(stream_ptr)))

    # ---- frames -----------------------------------------------------------------------
    def render(self, cam_c, argb, want_stats=True):
        """argb: uint32 [h,w] host array, blended in place."""
        assert argb.dtype == np.uint32 and argb.flags.c_contiguous
        assert argb.shape == (int(cam_c.h), int(cam_c.w))
        st = _lib.Stats()
        self._check(self._L.splat_render(self._h, C.byref(cam_c), argb.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         C.byref(st) if want_stats else None))
        return st

    def render_frame(self, cam_c, argb_out, want_stats=False):
        """splat_render_frame: the viewer loop's `clear; render_to_buffer` (src/main.rs:73-74) as one synchronous call; argb_out
        (uint32 [h,w], host) is written, never read.  A page-locked image (host_image / host_register) is written by the
        compositor itself."""
        assert argb_out.dtype == np.uint32 and argb_out.flags.c_contiguous
        assert argb_out.shape == (int(cam_c.h), int(cam_c.w))
        st = _lib.Stats() if want_stats else None
        self._check(self._L.splat_render_frame(self._h, C.byref(cam_c), argb_out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               C.byref(st) if want_stats else None))
        return st

    def render_device(self, cam_c, d_ptr, sync=False, want_stats=False):
        """d_ptr: device address of a w*h u32 image (e.g. torch_tensor.data_ptr())."""
        st = _lib.Stats() if want_stats else None
        self._check(self._L.splat_render_device(self._h, C.byref(cam_c), C.c_void_p(d_ptr), 1 if sync else 0,
                                                C.byref(st) if want_stats else None))
        return st

    # ---- multi-GPU, one process per GPU (form A of include/splat_hip.h "Multi-GPU") ---------
    @staticmethod
    def comm_unique_id():
        """rank 0: the 128-byte RCCL id to hand to the other ranks"""
        L = _lib.lib()
        buf = (C.c_uint8 * _lib.UNIQUE_ID_BYTES)()
        rc = L.splat_comm_unique_id(buf)
        if rc != 0:
            raise SplatError(rc, "splat_comm_unique_id")
        return bytes(buf)

    def comm_init(self, unique_id, n_ranks, rank):
        buf = (C.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._check(self._L.splat_comm_init_rank(self._h, buf, int(n_ranks), int(rank)))

    def comm_set_slabs(self, slabs):
        flat = (C.c_int32 * (2 * len(slabs)))(*[int(v) for s in slabs for v in s])
        self._check(self._L.splat_comm_set_slabs(self._h, flat))

    def comm_loopback(self, on=True):
        """test hook: a single-rank communicator's gather sends this rank's rows through RCCL to itself"""
        self._check(self._L.splat_comm_loopback(self._h, 1 if on else 0))

    def comm_gather(self, d_ptr, w, h, root=0):
        """enqueue the gather of slab rows to `root` on the context's stream (grouped ncclSend / ncclRecv)"""
        self._check(self._L.splat_comm_gather(self._h, C.c_void_p(d_ptr), int(w), int(h), int(root)))

    # ---- device images without a HIP toolchain on the caller's side ------------------------
    def device_image(self, init):
        """Device copy of a uint32 [h,w] host image; returns its device address (free with device_free)."""
        assert init.dtype == np.uint32 and init.flags.c_contiguous
        p = self._L.splat_device_alloc(self._h, init.nbytes)
        if not p:
            raise SplatError(_lib.ERR_HIP, (self._L.splat_last_error(self._h) or b"").decode())
        self._check(self._L.splat_device_upload(self._h, C.c_void_p(p), C.c_void_p(init.ctypes.data), init.nbytes))
        return p

    def device_download(self, d_ptr, h, w):
        out = np.zeros((int(h), int(w)), np.uint32)
        self._check(self._L.splat_device_download(self._h, C.c_void_p(out.ctypes.data), C.c_void_p(d_ptr), out.nbytes))
        return out

    def device_free(self, d_ptr):
        self._L.splat_device_free(self._h, C.c_void_p(d_ptr))

    def render_frame_device(self, cam_c, d_ptr, sync=False, want_stats=False):
        """the viewer loop's frame (clear + render_to_buffer, src/main.rs:73-74) on a device image; the clear is fused
        into the compositor"""
        st = _lib.Stats() if want_stats else None
        self._check(self._L.splat_render_frame_device(self._h, C.byref(cam_c), C.c_void_p(d_ptr), 1 if sync else 0,
                                                      C.byref(st) if want_stats else None))
        return st

    def sync(self):
        """Wait for everything enqueued.  Raises SplatError(ERR_CAPACITY) once if an ASYNCHRONOUS frame was
        skipped on the device (storage has been grown: render it again)."""
        self._check(self._L.splat_sync(self._h))

    def device_bytes(self):
        """(bytes of device memory the context holds now, the most it has held)"""
        peak = C.c_uint64()
        now = int(self._L.splat_device_bytes(self._h, C.byref(peak)))
        return now, int(peak.value)

    def frames_dropped(self):
        """frames skipped on the device since the context was created (redone internally or reported)"""
        return int(self._L.splat_frames_dropped(self._h))

    # ---- viewer-loop streaming (src/main.rs:69-78): cleared frame -> async copy into a host buffer
    @staticmethod
    def host_register(arr):
        """page-lock a numpy image the caller owns (splat_host_register): render() then copies it at the PCIe rate;
        call host_unregister(arr) before the array goes away"""
        rc = _lib.lib().splat_host_register(C.c_void_p(arr.ctypes.data), arr.nbytes)
        if rc != 0:
            raise SplatError(rc, "splat_host_register")

    @staticmethod
    def host_unregister(arr):
        _lib.lib().splat_host_unregister(C.c_void_p(arr.ctypes.data))

    def host_image(self, h, w):
        """a pinned (page-locked) h x w uint32 image; keep the Renderer alive while it is in use"""
        p = self._L.splat_host_alloc(int(h) * int(w) * 4)
        if not p:
            raise MemoryError("splat_host_alloc")
        arr = np.ctypeslib.as_array((C.c_uint32 * (int(h) * int(w))).from_address(p)).reshape(int(h), int(w))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return arr

    def render_stream(self, cam_c, out):
        assert out.dtype == np.uint32 and out.flags.c_contiguous
        self._check(self._L.splat_render_stream(self._h, C.byref(cam_c), C.c_void_p(out.ctypes.data)))

    def stream_wait(self, out):
        self._check(self._L.splat_stream_wait(self._h, C.c_void_p(out.ctypes.data)))

    def timing(self, reset=True):
        ms = (C.c_double * 6)()
        frames = C.c_uint64()
        self._check(self._L.splat_get_timing(self._h, ms, C.byref(frames), 1 if reset else 0))
        names = ("preprocess", "scan", "emit", "sort", "composite", "status")
        return {k: ms[i] for i, k in enumerate(names)}, frames.value

    # ---- debug / stage parity -----------------------------------------------------------
    def records(self):
        dt = np.dtype([("cx", "f4"), ("cy", "f4"), ("hx", "f4"), ("hy", "f4"), ("conic", "f4", 3),
                       ("opacity", "f4"), ("rgb", "f4", 3), ("depth", "f4"), ("px0", "i4"), ("px1", "i4"),
                       ("py0", "i4"), ("py1", "i4")])
        assert dt.itemsize == C.sizeof(_lib.Record)
        out = np.zeros(self.n, dt)
        self._check(self._L.splat_get_records(self._h, out.ctypes.data_as(C.POINTER(_lib.Record)), self.n))
        return out

    def binning_mode(self):
        """bucket size (keys per tile) of the last frame's one-pass binning, 0 = two-pass, < 0 = no frame"""
        return int(self._L.splat_binning_mode(self._h))

    def tile_lists(self, n_tiles, n_pairs):
        off = np.zeros(n_tiles + 1, np.uint32)
        order = np.zeros(n_pairs, np.uint32)
        self._check(self._L.splat_get_tile_lists(self._h, off.ctypes.data_as(C.POINTER(C.c_uint32)), n_tiles + 1,
                                                 order.ctypes.data_as(C.POINTER(C.c_uint32)), n_pairs))
        return off, order
