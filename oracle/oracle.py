"""ctypes wrapper around oracle/liboracle.so -- the CPU ORACLE.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from splat_amd/ (the product).  See
oracle/splat_oracle.h for what it restates and its parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Conventions(C.Structure):
    _fields_ = [("y_up", C.c_int32), ("sample_half", C.c_int32), ("zclip", C.c_int32),
                ("zmin", C.c_float), ("zmax", C.c_float), ("raster", C.c_int32), ("corrected_projection", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("w", C.c_float), ("h", C.c_float),
                ("htanx", C.c_float), ("htany", C.c_float), ("focal", C.c_float),
                ("cam_pos", C.c_float * 3), ("lowpass", C.c_float), ("sh_dim", C.c_int32)]


class Record(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("hx", C.c_float), ("hy", C.c_float),
                ("conic", C.c_float * 3), ("opacity", C.c_float), ("rgb", C.c_float * 3),
                ("depth", C.c_float), ("ndc", C.c_float * 4), ("cov2d", C.c_float * 4),
                ("visible", C.c_int32), ("px0", C.c_int32), ("px1", C.c_int32),
                ("py0", C.c_int32), ("py1", C.c_int32)]


RECORD_DTYPE = np.dtype([("cx", "f4"), ("cy", "f4"), ("hx", "f4"), ("hy", "f4"), ("conic", "f4", 3),
                         ("opacity", "f4"), ("rgb", "f4", 3), ("depth", "f4"), ("ndc", "f4", 4),
                         ("cov2d", "f4", 4), ("visible", "i4"), ("px0", "i4"), ("px1", "i4"),
                         ("py0", "i4"), ("py1", "i4")])
assert RECORD_DTYPE.itemsize == C.sizeof(Record)


class Stats(C.Structure):
    _fields_ = [("n_visible", C.c_uint64), ("n_singular", C.c_uint64), ("n_fragments", C.c_uint64),
                ("n_tile_pairs", C.c_uint64), ("ms_preprocess", C.c_double), ("ms_sort", C.c_double),
                ("ms_raster", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        L.orc_default_conventions.argtypes = [C.POINTER(Conventions)]
        L.orc_camera_make.argtypes = [C.c_float, C.c_float, fp, C.c_float, C.c_float, C.c_float, C.c_int32,
                                      C.POINTER(Camera)]
        L.orc_compute_cov3d.argtypes = [C.c_uint64, fp, fp, fp]
        L.orc_eval_sh.argtypes = [fp, C.c_int32, fp, fp]
        L.orc_project_cov2d.argtypes = [fp, fp, C.POINTER(Camera), fp]
        L.orc_project_cov2d_corrected.argtypes = [fp, fp, C.POINTER(Camera), fp]
        L.orc_sort.argtypes = [C.c_uint64, fp, fp, C.POINTER(C.c_uint32)]
        L.orc_preprocess.argtypes = [C.c_uint64, fp, fp, fp, fp, C.POINTER(Camera), C.POINTER(Conventions),
                                     C.c_void_p]
        L.orc_fragment.argtypes = [fp, fp]
        L.orc_blend.argtypes = [C.c_uint32, fp]
        L.orc_blend.restype = C.c_uint32
        L.orc_render.argtypes = [C.c_uint64, fp, fp, fp, fp, C.POINTER(Camera), C.POINTER(Conventions),
                                 C.POINTER(C.c_uint32), C.c_int32, C.c_int32, C.c_int32, C.POINTER(Stats)]
        L.orc_render.restype = C.c_int
        L.orc_load_ply.argtypes = [C.c_char_p, fp, fp, fp, fp, fp]
        L.orc_load_ply.restype = C.c_int64
        _LIB = L
    return _LIB


def _fp(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_conventions(**kw):
    c = Conventions()
    lib().orc_default_conventions(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def camera(h, w, pos=(0.0, 0.0, 3.0), yaw=0.0, pitch=0.0, lowpass=0.01, sh_dim=15):
    cam = Camera()
    p = np.asarray(pos, np.float32)
    lib().orc_camera_make(float(h), float(w), _fp(p), float(yaw), float(pitch), float(lowpass), int(sh_dim),
                          C.byref(cam))
    return cam


def compute_cov3d(scales, rot):
    scales = np.ascontiguousarray(scales, np.float32)
    rot = np.ascontiguousarray(rot, np.float32)
    n = scales.shape[0]
    out = np.zeros((n, 9), np.float32)
    lib().orc_compute_cov3d(n, _fp(scales), _fp(rot), _fp(out))
    return out


def eval_sh(sh48, sh_dim, d):
    sh48 = np.ascontiguousarray(sh48, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    out = np.zeros(3, np.float32)
    lib().orc_eval_sh(_fp(sh48), int(sh_dim), _fp(d), _fp(out))
    return out


def project_cov2d(pos, cov3d, cam, corrected=False):
    pos = np.ascontiguousarray(pos, np.float32)
    cov3d = np.ascontiguousarray(cov3d, np.float32)
    out = np.zeros(4, np.float32)
    (lib().orc_project_cov2d_corrected if corrected else lib().orc_project_cov2d)(_fp(pos), _fp(cov3d), C.byref(cam), _fp(out))
    return out.reshape(2, 2).T  # column-major -> [r][c]


def sort(pos4, view16):
    pos4 = np.ascontiguousarray(pos4, np.float32)
    v = np.ascontiguousarray(view16, np.float32)
    n = pos4.shape[0]
    out = np.zeros(n, np.uint32)
    lib().orc_sort(n, _fp(pos4), _fp(v), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def preprocess(scene, cam, conv=None):
    conv = conv or default_conventions()
    n = scene["pos4"].shape[0]
    out = np.zeros(n, RECORD_DTYPE)
    lib().orc_preprocess(n, _fp(scene["pos4"]), _fp(scene["cov3d"]), _fp(scene["opacity"]), _fp(scene["sh"]),
                         C.byref(cam), C.byref(conv), out.ctypes.data)
    return out


def fragment(vdata9):
    v = np.ascontiguousarray(vdata9, np.float32)
    out = np.zeros(4, np.float32)
    lib().orc_fragment(_fp(v), _fp(out))
    return out


def blend(old, frag4):
    f = np.ascontiguousarray(frag4, np.float32)
    return int(lib().orc_blend(int(old) & 0xFFFFFFFF, _fp(f)))


def render(scene, cam, conv=None, argb=None, rows=None, nthreads=1):
    """scene: dict(pos4[n,4], cov3d[n,9], opacity[n], sh[n,48]) float32 C-contiguous.
    Returns (argb[h,w] uint32, Stats)."""
    conv = conv or default_conventions()
    W, H = int(cam.w), int(cam.h)
    if argb is None:
        argb = np.zeros((H, W), np.uint32)
    assert argb.dtype == np.uint32 and argb.shape == (H, W) and argb.flags.c_contiguous
    r0, r1 = rows if rows is not None else (0, H)
    st = Stats()
    n = scene["pos4"].shape[0]
    rc = lib().orc_render(n, _fp(scene["pos4"]), _fp(scene["cov3d"]), _fp(scene["opacity"]), _fp(scene["sh"]),
                          C.byref(cam), C.byref(conv), argb.ctypes.data_as(C.POINTER(C.c_uint32)),
                          int(r0), int(r1), int(nthreads), C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_render failed: %d" % rc)
    return argb, st


def load_ply(path):
    L = lib()
    n = L.orc_load_ply(path.encode(), None, None, None, None, None)
    if n < 0:
        raise RuntimeError("orc_load_ply(%s) failed: %d" % (path, n))
    pos4 = np.zeros((n, 4), np.float32)
    scales = np.zeros((n, 3), np.float32)
    opacity = np.zeros(n, np.float32)
    rot = np.zeros((n, 4), np.float32)
    sh = np.zeros((n, 48), np.float32)
    m = L.orc_load_ply(path.encode(), _fp(pos4), _fp(scales), _fp(opacity), _fp(rot), _fp(sh))
    if m != n:
        raise RuntimeError("orc_load_ply(%s) failed: %d" % (path, m))
    return dict(pos4=pos4, scales=scales, opacity=opacity, rot=rot, sh=sh)
