#!/usr/bin/env python3
"""Front half of a K1 wave's life (experiment build -DSPLAT_K1X=31): SPLAT_AMD_LIB=build/libsplat_k1x31.so SPLAT_PIPELINE=1 python tools/k1_timeline_front.py"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np, splat_amd
from splat_amd import _lib
from bench import WORKLOADS
n, W, H, seed = WORKLOADS["C3"]
R = splat_amd.Renderer(); g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R)
cam = splat_amd.Camera(H, W, (0, 0, 5.0)); cam.update_camera_pose(); R.upload(g)
img = np.zeros((H, W), np.uint32)
for _ in range(3):
    R.render(cam.to_c(0.01, 15), img)
L = _lib.lib()
nb = (n + 255) // 256
buf = np.zeros((nb * 4, 8), np.uint64)
L.splat_debug_k1_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert L.splat_debug_k1_stamps(R._h, buf.ctypes.data, buf.size) == 0
t = buf.astype(np.int64).reshape(nb, 4, 8)
names = ["start -> table cleared / culling test + barrier", "geometry planes landed", "projection + conic + NDC arithmetic", "covered pixel intervals, SH loads issued",
         "count pass", "SH loads waited + barrier + reservations issued", "SH arithmetic, publish, hand-out, stores -> end"]
ok = (t > 0).all(axis=(1, 2))
t = t[ok]
d = np.diff(t, axis=2)
for k, nm in enumerate(names):
    v = d[:, :, k]
    print("%-52s per wave: median %7.0f  mean %7.0f  p90 %7.0f" % (nm, np.median(v), v.mean(), np.percentile(v, 90)))
for w in range(4):
    print("wave %d: start->barrier0 mean %.0f" % (w, d[:, w, 0].mean()))
