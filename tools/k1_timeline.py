#!/usr/bin/env python3
"""Where a K1 block's life goes (experiment build with s_memtime stamps):
SPLAT_AMD_LIB=build/libsplat_tl.so python tools/k1_timeline.py"""
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
buf = np.zeros((8192, 8), np.uint64)
rc = L.splat_debug_k1_stamps(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
nb = (n + 255) // 256
t = buf[:nb].astype(np.int64)
names = ["start->geometry loaded", "geometry math", "count + barrier", "reservations issued", "SH + record", "reservations waited + barrier", "scatter + close-ups"]
d = np.diff(t, axis=1)
ok = (t > 0).all(axis=1)
print("blocks", nb, "complete stamps", ok.sum())
for k, nm in enumerate(names):
    v = d[ok, k]
    print("%-32s median %7.0f  mean %7.0f  p90 %7.0f ticks" % (nm, np.median(v), v.mean(), np.percentile(v, 90)))
life = t[ok, 7] - t[ok, 0]
print("block life median %.0f mean %.0f; kernel span %.0f ticks" % (np.median(life), life.mean(), t[ok, 7].max() - t[ok, 0].min()))
