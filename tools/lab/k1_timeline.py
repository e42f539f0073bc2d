#!/usr/bin/env python3
"""Where a K1 wave's life goes (experiment build with s_memtime stamps: -DSPLAT_K1X=30):
SPLAT_AMD_LIB=build/libsplat_k1x30.so SPLAT_PIPELINE=1 python tools/k1_timeline.py"""
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
rc = L.splat_debug_k1_stamps(R._h, buf.ctypes.data, buf.size)
assert rc == 0
t = buf.astype(np.int64).reshape(nb, 4, 8)
names = ["start -> geometry done, SH loads issued", "count pass (LDS atomics)", "SH loads waited", "barrier + reservations issued",
         "SH arithmetic", "reservations waited + published", "record store? + barrier of the hand-out"]
ok = (t > 0).all(axis=(1, 2))
print("blocks", nb, "complete stamps", ok.sum())
t = t[ok]
d = np.diff(t, axis=2)
for k, nm in enumerate(names):
    v = d[:, :, k]
    print("%-44s per wave: median %7.0f  mean %7.0f  p90 %7.0f | slowest wave of the block: median %7.0f ticks" % (nm, np.median(v), v.mean(), np.percentile(v, 90), np.median(v.max(axis=1))))
arr = t[:, :, 6]          # arrival at the last barrier
print("arrival spread at the hand-out barrier (max - min over the block's waves): median %.0f mean %.0f p90 %.0f ticks" % (
    np.median(arr.max(1) - arr.min(1)), (arr.max(1) - arr.min(1)).mean(), np.percentile(arr.max(1) - arr.min(1), 90)))
arr1 = t[:, :, 3]
print("arrival spread at the count barrier: median %.0f mean %.0f" % (np.median(arr1.max(1) - arr1.min(1)), (arr1.max(1) - arr1.min(1)).mean()))
life = t[:, :, 7].max(1) - t[:, :, 0].min(1)
print("block life to the hand-out barrier: median %.0f mean %.0f; kernel span %.0f ticks" % (np.median(life), life.mean(), t[:, :, 7].max() - t[:, :, 0].min()))
for w in range(4):
    print("wave %d: mean time start->barrier2 arrival %.0f" % (w, (t[:, w, 6] - t[:, w, 0]).mean()))
# ---- residency: which CU every block ran on, and how many blocks a CU held at a time
hw = np.zeros(nb * 4, np.uint64)
L.splat_debug_k1_hwid.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
assert L.splat_debug_k1_hwid(R._h, hw.ctypes.data, hw.size) == 0
hw = hw.reshape(nb, 4)[ok]
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
# gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx940: se_id [14:13]?) -- print what varies
for name, lo, bits in (("wave", 0, 4), ("simd", 4, 2), ("pipe", 6, 2), ("cu", 8, 4), ("sh", 12, 1), ("se", 13, 3), ("tg", 16, 4), ("vm", 20, 4), ("queue", 24, 3), ("state", 27, 3), ("me", 30, 2)):
    v = (hwid >> lo) & ((1 << bits) - 1)
    print("HW_ID %-6s distinct values %s" % (name, np.unique(v)[:20]))
print("XCC_ID distinct", np.unique(xcc), " block%8 == xcc for", float((xcc[:, 0] == (np.nonzero(ok)[0] % 8)).mean()))
cu_key = xcc[:, 0] * 4096 + ((hwid[:, 0] >> 8) & 0xff)          # xcc, (se, sh, cu)
simds = (hwid >> 4) & 3
print("waves of a block on distinct SIMDs:", float((np.sort(simds, axis=1) == np.arange(4)).all(axis=1).mean()))
start, end = t[:, :, 0].min(1), t[:, :, 7].max(1)
print("block life start->end: median %.0f mean %.0f ticks" % (np.median(end - start), (end - start).mean()))
res = []
for k in np.unique(cu_key):
    m = cu_key == k
    ev = np.concatenate([np.stack([start[m], np.ones(m.sum())], 1), np.stack([end[m], -np.ones(m.sum())], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    conc = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0])
    res.append((m.sum(), (conc[:-1] * dt).sum() / max(dt.sum(), 1), conc.max(), ev[-1, 0] - ev[0, 0]))
res = np.array(res)
print("CUs seen %d; blocks per CU mean %.1f; time-averaged resident blocks per CU %.2f (max seen %d); busy span per CU mean %.0f ticks" % (
    len(res), res[:, 0].mean(), res[:, 1].mean(), res[:, 2].max(), res[:, 3].mean()))
d2 = (t[:, :, 7] - t[:, :, 6])
print("hand-out barrier -> end per wave: median %.0f mean %.0f p90 %.0f" % (np.median(d2), d2.mean(), np.percentile(d2, 90)))
