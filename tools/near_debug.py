#!/usr/bin/env python3
"""Per-tile state of the near selection over a few synchronous frames of one pose: which tiles get repaired, with what
hints and selections.  usage: near_debug.py [workload] [pose: bench|orbit250|...]"""
import ctypes as C, math, sys
sys.path.insert(0, ".")
import numpy as np, torch, splat_amd
from splat_amd import _lib as L
from bench import WORKLOADS, make_scene
wl = sys.argv[1] if len(sys.argv) > 1 else "C3s"
n, W, H, seed = WORKLOADS[wl]
R = splat_amd.Renderer(); g = make_scene(wl); g.compute_cov3d(R); R.upload(g)
cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose(); cam_c = cam.to_c(0.01, 15)
img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
m = ((W + 15) // 16) * ((H + 15) // 16)
lib = L.lib()
def state():
    a = [np.zeros(m, np.uint32), np.zeros(m, np.uint32), np.zeros(m, np.uint32), np.zeros(4 * m, np.uint32)]
    rc = lib.splat_debug_near_state(R._h, *[x.ctypes.data_as(C.POINTER(C.c_uint32)) for x in a], C.c_uint32(m))
    assert rc == 0, rc
    return a
for f in range(8):
    st = R.render_frame_device(cam_c, img.data_ptr(), sync=True, want_stats=True)
    lens, near_m, mask, hint = state()
    rep = np.flatnonzero((mask != 0) & (lens > 2048))
    print("frame %d: served %d repaired %d | repaired tiles: %s" % (f, st.n_near_tiles, st.n_near_fallback,
          [(int(t), int(lens[t]), int(near_m[t]), hex(int(mask[t])), [int(v) if v != 0xffffffff else -1 for v in hint[4 * t:4 * t + 4]]) for t in rep[:6]]))
    if f == 0:
        track = rep[:3]
    for t in (track if f else []):
        print("    tile %d: len %d near_m %d mask %s hints %s" % (t, lens[t], near_m[t], hex(int(mask[t])), [int(v) if v != 0xffffffff else -1 for v in hint[4 * t:4 * t + 4]]))
R.close()
