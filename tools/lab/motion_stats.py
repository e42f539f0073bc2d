#!/usr/bin/env python3
"""What a MOVING camera's frame costs the compositor, against the same pose at rest: per pose of a yaw path, a synchronous
statistics frame right after the step (scan in the walks, selections sized for motion) and the sixth frame at that pose (start
hints): (wave, record) iterations of scan and blend, early-out retries, tiles served by a near selection / repaired, per-kernel
times alone.   usage: motion_stats.py [C3s] [step_degrees=3]"""
import math, sys
sys.path.insert(0, ".")
import numpy as np, torch, splat_amd
from splat_amd import _lib as L
from bench import WORKLOADS, make_scene
wl = sys.argv[1] if len(sys.argv) > 1 else "C3s"
step = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
n, W, H, seed = WORKLOADS[wl]
R = splat_amd.Renderer(); g = make_scene(wl); g.compute_cov3d(R); R.upload(g)
img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose()
R.set_option(L.OPT_TIMING_EVERY, 1)
for k in range(12):
    cam.update_yaw_angle(math.radians(step)); cam.update_camera_pose()
    c = cam.to_c(0.01, 15)
    # two plain frames along the path first, so that the statistics frame is not the first after a rest
    mv = R.render_frame_device(c, img.data_ptr(), sync=True, want_stats=True)
    for _ in range(5):
        R.render_frame_device(c, img.data_ptr(), sync=True)
    rs = R.render_frame_device(c, img.data_ptr(), sync=True, want_stats=True)
    f = lambda s: "scan %8d blend %8d retries %4d near %4d repaired %3d | K1 %.3f scan %.3f sel %.3f K4 %.3f" % (
        s.n_iter_scan, s.n_iter_blend, s.n_fallback, s.n_near_tiles, s.n_near_fallback, s.ms_preprocess, s.ms_scan, s.ms_sort, s.ms_composite)
    print("%s pose %2d pairs %9d maxlen %6d\n   moving : %s\n   at rest: %s" % (wl, k, mv.n_pairs, mv.max_tile_len, f(mv), f(rs)))
R.close()
