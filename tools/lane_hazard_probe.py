#!/usr/bin/env python3
"""Timing probe for the compositor lanes (splat_set_frame_overlap): three images in rotation, X (inside the cloud: long binning chain),
W (far away: long compositor), Y, X again -- after a few synchronous frames, so that every repetition starts from the same lanes.
Prints whether X ends up holding its second frame, and where it does not (a build that decided hazards from each lane's last frame
only left the first frame's tail in 50-95 k pixels of the first repetition).   usage: [OV=1|2] python tools/lane_hazard_probe.py"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, splat_amd
from helpers import make_camera
H, W = 480, 640
seq = [make_camera(H, W, (0.1, 0.1, 0.6), yaw=0.4).to_c(0.01), make_camera(H, W, (0.0, 0.0, 9.0)).to_c(0.01),
       make_camera(H, W, (0.11, 0.1, 0.61), yaw=0.4).to_c(0.01), make_camera(H, W, (0.02, 0.0, 9.05)).to_c(0.01)]
g = splat_amd.synthetic_scene(700000, 92)
r = splat_amd.Renderer(); g.compute_cov3d(r); r.upload(g)
init = np.zeros((H, W), np.uint32)
x, w, y = r.device_image(init), r.device_image(init), r.device_image(init)
want = []
for c in seq:
    st = r.render_frame_device(c, x, sync=True, want_stats=True)
    want.append(r.device_download(x, H, W))
    print("pose: pairs %d max list %d  ms: K1 %.3f sort %.3f K4 %.3f" % (st.n_pairs, st.max_tile_len, st.ms_preprocess, st.ms_sort, st.ms_composite))
import os
r.set_frame_overlap(int(os.environ.get('OV', '2')))
for outer in range(5):
    for c in seq * 3:
        r.render_frame_device(c, x, sync=True)
    for rep in range(2):
        d0 = r.frames_dropped()
        t0 = time.perf_counter()
        r.render_frame_device(seq[0], x); r.render_frame_device(seq[1], w); r.render_frame_device(seq[2], y); r.render_frame_device(seq[3], x)
        try: r.sync()
        except Exception as e: print("sync:", e)
        dt = time.perf_counter() - t0
        gx = r.device_download(x, H, W)
        m = gx != want[3]
        bad = np.argwhere(m)
        info = ""
        if len(bad):
            info = "rows %d-%d cols %d-%d, equals X1 there: %s, zero there: %s" % (bad[:, 0].min(), bad[:, 0].max(), bad[:, 1].min(), bad[:, 1].max(),
                                                                                 bool((gx[m] == want[0][m]).all()), bool((gx[m] == 0).all()))
        print("outer %d rep %d: %.3f ms, dropped %d, differing px vs X2 %d %s; W ok %s Y ok %s" % (outer, rep, dt * 1e3, r.frames_dropped() - d0, len(bad), info,
              np.array_equal(r.device_download(w, H, W), want[1]), np.array_equal(r.device_download(y, H, W), want[2])))
