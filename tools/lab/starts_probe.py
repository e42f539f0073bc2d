#!/usr/bin/env python3
"""How much of every tile list does the compositor's exact walk need?  SPLAT_DBG_STARTS=1 makes a statistics frame record
(list length, nearest keys needed) per wave; the library prints the totals per list-length class.  usage: starts_probe.py [workload ...]"""
import os, sys
os.environ["SPLAT_DBG_STARTS"] = "1"
sys.path.insert(0, ".")
import numpy as np, splat_amd
from bench import WORKLOADS, make_scene
for wl in (sys.argv[1:] or ["C3"]):
    n, W, H, seed = WORKLOADS[wl]
    R = splat_amd.Renderer(); g = make_scene(wl); g.compute_cov3d(R); R.upload(g)
    for name, pos, yaw in (("bench pose", (0.0, 0.0, 5.0), 0.0), ("inside", (0.3, 0.2, 0.4), 1.0)):
        cam = splat_amd.Camera(H, W, pos)
        if yaw: cam.update_yaw_angle(yaw)
        cam.update_camera_pose()
        img = np.zeros((H, W), np.uint32)
        R.render(cam.to_c(0.01, 15), img)
        sys.stderr.write("== %s, %s\n" % (wl, name)); sys.stderr.flush()
        st = R.render(cam.to_c(0.01, 15), img)
        sys.stderr.write("   pairs %d, longest list %d\n" % (st.n_pairs, st.max_tile_len))
    R.close()
