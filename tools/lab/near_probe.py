#!/usr/bin/env python3
"""Near selection at several poses and selection sizes: tiles served, tiles the repair launch had to take, synchronous frame
time.  usage: near_probe.py [workload ...]"""
import math, os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch, splat_amd
from splat_amd import _lib as L
from bench import WORKLOADS, make_scene
POSES = [("bench", (0.0, 0.0, 5.0), 0.0, 0.0), ("orbit70", (0.0, 0.0, 5.0), math.radians(70), 0.0), ("orbit250", (0.0, 0.0, 5.0), math.radians(250), 0.0),
         ("close", (0.0, 0.0, 2.0), 0.4, 0.1), ("inside", (0.3, 0.2, 0.4), 1.0, -0.2)]
for wl in (sys.argv[1:] or ["C3"]):
    n, W, H, seed = WORKLOADS[wl]
    R = splat_amd.Renderer(); g = make_scene(wl); g.compute_cov3d(R); R.upload(g)
    img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    for name, pos, yaw, pitch in POSES:
        cam = splat_amd.Camera(H, W, pos)
        if yaw: cam.update_yaw_angle(yaw)
        if pitch: cam.update_pitch_angle(pitch)
        cam.update_camera_pose(); cam_c = cam.to_c(0.01, 15)
        row = []
        for cap in (0, 2048, 1024):
            R.set_option(L.OPT_NEAR_SELECT_KEYS, cap)
            for _ in range(3):
                st = R.render_frame_device(cam_c, img.data_ptr(), sync=True, want_stats=True)
            t0 = time.perf_counter()
            for _ in range(20):
                R.render_frame_device(cam_c, img.data_ptr(), sync=True)
            ms = (time.perf_counter() - t0) / 20 * 1e3
            for _ in range(20):
                R.render_frame_device(cam_c, img.data_ptr())
            R.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                R.render_frame_device(cam_c, img.data_ptr())
            R.sync(); torch.cuda.synchronize()
            pms = (time.perf_counter() - t0) / 200 * 1e3
            row.append("%d: %d/%d %.3f %.3f" % (cap, st.n_near_fallback, st.n_near_tiles, ms, pms))
        print("%s %-8s pairs %9d longest %6d | cap: repaired/served, sync ms, pipelined ms | %s" % (wl, name, st.n_pairs, st.max_tile_len, " | ".join(row)))
    R.close()
