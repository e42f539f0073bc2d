#!/usr/bin/env python3
"""Smooth camera motion (yaw steps of 1, 3, 10 degrees a frame, asynchronous device-resident frames): frames/s and frames
dropped with near selection on / off, and the repairs of a few synchronous statistics frames along the path.
usage: motion_probe.py [workload ...]"""
import math, sys, time
sys.path.insert(0, ".")
import numpy as np, torch, splat_amd
from splat_amd import _lib as L
from bench import WORKLOADS, make_scene
for wl in (sys.argv[1:] or ["C3s"]):
    n, W, H, seed = WORKLOADS[wl]
    R = splat_amd.Renderer(); g = make_scene(wl); g.compute_cov3d(R); R.upload(g)
    img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    for step in (1.0, 3.0, 10.0):
        cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose()
        poses = []
        for k in range(240):
            poses.append(cam.to_c(0.01, 15))
            cam.update_yaw_angle(math.radians(step)); cam.update_camera_pose()
        for cap in (0, 2048):
            R.set_option(L.OPT_NEAR_SELECT_KEYS, cap)
            for k in range(40):
                R.render_frame_device(poses[k], img.data_ptr())
            try: R.sync()
            except Exception: pass
            d0 = R.frames_dropped()
            t0 = time.perf_counter()
            for k in range(40, 240):
                R.render_frame_device(poses[k], img.data_ptr())
            try: R.sync()
            except Exception: pass
            torch.cuda.synchronize()
            fps = 200 / (time.perf_counter() - t0)
            rep = []
            for k in range(0, 48, 4):
                st = R.render_frame_device(poses[k], img.data_ptr(), sync=True, want_stats=True)
                rep.append(int(st.n_near_fallback))
            print("%s yaw %4.1f deg/frame, near %4d: %6.0f frames/s, %3d of 200 dropped; repairs in statistics frames four poses apart: %s" % (wl, step, cap, fps, R.frames_dropped() - d0, rep))
    R.close()
