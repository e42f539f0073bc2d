#!/usr/bin/env python3
"""The 36-pose yaw orbit (10 degrees a frame, src/main.rs:53-60), asynchronous device-resident frames: frames/s with and
without near selection, repairs per frame.  usage: orbit_probe.py [workload ...]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch, splat_amd
from splat_amd import _lib as L
from bench import WORKLOADS, make_scene, orbit_poses
for wl in (sys.argv[1:] or ["C3"]):
    n, W, H, seed = WORKLOADS[wl]
    R = splat_amd.Renderer(); g = make_scene(wl); g.compute_cov3d(R); R.upload(g)
    cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose()
    pipe = splat_amd.GaussianSplatPipeline01(g, cam, renderer=R)
    poses = orbit_poses(pipe, cam)
    img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    for cap in (0, 2048):
        R.set_option(L.OPT_NEAR_SELECT_KEYS, cap)
        for k in range(36):
            R.render_frame_device(poses[k], img.data_ptr())
        try: R.sync()
        except Exception as e: print("(", e, ")")
        d0 = R.frames_dropped()
        t0 = time.perf_counter()
        for k in range(108):
            R.render_frame_device(poses[k % 36], img.data_ptr())
        try: R.sync()
        except Exception as e: print("(", e, ")")
        torch.cuda.synchronize()
        fps = 108 / (time.perf_counter() - t0)
        rep = []
        for k in range(12):       # synchronous statistics frames along the orbit: repairs with hints one pose old
            st = R.render_frame_device(poses[k * 3 % 36], img.data_ptr(), sync=True, want_stats=True)
            rep.append(int(st.n_near_fallback))
        print("%s orbit, near %4d: %.0f frames/s, %d dropped; repairs in 12 statistics frames three poses apart: %s" % (wl, cap, fps, R.frames_dropped() - d0, rep))
    R.close()
