#!/usr/bin/env python3
"""Smooth camera motion (yaw steps of 1, 3, 10 degrees a frame, asynchronous device-resident frames): frames/s and frames
dropped with near selection on / off, and the repairs of a few synchronous statistics frames along the path.
usage: motion_probe.py [--steps 1,3,10] [--caps 0,2048] [workload ...]
Per leg also the library's own per-kernel averages (HIP events on every 8th frame) and the repairs / redone frames seen."""
import math, sys, time
sys.path.insert(0, ".")
import numpy as np, torch, splat_amd
from splat_amd import _lib as L
from bench import WORKLOADS, make_scene
argv = sys.argv[1:]
STEPS, CAPS = (1.0, 3.0, 10.0), (0, 2048)
while argv and argv[0].startswith("--"):
    if argv[0] == "--steps": STEPS = tuple(float(x) for x in argv[1].split(","))
    elif argv[0] == "--caps": CAPS = tuple(int(x) for x in argv[1].split(","))
    argv = argv[2:]
for wl in (argv or ["C3s"]):
    n, W, H, seed = WORKLOADS[wl]
    R = splat_amd.Renderer(); g = make_scene(wl); g.compute_cov3d(R); R.upload(g)
    img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    for step in STEPS:
        cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose()
        poses = []
        for k in range(240):
            poses.append(cam.to_c(0.01, 15))
            cam.update_yaw_angle(math.radians(step)); cam.update_camera_pose()
        for cap in CAPS:
            R.set_option(L.OPT_NEAR_SELECT_KEYS, cap)
            for k in range(40):
                R.render_frame_device(poses[k], img.data_ptr())
            try: R.sync()
            except Exception: pass
            d0 = R.frames_dropped()
            R.timing(reset=True)
            t0 = time.perf_counter()
            for k in range(40, 240):
                R.render_frame_device(poses[k], img.data_ptr())
            try: R.sync()
            except Exception: pass
            torch.cuda.synchronize()
            fps = 200 / (time.perf_counter() - t0)
            ms, fr = R.timing(reset=True)
            kern = " ".join("%s %.3f" % (k[:4], v / max(fr, 1)) for k, v in ms.items() if k != "status")
            rep = []
            for k in range(0, 48, 4):
                st = R.render_frame_device(poses[k], img.data_ptr(), sync=True, want_stats=True)
                rep.append(int(st.n_near_fallback))
            print("%s yaw %4.1f deg/frame, near %4d: %6.0f frames/s, %3d of 200 dropped; repairs in statistics frames four poses apart: %s; kernel ms: %s; keys/slot %d, device peak %.2f GB" % (wl, step, cap, fps, R.frames_dropped() - d0, rep, kern, R.binning_mode(), R.device_bytes()[1] / 1e9))
    R.close()
