#!/usr/bin/env python3
"""One leg of tools/motion_probe.py on its own (for rocprofv3 --kernel-trace --stats): `frames` asynchronous device-resident
frames of a yaw orbit at `step` degrees a frame.  usage: motion_trace.py [workload] [step_deg] [frames] [near_cap]"""
import math, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, splat_amd
from splat_amd import _lib as L
from bench import WORKLOADS, make_scene
wl = sys.argv[1] if len(sys.argv) > 1 else "C3s"
step = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
n, W, H, seed = WORKLOADS[wl]
R = splat_amd.Renderer(); g = make_scene(wl); g.compute_cov3d(R); R.upload(g)
R.set_option(L.OPT_NEAR_SELECT_KEYS, cap)
img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose()
poses = []
for k in range(40 + frames):
    poses.append(cam.to_c(0.01, 15))
    cam.update_yaw_angle(math.radians(step)); cam.update_camera_pose()
for k in range(40):
    R.render_frame_device(poses[k], img.data_ptr())
try: R.sync()
except Exception as e: print('warm-up:', e)
d0 = R.frames_dropped()
t0 = time.perf_counter()
for k in range(40, 40 + frames):
    R.render_frame_device(poses[k], img.data_ptr())
try: R.sync()
except Exception as e: print('timed:', e)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s yaw %.1f deg/frame near %d: %.0f frames/s (%.3f ms), %d of %d dropped" % (wl, step, cap, frames / dt, 1e3 * dt / frames, R.frames_dropped() - d0, frames))
R.close()
