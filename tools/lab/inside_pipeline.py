#!/usr/bin/env python3
"""Pipelined frame rate and the library's own per-kernel averages (HIP events on every 8th frame) at a fixed pose, next to the
kernels alone (synchronous frames): where a pipelined frame from INSIDE the cloud spends its time.
usage: inside_pipeline.py [workload] [pose: bench|inside|inside2|near]  (environment variables select library / options)"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch, splat_amd
from splat_amd import _lib
from bench import WORKLOADS, make_scene
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
which = sys.argv[2] if len(sys.argv) > 2 else "inside"
n, W, H, seed = WORKLOADS[wl]
R = splat_amd.Renderer(); g = make_scene(wl); g.compute_cov3d(R); R.upload(g)
img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
rng = np.random.default_rng(36)
rand = []
for k in range(36):
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    radius = rng.uniform(0.2, 1.2) if k % 3 == 0 else rng.uniform(2.5, 7.0)
    cam = splat_amd.Camera(H, W, tuple(float(v) for v in d * radius))
    cam.update_yaw_angle(float(rng.uniform(0.0, 2.0 * np.pi))); cam.update_pitch_angle(float(rng.uniform(-0.6, 0.6)))
    cam.update_camera_pose()
    rand.append(cam.to_c(0.01, 15))
def fixed(pos):
    c = splat_amd.Camera(H, W, pos); c.update_camera_pose(); return c.to_c(0.01, 15)
pose = {"bench": fixed((0, 0, 5.0)), "near": fixed((0, 0, 1.0)), "inside": rand[0], "inside2": rand[3]}[which]
st = R.render_frame_device(pose, img.data_ptr(), sync=True, want_stats=True)
for rep in range(2):
    for _ in range(60):
        R.render_frame_device(pose, img.data_ptr())
    R.sync(); torch.cuda.synchronize()
    R.timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(400):
        R.render_frame_device(pose, img.data_ptr())
    R.sync(); torch.cuda.synchronize()
    fps = 400 / (time.perf_counter() - t0)
    ms, fr = R.timing(reset=True)
    pip = {k: v / max(fr, 1) for k, v in ms.items()}
    R.set_option(_lib.OPT_TIMING_EVERY, 1)
    for _ in range(3):
        R.render_frame_device(pose, img.data_ptr(), sync=True)
    R.timing(reset=True)
    for _ in range(10):
        R.render_frame_device(pose, img.data_ptr(), sync=True)
    ms, fr = R.timing(reset=True)
    iso = {k: v / max(fr, 1) for k, v in ms.items()}
    R.set_option(_lib.OPT_TIMING_EVERY, 8)
    print("%s %-7s pairs %9d maxlen %6d near tiles %5d | %7.1f frames/s = %.3f ms | pipelined: K1 %.3f scan %.3f select %.3f K4 %.3f (sum %.3f) | alone: K1 %.3f scan %.3f select %.3f K4 %.3f (sum %.3f)" % (
        wl, which, st.n_pairs, st.max_tile_len, st.n_near_tiles, fps, 1e3 / fps, pip["preprocess"], pip["scan"], pip["sort"], pip["composite"],
        pip["preprocess"] + pip["scan"] + pip["sort"] + pip["composite"], iso["preprocess"], iso["scan"], iso["sort"], iso["composite"],
        iso["preprocess"] + iso["scan"] + iso["sort"] + iso["composite"]))
R.close()
