#!/usr/bin/env python3
"""Where a SYNCHRONOUS frame at a new pose spends its time -- the reference's only frame (src/main.rs:43-78: pose update,
clear, render_to_buffer, present).  Per pose of bench.py's 36 uncorrelated poses, of the 10-degree orbit, and of the jump into
the cloud: wall time of the call and the frame's own per-kernel device times (HIP events around the launches on their streams;
SPLAT_OPT_TIMING_EVERY = 1, which costs each frame ~25 us of event bubbles -- the wall column of the untimed pass is the one
to quote).  Columns:  K1 = preprocess (the binning pass proper), scan+redo = scan, and -- when the frame carries them -- the
overflow-redo launches (layout, K1 again, scan), select = near selection / sort launches, K4 = compositor,
rest = wall - sum: launch overhead, the count-first pass (K1's count flavour + layout, enqueued in front of the K1 event),
stream hand-overs, the wait.
usage: python tools/lab/sync_phases.py [C3|C5|C3s] [--host]     (--host: splat_render_frame into a page-locked host image)"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import splat_amd
from splat_amd import _lib
from bench import WORKLOADS, make_scene

wl = next((a for a in sys.argv[1:] if not a.startswith("--")), "C3")
host = "--host" in sys.argv
n, W, H, seed = WORKLOADS[wl]
g = make_scene(wl)
R = splat_amd.Renderer()
g.compute_cov3d(R)
R.upload(g)
image = torch.zeros((H, W), dtype=torch.int32, device="cuda")
himg = R.host_image(H, W) if host else None


def frame(cam_c):
    t0 = time.perf_counter()
    if host:
        R.render_frame(cam_c, himg)
    else:
        R.render_frame_device(cam_c, image.data_ptr(), sync=True)
    return (time.perf_counter() - t0) * 1e3


def random_poses():
    rng = np.random.default_rng(36)
    poses = []
    for k in range(36):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        radius = rng.uniform(0.2, 1.2) if k % 3 == 0 else rng.uniform(2.5, 7.0)
        cam = splat_amd.Camera(H, W, tuple(float(v) for v in d * radius))
        cam.update_yaw_angle(float(rng.uniform(0.0, 2.0 * np.pi)))
        cam.update_pitch_angle(float(rng.uniform(-0.6, 0.6)))
        cam.update_camera_pose()
        poses.append((cam.to_c(0.01, 15), "inside" if k % 3 == 0 else "outside"))
    return [poses[k] for k in rng.permutation(36)]


def orbit_poses():
    cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0))
    out = []
    for _ in range(36):
        cam.update_camera_pose()
        out.append((cam.to_c(0.01, 15), "orbit"))
        cam.update_yaw_angle(10.0 * np.pi / 180.0)
    return out


def run(name, poses):
    # pass 1: untimed (the wall time to quote); pass 2: every frame carries its events
    R.set_option(_lib.OPT_TIMING_EVERY, 1000000)
    for c, _ in poses[:2]:
        frame(c)
    d0 = R.frames_dropped()
    wall = [frame(c) for c, _ in poses]
    redone = R.frames_dropped() - d0
    R.set_option(_lib.OPT_TIMING_EVERY, 1)
    rows = []
    for c, kind in poses:
        R.timing(reset=True)
        w = frame(c)
        ms, fr = R.timing(reset=True)
        st = None
        rows.append((kind, w, ms, fr))
    print("%s %s%s: %d poses, untimed wall mean %.3f median %.3f max %.3f ms (%.0f frames/s), frames redone inside their call %d" %
          (wl, name, " host-visible" if host else "", len(poses), np.mean(wall), np.median(wall), np.max(wall), 1e3 / np.mean(wall), redone))
    print("  %-8s %8s %8s %10s %8s %8s %8s   (timed pass, ms)" % ("pose", "wall", "K1", "scan+redo", "select", "K4", "rest"))
    agg = {}
    for kind, w, ms, fr in rows:
        fr = max(fr, 1)
        k1, sc, se, k4 = ms["preprocess"] / fr, ms["scan"] / fr, (ms["sort"] + ms["emit"]) / fr, ms["composite"] / fr
        agg.setdefault(kind, []).append((w, k1, sc, se, k4, w - (k1 + sc + se + k4)))
    for kind, v in agg.items():
        a = np.array(v)
        for label, f in (("mean", np.mean), ("max", np.max)):
            r = f(a, axis=0)
            print("  %-8s %8.3f %8.3f %10.3f %8.3f %8.3f %8.3f   %s of %d" % (kind, r[0], r[1], r[2], r[3], r[4], r[5], label, len(v)))


frame(orbit_poses()[0][0])
run("orbit, 10 degrees a frame", orbit_poses())
run("uncorrelated poses", random_poses())
# the jump: 40 asynchronous frames at the bench pose, then one synchronous frame from inside the cloud
R.set_option(_lib.OPT_TIMING_EVERY, 1000000)
bench_pose = orbit_poses()[0][0]
inside = next(c for c, k in random_poses() if k == "inside")
for rep in range(3):
    for _ in range(40):
        R.render_frame_device(bench_pose, image.data_ptr())
    R.sync()
    torch.cuda.synchronize()
    t_in = frame(inside)
    t_again = frame(inside)
    t_back = frame(bench_pose)
    print("%s jump %d: into the cloud %.3f ms, same pose again %.3f, back to the bench pose %.3f; device bytes %.2f GB" %
          (wl, rep, t_in, t_again, t_back, R.device_bytes()[0] / 1e9 if hasattr(R, "device_bytes") else -1))
R.close()
