#!/usr/bin/env python3
"""SPLAT_MODE_FAST against SPLAT_MODE_EXACT on one workload: frames/s (frames back to back, device-resident, clear
fused), the largest channel difference, the share of pixels that differ, early-out retries.  The start threshold of
the fast mode's transmittance scan is swept through SPLAT_EARLY_EPS (read at splat_create).
usage: python tools/fast_mode_probe.py [C1|C2|C3|C5] [eps ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import splat_amd
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
eps_list = [float(x) for x in sys.argv[2:]] or [1e-3]
n, W, H, seed = WORKLOADS[wl]
g = splat_amd.synthetic_scene(n, seed)
poses = []
for pos, yaw in (((0, 0, 5.0), 0.0), ((0, 0, 5.0), 1.2), ((0.3, 0.2, 0.4), 1.0)):
    cam = splat_amd.Camera(H, W, pos)
    if yaw: cam.update_yaw_angle(yaw)
    cam.update_camera_pose()
    poses.append(cam.to_c(0.01, 15))


done_cov = []


def run(mode, eps=None):
    if eps is None: os.environ.pop("SPLAT_EARLY_EPS", None)
    else: os.environ["SPLAT_EARLY_EPS"] = repr(eps)
    R = splat_amd.Renderer(mode=mode)
    if not done_cov: g.compute_cov3d(R); done_cov.append(1)
    R.upload(g)
    img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    frames, fb = [], []
    for p in poses:
        img.zero_()
        st = R.render_device(p, img.data_ptr(), sync=True, want_stats=True)
        frames.append(img.cpu().numpy().view(np.uint32).copy()); fb.append(int(st.n_fallback))
    for _ in range(3): R.render_device(poses[0], img.data_ptr(), sync=True)      # the sort grids follow the pose
    for _ in range(10): R.render_frame_device(poses[0], img.data_ptr())
    R.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): R.render_frame_device(poses[0], img.data_ptr())
    R.sync(); torch.cuda.synchronize()
    fps = 200 / (time.perf_counter() - t0)
    R.timing(reset=True)
    for _ in range(10): R.render_device(poses[0], img.data_ptr(), sync=True)
    ms, nfr = R.timing(reset=True)
    R.close()
    return frames, fps, fb, ms["composite"] / max(1, nfr)


def chans(a):
    return np.stack([(a >> s) & 0xff for s in (24, 16, 8, 0)]).astype(np.int32)


ex, fps0, fb0, k4 = run(splat_amd.MODE_EXACT)
print("%s exact: %.0f fps, compositor alone %.3f ms, retries %s" % (wl, fps0, k4, fb0))
for eps in eps_list:
    fa, fps, fb, k4 = run(splat_amd.MODE_FAST, eps)
    line = "%s fast eps %g: %.0f fps (x%.2f), compositor alone %.3f ms, retries %s;" % (wl, eps, fps, fps / fps0, k4, fb)
    for i, (a, b) in enumerate(zip(ex, fa)):
        d = np.abs(chans(a) - chans(b))
        line += " pose%d max diff %d (alpha %d), %.1f%% px differ;" % (i, d[1:].max(), d[0].max(), 100.0 * (d.max(0) > 0).mean())
    print(line)
