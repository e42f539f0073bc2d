#!/usr/bin/env python3
"""Host-side cost of enqueueing one frame (no sync inside the loop) vs the GPU time of the frame."""
import sys, time
sys.path.insert(0, ".")
import torch
import splat_amd
from bench import WORKLOADS
for wl in ("C1", "C2", "C3"):
    n, W, H, seed = WORKLOADS[wl]
    R = splat_amd.Renderer()
    g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R)
    cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose()
    cam_c = cam.to_c(0.01, 15)
    R.upload(g)
    img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream(); R.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        R.render_device(cam_c, img.data_ptr(), sync=True)
        for _ in range(5): R.render_device(cam_c, img.data_ptr())
        R.sync(); torch.cuda.synchronize()
        K = 200
        t0 = time.perf_counter()
        for _ in range(K):
            R.render_device(cam_c, img.data_ptr())
        t1 = time.perf_counter()
        R.sync(); torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("%s: host enqueue %.1f us/frame, wall %.1f us/frame" % (wl, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
    R.close()
