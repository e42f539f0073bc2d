#!/usr/bin/env python3
"""Per-kernel times of a workload's poses, kernels alone (a sync after every frame) and the pipelined frame rate.
usage: [SPLAT_AMD_LIB=...] python tools/pose_breakdown.py [workload=C3s]
For C3s the poses are bench.py's (the bench pose, and the one inside the scene where a Gaussian covers ~30 tiles)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, splat_amd
from bench import WORKLOADS, make_scene

wl = sys.argv[1] if len(sys.argv) > 1 else "C3s"
n, W, H, seed = WORKLOADS[wl]
g = make_scene(wl, via_ply=False)
R = splat_amd.Renderer()
g.compute_cov3d(R); R.upload(g)
img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
poses = [("bench_pose", (0.0, 0.0, 5.0), 0.0)]
if wl in ("C3s",):
    poses.append(("inside_pose", (0.3, 0.2, 0.4), 1.0))
for name, pos, yaw in poses:
    cam = splat_amd.Camera(H, W, pos)
    if yaw:
        cam.update_yaw_angle(yaw)
    cam.update_camera_pose()
    cam_c = cam.to_c(0.01, 15)
    for _ in range(4):
        R.render_frame_device(cam_c, img.data_ptr(), sync=True)
    R.timing(reset=True)
    for _ in range(20):
        R.render_frame_device(cam_c, img.data_ptr(), sync=True)
    ms, frames = R.timing(reset=True)
    st = R.render_device(cam_c, img.data_ptr(), sync=True, want_stats=True)
    for _ in range(30):
        R.render_frame_device(cam_c, img.data_ptr())
    R.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        R.render_frame_device(cam_c, img.data_ptr())
    R.sync()
    fps = 100 / (time.perf_counter() - t0)
    print("%s %-11s alone: K1 %.4f scan %.4f sort %.4f K4 %.4f ms | pipelined %.0f fps (%.3f ms) | visible %d pairs %d (%.1f per visible) max list %d, "
          "compositor iterations scan %d blend %d, dropped %d" %
          (wl, name, ms["preprocess"] / frames, ms["scan"] / frames, ms["sort"] / frames, ms["composite"] / frames, fps, 1e3 / fps,
           st.n_visible, st.n_pairs, st.n_pairs / max(1, st.n_visible), st.max_tile_len,
           st.n_iter_scan, st.n_iter_blend, R.frames_dropped()))
