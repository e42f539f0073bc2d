#!/usr/bin/env python3
"""Per-kernel times (every kernel alone: a sync after each frame) of one workload at chosen poses, for one or more builds.
usage: pose_ab.py [--wl C3] [--poses bench,inside,inside2,near] product lib.so [lib.so ...]
poses: bench = Camera(0,0,5); inside / inside2 = the first two 'inside the cloud' poses of bench.py's uncorrelated set;
near = Camera(0,0,1)."""
import os, subprocess, sys
if len(sys.argv) >= 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ".")
    import numpy as np, torch, splat_amd
    from splat_amd import _lib
    from bench import WORKLOADS, make_scene
    wl, names = sys.argv[2], sys.argv[3].split(",")
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
    poses = {"bench": fixed((0, 0, 5.0)), "near": fixed((0, 0, 1.0)), "inside": rand[0], "inside2": rand[3], "inside3": rand[6]}
    name = os.path.basename(os.environ.get("SPLAT_AMD_LIB", "product"))
    R.set_option(_lib.OPT_TIMING_EVERY, 1)
    for p in names:
        c = poses[p]
        st = R.render_frame_device(c, img.data_ptr(), sync=True, want_stats=True)
        for _ in range(3):
            R.render_frame_device(c, img.data_ptr(), sync=True)
        R.timing(reset=True)
        for _ in range(10):
            R.render_frame_device(c, img.data_ptr(), sync=True)
        ms, fr = R.timing(reset=True)
        print("%-22s %s %-8s visible %8d pairs %9d maxlen %6d | alone: K1 %.4f scan %.4f sort %.4f K4 %.4f ms" %
              (name, wl, p, st.n_visible, st.n_pairs, st.max_tile_len, ms["preprocess"] / fr, ms["scan"] / fr, ms["sort"] / fr, ms["composite"] / fr))
    R.close()
    sys.exit(0)
args = sys.argv[1:]
wl, poses = "C3", "bench,inside,inside2"
while args and args[0].startswith("--"):
    if args[0] == "--wl": wl = args[1]; args = args[2:]
    elif args[0] == "--poses": poses = args[1]; args = args[2:]
for lib in args:
    env = dict(os.environ)
    if lib != "product": env["SPLAT_AMD_LIB"] = os.path.abspath(lib)
    subprocess.call([sys.executable, __file__, "--one", wl, poses], env=env)
