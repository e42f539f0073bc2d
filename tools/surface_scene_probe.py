#!/usr/bin/env python3
"""A scene that looks more like a trained one than the uniform synthetic cloud: flat, anisotropic Gaussians on thin
surfaces (a sphere shell, a ground plane, a wall), so that a tile sees hundreds of splats within a sliver of depth.
Checks parity against the oracle and prints the counters that would show a performance cliff (sort fallbacks,
early-out retries, longest list, binning mode).   usage: python tools/surface_scene_probe.py [n] [W H]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import splat_amd
from oracle import oracle as O
from helpers import scene_dict, oracle_camera, image_diff

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (960, 540)
rng = np.random.default_rng(11)
g = splat_amd.synthetic_scene(n, 23)
k = n // 2
d = rng.normal(size=(k, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
g.positions[:k, :3] = (2.0 + rng.normal(0, 0.002, (k, 1))) * d                      # sphere shell, 2 mm thick
m = n // 4
g.positions[k:k + m, 0] = rng.uniform(-4, 4, m); g.positions[k:k + m, 2] = rng.uniform(-4, 4, m)
g.positions[k:k + m, 1] = 1.5 + rng.normal(0, 0.001, m)                              # ground plane (up is -y)
g.positions[k + m:, 0] = rng.uniform(-4, 4, n - k - m); g.positions[k + m:, 1] = rng.uniform(-3, 1.5, n - k - m)
g.positions[k + m:, 2] = -3.0 + rng.normal(0, 0.001, n - k - m)                      # back wall
g.scales[:, 0] = np.exp(rng.normal(-3.3, 0.5, n)); g.scales[:, 1] = np.exp(rng.normal(-3.3, 0.5, n))
g.scales[:, 2] = np.exp(rng.normal(-7.0, 0.3, n))                                    # flat
g.opacities[:] = 1.0 / (1.0 + np.exp(-rng.normal(2.0, 1.5, n)))                     # mostly opaque
R = splat_amd.Renderer()
g.compute_cov3d(R); R.upload(g)
for pos, yaw in (((0.0, 0.0, 5.0), 0.0), ((0.5, -0.5, 3.0), 0.6)):
    cam = splat_amd.Camera(H, W, pos)
    if yaw: cam.update_yaw_angle(yaw)
    cam.update_camera_pose()
    img = np.zeros((H, W), np.uint32)
    st = R.render(cam.to_c(0.01, 15), img)
    t0 = time.perf_counter()
    for _ in range(20): st = R.render(cam.to_c(0.01, 15), img * 0)
    ms = (time.perf_counter() - t0) / 20 * 1e3
    ref, ost = O.render(scene_dict(g), oracle_camera(cam, 0.01), nthreads=64)
    mx, cnt = image_diff(img, ref)
    print("pose", pos, yaw, "pairs", st.n_pairs, "== oracle", st.n_pairs == ost.n_tile_pairs, "longest list", st.max_tile_len,
          "sort fallbacks", st.n_sort_fallback, "early-out retries", st.n_fallback, "binning", R.binning_mode() if hasattr(R, "binning_mode") else "?",
          "| max diff", mx, "px differing", cnt, "| kernels ms: K1 %.3f sort %.3f K4 %.3f, host-visible frame %.2f ms" % (st.ms_preprocess, st.ms_sort, st.ms_composite, ms))
