"""Latency of the per-tile sort for ONE long list: squeeze n Gaussians into about one 16x16 tile and
read the sort's HIP-event time.  (python tools/sort_latency.py)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import splat_amd
from splat_amd.camera import Camera

r = splat_amd.Renderer()
for n in (1500, 3000, 6000, 10000, 16000):
    g = splat_amd.synthetic_scene(n, 17)
    g.positions[:, :2] *= 0.004          # x, y only: depths stay spread (no artificial ties)
    g.positions[:, 0] += 0.02
    g.positions[:, 1] += 0.02
    g.compute_cov3d(r)
    r.upload(g)
    cam = Camera(96, 96, (0.0, 0.0, 5.0))
    cam.update_camera_pose()
    img = np.zeros((96, 96), np.uint32)
    best = None
    for _ in range(8):
        st = r.render(cam.to_c(0.01), img)
        t = (st.ms_preprocess, st.ms_scan, st.ms_emit, st.ms_sort, st.ms_composite)
        best = t if best is None else tuple(min(a, b) for a, b in zip(best, t))
    print(n, "sort_fallback", st.n_sort_fallback, "max_tile_len", st.max_tile_len, "pairs", st.n_pairs, "ms pre/scan/emit/sort/comp", " ".join("%.4f" % x for x in best))
r.close()
