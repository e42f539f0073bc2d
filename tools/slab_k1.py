#!/usr/bin/env python3
"""K1 (+ scan, sort, compositor) of the slabs of an 8-way balanced partition, kernels alone (SPLAT_PIPELINE=1, a sync per frame).
usage: [SPLAT_AMD_LIB=...] python tools/slab_k1.py"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch, splat_amd
from splat_amd import dist as sdist
from bench import WORKLOADS
n, W, H, seed = WORKLOADS["C3"]
R = splat_amd.Renderer(); g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R)
cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose(); cam_c = cam.to_c(0.01, 15)
R.upload(g)
img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
loads = R.tile_row_loads(cam_c)
for s in sdist.slab_partition_balanced(loads, 8, row_overhead=2000.0) + [(0, 68)]:
    R.set_slab(*s)
    for _ in range(3):
        R.render_device(cam_c, img.data_ptr(), sync=True)
    R.timing(reset=True)
    for _ in range(20):
        R.render_device(cam_c, img.data_ptr(), sync=True)
    ms, frames = R.timing(reset=True)
    st = R.render_device(cam_c, img.data_ptr(), sync=True, want_stats=True)
    print("slab %-9s K1 %.4f scan %.4f sort %.4f K4 %.4f | culled %d of %d" % (s, ms["preprocess"] / frames, ms["scan"] / frames, ms["sort"] / frames, ms["composite"] / frames, st.n_blocks_culled, (n + 255) // 256))
