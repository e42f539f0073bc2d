#!/usr/bin/env python3
"""Single-process check: rendering the frame slab by slab == rendering it at once (device images)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import splat_amd
from splat_amd import dist as sdist
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
n, W, H, seed = WORKLOADS[wl]
R = splat_amd.Renderer()
g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R)
cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose()
cam_c = cam.to_c(0.01, 15)
R.upload(g)
full = torch.zeros((H, W), dtype=torch.int32, device="cuda")
R.render_device(cam_c, full.data_ptr(), sync=True)
loads = R.tile_row_loads(cam_c)
for world in (2, 3, 4, 8):
    for name, slabs in (("equal", sdist.slab_partition(H, world)), ("balanced", sdist.slab_partition_balanced(loads, world, 2000.0))):
        parts = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        for s in slabs:
            R.set_slab(*s)
            R.render_device(cam_c, parts.data_ptr(), sync=True)
        R.set_slab(0, -1)
        d = (parts != full)
        rows = torch.nonzero(d.any(dim=1)).flatten().tolist()
        print(world, name, slabs, "equal" if not d.any() else "DIFF px=%d rows=%s" % (int(d.sum()), rows[:10]))
