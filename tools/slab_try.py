#!/usr/bin/env python3
"""Time hand-picked tile-row partitions of C3 on one GPU (each slab's frames back to back, as tools/slab_scaling.py does).
usage: [SLAB_SWAP=1] python tools/slab_try.py 16,7,6,5,5,6,7,16 [14,8,6,6,6,6,8,14 ...]   (SLAB_SWAP=1: two images in turn, splat_set_frame_overlap(2))"""
import sys, time
sys.path.insert(0, ".")
import torch, splat_amd
from bench import WORKLOADS
n, W, H, seed = WORKLOADS["C3"]
R = splat_amd.Renderer(); g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R)
cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose(); cam_c = cam.to_c(0.01, 15)
R.upload(g)
import os
img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
n_imgs = int(os.environ.get("SLAB_SWAP", "0") or 0)
n_imgs = 2 if n_imgs == 1 else n_imgs          # SLAB_SWAP=1: two images; SLAB_SWAP=n: n images in rotation
imgs = [img] + [torch.zeros((H, W), dtype=torch.int32, device="cuda") for _ in range(max(0, n_imgs - 1))]
if len(imgs) >= 2 and os.environ.get("SLAB_NO_OVERLAP") != "1":
    R.set_frame_overlap(2)        # a swap chain: consecutive frames composite side by side


def slab_time(slab, reps=60):
    R.set_slab(*slab)
    for _ in range(3):
        R.render_device(cam_c, img.data_ptr(), sync=True)
    for k in range(8):
        R.render_frame_device(cam_c, imgs[k % len(imgs)].data_ptr())
    R.sync()
    t0 = time.perf_counter()
    for k in range(reps):
        R.render_frame_device(cam_c, imgs[k % len(imgs)].data_ptr())
    R.sync()
    return (time.perf_counter() - t0) / reps * 1e3


loads = None
for spec in sys.argv[1:]:
    if spec.startswith("auto:"):          # auto:<ranks>:<row_overhead> -- the library's balanced partition for that overhead
        from splat_amd import dist as sdist
        if loads is None:
            R.set_slab(0, (H + 15) // 16)
            loads = R.tile_row_loads(cam_c)
        _, k, ov = spec.split(":")
        rows = [b - a for a, b in sdist.slab_partition_balanced(loads, int(k), row_overhead=float(ov))]
    else:
        rows = [int(x) for x in spec.split(",")]
    assert sum(rows) == (H + 15) // 16, (sum(rows), (H + 15) // 16)
    slabs, a = [], 0
    for r in rows:
        slabs.append((a, a + r)); a += r
    t = [slab_time(s) for s in slabs]
    print("rows %s slab ms %s -> max %.3f ms = %.0f fps" % (rows, ["%.3f" % x for x in t], max(t), 1e3 / max(t)))
