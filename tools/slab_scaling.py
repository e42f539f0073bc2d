#!/usr/bin/env python3
"""Predict multi-GPU strong scaling on ONE GPU: render every slab of a k-way partition separately and
take the slowest slab's GPU time (+ nothing for the gather, which is ~20-40 us over xGMI).
Usage: python tools/slab_scaling.py [workload] [fast] [swap]
  fast: SPLAT_MODE_FAST, every colour byte within 1 of the exact frame
  swap: every rank renders into a two-image swap chain (splat_set_frame_overlap(2), as bench.py's ranks do over RCCL): the
        compositors of consecutive frames share the chip, and the partition weighs a tile row's fixed cost accordingly"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import splat_amd
from splat_amd import dist as sdist
from bench import WORKLOADS

from bench import ROW_OVERHEAD, ROW_OVERHEAD_SWAP
wl = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] not in ("fast", "swap") else "C3"
n, W, H, seed = WORKLOADS[wl]
SWAP = "swap" in sys.argv[1:]
R = splat_amd.Renderer(mode=splat_amd.MODE_FAST if "fast" in sys.argv[1:] else 0)
g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R)
cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose()
cam_c = cam.to_c(0.01, 15)
R.upload(g)
img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
imgs = [img, torch.zeros((H, W), dtype=torch.int32, device="cuda")] if SWAP else [img]
if SWAP:
    R.set_frame_overlap(2)
loads = R.tile_row_loads(cam_c)
print("row loads: total %d, max row %d" % (loads.sum(), loads.max()))


def slab_time(slab, reps=60):
    """wall time per frame of this slab alone on the GPU, frames enqueued back to back (they overlap
    on the device exactly as in bench.py)"""
    import time
    R.set_slab(*slab)
    for _ in range(3):
        R.render_device(cam_c, img.data_ptr(), sync=True)
    for k in range(6):
        R.render_frame_device(cam_c, imgs[k % len(imgs)].data_ptr())
    R.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        R.render_frame_device(cam_c, imgs[k % len(imgs)].data_ptr())      # clear + render, what a rank's viewer-loop frame is
    R.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for world in (1, 2, 4, 8):
    for name, slabs in (("equal", sdist.slab_partition(H, world)),
                        ("balanced", sdist.slab_partition_balanced(loads, world, row_overhead=ROW_OVERHEAD_SWAP if SWAP else ROW_OVERHEAD))):
        t = [slab_time(s) for s in slabs]
        print("world %d %-8s rows %s  slab ms %s  -> max %.3f ms = %.0f fps (one rank's frames back to back; gather not included)" %
              (world, name, [b - a for a, b in slabs], ["%.2f" % x for x in t], max(t), 1000.0 / max(t)))
        if world == 1:
            break

# per-kernel breakdown of the heaviest 8-way slab
slabs = sdist.slab_partition_balanced(loads, 8, row_overhead=ROW_OVERHEAD_SWAP if SWAP else ROW_OVERHEAD)
for s in (slabs[0], slabs[3]):
    R.set_slab(*s)
    for _ in range(3):
        R.render_device(cam_c, img.data_ptr(), sync=True)
    R.timing(reset=True)
    for k in range(10):
        R.render_frame_device(cam_c, imgs[k % len(imgs)].data_ptr())
    ms, frames = R.timing(reset=True)
    st = R.render_device(cam_c, img.data_ptr(), sync=True, want_stats=True)
    print("slab", s, {k: round(v / frames, 4) for k, v in ms.items()},
          "K1 blocks culled %d of %d, visible %d, pairs %d" % (st.n_blocks_culled, (n + 255) // 256, st.n_visible, st.n_pairs))
