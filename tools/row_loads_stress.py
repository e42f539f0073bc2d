#!/usr/bin/env python3
"""splat_tile_row_loads under contention: k processes on one GPU, each asks for the C3 frame's per-tile-row pair counts again and
again (between frames, as bench.py's ranks do once) -- every answer must be the same.  usage: row_loads_stress.py [procs] [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, splat_amd
    from bench import WORKLOADS, make_scene
    rounds = int(sys.argv[2])
    n, W, H, seed = WORKLOADS["C3"]
    R = splat_amd.Renderer(); g = make_scene("C3"); g.compute_cov3d(R); R.upload(g)
    cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose(); c = cam.to_c(0.01, 15)
    img = torch.zeros((H, W), dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
    first, bad = None, 0
    for k in range(rounds):
        loads = np.array(R.tile_row_loads(c))
        if first is None: first = loads
        elif not np.array_equal(loads, first):
            bad += 1
            print("pid %d round %d: row loads differ: total %d vs %d, zero rows %d" % (os.getpid(), k, loads.sum(), first.sum(), int((loads == 0).sum())), flush=True)
        if k % 3 == 0:
            for _ in range(3): R.render_frame_device(c, img.data_ptr())
            R.sync()
    print("pid %d: total %d, %d of %d answers differ" % (os.getpid(), int(first.sum()), bad, rounds), flush=True)
    sys.exit(1 if bad or int(first.sum()) != 8025623 else 0)
procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(rounds)]) for _ in range(procs)]
rc = [p.wait() for p in ps]
print("exit codes", rc)
sys.exit(1 if any(rc) else 0)
