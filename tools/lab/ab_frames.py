#!/usr/bin/env python3
"""Are two builds' frames the same bytes?  usage: ab_frames.py libA.so libB.so  (renders C1/C2/C3 poses with each, in
separate processes, and compares the images)"""
import os, subprocess, sys, tempfile
import numpy as np
if len(sys.argv) == 4 and sys.argv[1] == "--render":
    sys.path.insert(0, ".")
    import splat_amd
    from bench import WORKLOADS
    out = {}
    for wl, poses in (("C1", [((0, 0, 5.0), 0.0)]), ("C2", [((0, 0, 5.0), 0.0), ((0.3, 0.2, 0.4), 1.0)]),
                      ("C3", [((0, 0, 5.0), 0.0), ((0, 0, 5.0), 1.2), ((0.3, 0.2, 0.4), 1.0)])):
        n, W, H, seed = WORKLOADS[wl]
        R = splat_amd.Renderer(); g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R); R.upload(g)
        for k, (pos, yaw) in enumerate(poses):
            cam = splat_amd.Camera(H, W, pos)
            if yaw: cam.update_yaw_angle(yaw)
            cam.update_camera_pose()
            rng = np.random.default_rng(7)
            img = rng.integers(0, 2**32, (H, W), dtype=np.uint64).astype(np.uint32)
            R.render(cam.to_c(0.01, 15), img)
            out["%s_%d" % (wl, k)] = img
        R.close()
    np.savez(sys.argv[3], **out)
    sys.exit(0)
a, b = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp()
for lib, name in ((a, "a"), (b, "b")):
    env = dict(os.environ, SPLAT_AMD_LIB=lib)
    subprocess.check_call([sys.executable, __file__, "--render", lib, os.path.join(tmp, name + ".npz")], env=env)
A, B = np.load(os.path.join(tmp, "a.npz")), np.load(os.path.join(tmp, "b.npz"))
ok = True
for k in A.files:
    same = np.array_equal(A[k], B[k])
    ok &= same
    print(k, "identical" if same else "DIFFER: %d pixels" % int((A[k] != B[k]).sum()))
sys.exit(0 if ok else 1)
