#!/usr/bin/env python3
"""K1 alone on the chip, for one or more builds (A/B libraries, e.g. experiment builds with -D switches): usage k1_ab.py [--wl C3] lib.so [lib.so ...]
(each build in its own process, SPLAT_PIPELINE=1, statistics frames -> HIP-event time of the preprocess launch)."""
import os, subprocess, sys
if len(sys.argv) >= 3 and sys.argv[1] == "--one":
    sys.path.insert(0, ".")
    import numpy as np, splat_amd
    from bench import WORKLOADS
    n, W, H, seed = WORKLOADS[sys.argv[2]]
    R = splat_amd.Renderer(); g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R); R.upload(g)
    cam = splat_amd.Camera(H, W, (0, 0, 5.0)); cam.update_camera_pose()
    img = np.zeros((H, W), np.uint32)
    t = []
    for k in range(24):
        st = R.render(cam.to_c(0.01, 15), img)
        t.append(st.ms_preprocess)
    t = np.array(t[4:])
    print("%-28s K1 min %.4f median %.4f max %.4f ms  (pairs %d)" % (os.path.basename(os.environ.get("SPLAT_AMD_LIB", "product")), t.min(), np.median(t), t.max(), st.n_pairs))
    R.close()
    sys.exit(0)
args = sys.argv[1:]
wl = "C3"
if args and args[0] == "--wl": wl = args[1]; args = args[2:]
for lib in args:
    env = dict(os.environ, SPLAT_PIPELINE="1")
    if lib != "product": env["SPLAT_AMD_LIB"] = os.path.abspath(lib)
    subprocess.call([sys.executable, __file__, "--one", wl], env=env)
