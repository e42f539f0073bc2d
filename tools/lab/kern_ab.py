#!/usr/bin/env python3
"""Per-kernel times alone on the chip (a sync after every frame) and the pipelined frame rate, for one or more builds
(A/B libraries built with -DSPLAT_EXP=...): usage kern_ab.py [--wl C3] [--slabs] product lib.so [lib.so ...]
Each build runs in its own process; `--slabs` adds K1 alone on every slab of the balanced 8-way partition."""
import os, subprocess, sys, time
if len(sys.argv) >= 3 and sys.argv[1] == "--one":
    sys.path.insert(0, ".")
    import numpy as np, torch, splat_amd
    from bench import WORKLOADS, ROW_OVERHEAD
    wl, slabs_too = sys.argv[2], sys.argv[3] == "1"
    n, W, H, seed = WORKLOADS[wl]
    R = splat_amd.Renderer(); g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R); R.upload(g)
    cam = splat_amd.Camera(H, W, (0, 0, 5.0)); cam.update_camera_pose(); cam_c = cam.to_c(0.01, 15)
    img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    st0 = R.render_frame_device(cam_c, img.data_ptr(), sync=True, want_stats=True)
    for _ in range(100):
        R.render_frame_device(cam_c, img.data_ptr())
    try:
        R.sync()
    except Exception as e:
        print("(setup frames: %s)" % e)
    torch.cuda.synchronize()
    rows = []
    for rep in range(3):
        R.timing(reset=True)
        for _ in range(30):
            R.render_frame_device(cam_c, img.data_ptr(), sync=True)
        ms, fr = R.timing(reset=True)
        iso = {k: v / fr for k, v in ms.items()}
        for _ in range(20):
            R.render_frame_device(cam_c, img.data_ptr())
        R.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            R.render_frame_device(cam_c, img.data_ptr())
        R.sync(); torch.cuda.synchronize()
        fps = 300 / (time.perf_counter() - t0)
        rows.append((fps, iso))
    name = os.path.basename(os.environ.get("SPLAT_AMD_LIB", "product"))
    print("%-22s %s near-selection tiles %d, repaired %d, early-out retries %d" % (name, wl, st0.n_near_tiles, st0.n_near_fallback, st0.n_fallback))
    for fps, iso in rows:
        print("%-22s %s %8.1f fps | alone: K1 %.4f scan %.4f sort %.4f K4 %.4f" % (name, wl, fps, iso["preprocess"], iso["scan"], iso["sort"], iso["composite"]))
    if slabs_too:
        loads = R.tile_row_loads(cam_c)
        out = []
        for s in splat_amd.slab_partition_native(loads, 8, ROW_OVERHEAD):
            R.set_slab(*s)
            for _ in range(5):
                R.render_frame_device(cam_c, img.data_ptr(), sync=True)
            R.timing(reset=True)
            for _ in range(30):
                R.render_frame_device(cam_c, img.data_ptr(), sync=True)
            ms, fr = R.timing(reset=True)
            t0 = time.perf_counter()
            for _ in range(200):
                R.render_frame_device(cam_c, img.data_ptr())
            R.sync(); torch.cuda.synchronize()
            out.append("%.3f/%.3f" % (ms["preprocess"] / fr, (time.perf_counter() - t0) / 200 * 1e3))
        print("%-22s %s 8 slabs, K1 alone / frame pipelined (ms): %s" % (name, wl, " ".join(out)))
    R.close()
    sys.exit(0)
args = sys.argv[1:]
wl, slabs = "C3", "0"
while args and args[0].startswith("--"):
    if args[0] == "--wl": wl = args[1]; args = args[2:]
    elif args[0] == "--slabs": slabs = "1"; args = args[1:]
for lib in args:
    env = dict(os.environ)
    if lib != "product": env["SPLAT_AMD_LIB"] = os.path.abspath(lib)
    subprocess.call([sys.executable, __file__, "--one", wl, slabs], env=env)
