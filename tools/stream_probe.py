#!/usr/bin/env python3
"""Host-visible frame rate (pixels in host memory): splat_render (host image in and out, synchronous)
vs splat_render_stream (cleared device image, asynchronous copy-out, two pinned frames in flight)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import splat_amd
from bench import WORKLOADS

for wl in ("C1", "C2", "C3"):
    n, W, H, seed = WORKLOADS[wl]
    R = splat_amd.Renderer()
    g = splat_amd.synthetic_scene(n, seed); g.compute_cov3d(R)
    cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0)); cam.update_camera_pose()
    cam_c = cam.to_c(0.01, 15)
    R.upload(g)
    himg = np.zeros((H, W), np.uint32)
    R.render(cam_c, himg)
    t0 = time.perf_counter()
    for _ in range(20):
        himg[:] = 0
        R.render(cam_c, himg)
    t_sync = (time.perf_counter() - t0) / 20
    bufs = [R.host_image(H, W), R.host_image(H, W)]
    R.render_stream(cam_c, bufs[0]); R.stream_wait(bufs[0])
    assert np.array_equal(bufs[0], himg)
    res = {}
    for mode in ("two in flight", "one in flight"):
        K = 100
        t0 = time.perf_counter()
        for k in range(K):
            R.render_stream(cam_c, bufs[k & 1])
            if mode == "two in flight" and k:
                R.stream_wait(bufs[(k - 1) & 1])
            if mode == "one in flight":
                R.stream_wait(bufs[k & 1])
        R.stream_wait(bufs[(K - 1) & 1])
        res[mode] = (time.perf_counter() - t0) / K
    print("%s: host-visible frame: splat_render %.3f ms (%.0f fps); splat_render_stream %.3f ms (%.0f fps) with two "
          "frames in flight, %.3f ms with one" % (wl, t_sync * 1e3, 1 / t_sync, res["two in flight"] * 1e3,
                                                  1 / res["two in flight"], res["one in flight"] * 1e3))
    R.close()
