#!/usr/bin/env python3
"""Scene-load timing at full size (reference: src/main.rs:18-26 = load_from_ply + compute_cov3d,
then the first render_to_buffer builds the pipeline): a synthetic PLY of the workload's size (C5: 6 M Gaussians =
1.49 GB on disk) is written to a scratch directory, then
  load      the C++ loader (mmap, decode + activations straight into the SoA buffers, sequential-f32 recentring),
            with ONE host thread and with all of them -- the result is the same bits either way
  cov3d     compute_cov3d on the GPU (K0, including its PCIe copies)
  upload    splat_upload_scene (host Morton order + block bounds, H2D, pack kernel)
  frame     the first frame (sizes the per-frame storage) and a steady-state frame
usage: python tools/load_probe.py [C5] [out.json]"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
import numpy as np
import splat_amd
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "C5"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
n, W, H, seed = WORKLOADS[wl]
res = {"workload": wl, "n": n, "host_threads": os.cpu_count()}
tmp = tempfile.mkdtemp(prefix="splat_load_")
path = os.path.join(tmp, "scene.ply")
t0 = time.perf_counter()
raw = splat_amd.gaussians.synthetic_raw(n, seed)
splat_amd.write_ply(path, raw, n)
del raw
res["write_s"] = time.perf_counter() - t0
res["file_bytes"] = os.path.getsize(path)

L = C.CDLL(os.path.join(os.path.dirname(splat_amd.__file__), "libsplat_host.so"))
L.splat_host_time_load.restype = C.c_double
L.splat_host_time_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_longlong), C.c_char_p, C.c_int]
cnt = C.c_longlong()
err = C.create_string_buffer(256)
for label, threads in (("load_all_threads_s", 0), ("load_1_thread_s", 1), ("load_all_threads_again_s", 0)):
    t = L.splat_host_time_load(path.encode(), threads, C.byref(cnt), err, 256)
    if t < 0:
        sys.exit("load failed: %s" % err.value.decode())
    res[label] = t
res["load_GBps_all_threads"] = res["file_bytes"] / res["load_all_threads_again_s"] / 1e9

t0 = time.perf_counter()
g = splat_amd.load_from_ply(path)                 # the Python binding of the same loader (+ one copy into numpy arrays)
res["python_load_from_ply_s"] = time.perf_counter() - t0
os.remove(path)
os.rmdir(tmp)
R = splat_amd.Renderer()
t0 = time.perf_counter()
g.compute_cov3d(R)
res["cov3d_gpu_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
R.upload(g)
res["upload_s"] = time.perf_counter() - t0
cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0))
cam.update_camera_pose()
img = np.zeros((H, W), np.uint32)
t0 = time.perf_counter()
R.render(cam.to_c(0.01, 15), img)
res["first_frame_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
st = R.render(cam.to_c(0.01, 15), img)
res["second_frame_s"] = time.perf_counter() - t0
res["n_pairs"] = int(st.n_pairs)
R.close()
res["start_to_first_frame_s"] = res["load_all_threads_again_s"] + res["cov3d_gpu_s"] + res["upload_s"] + res["first_frame_s"]
print(json.dumps(res))
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
