#!/usr/bin/env python3
"""Full-size parity sweep: the HIP frame against the CPU oracle for several camera poses of a
workload (orbit poses, near and inside the cloud).  Prints one line per pose and a JSON summary.
usage: python tools/parity_sweep.py [C3] [out.json] [libm]     (libm: SPLAT_MODE_LIBM_EXP -- the frames must then be the oracle's bit for bit)"""
import json, math, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import splat_amd
from oracle import oracle as O
from helpers import scene_dict, oracle_camera, image_diff
from bench import WORKLOADS, make_scene

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
out = sys.argv[2] if len(sys.argv) > 2 else None
n, W, H, seed = WORKLOADS[wl]
R = splat_amd.Renderer(mode=splat_amd.MODE_LIBM_EXP if (len(sys.argv) > 3 and sys.argv[3] == "libm") else 0)
g = make_scene(wl, via_ply=False); g.compute_cov3d(R)        # (C3s: the trained-like surface scene)
R.upload(g)
sd = scene_dict(g)
poses = [((0, 0, 5.0), 0.0, 0.0), ((0, 0, 5.0), math.radians(70), 0.0), ((0, 0, 5.0), math.radians(160), 0.0),
         ((0, 0, 5.0), math.radians(250), 0.3), ((0, 0, 2.5), 0.0, 0.0), ((0.3, 0.2, 0.4), 1.0, -0.2)]
rows = []
for pos, yaw, pitch in poses:
    cam = splat_amd.Camera(H, W, pos)
    if yaw: cam.update_yaw_angle(yaw)
    if pitch: cam.update_pitch_angle(pitch)
    cam.update_camera_pose()
    img = np.zeros((H, W), np.uint32)
    st = R.render(cam.to_c(0.01, 15), img)
    # ... and the frame of a camera AT REST at this pose (the sixth in a row: start hints instead of the early-out's scan, the
    # selection sized tightly and in one pass) -- it must be the same frame, byte for byte
    rest = R.device_image(np.zeros((H, W), np.uint32))
    for _ in range(5):
        R.render_frame_device(cam.to_c(0.01, 15), rest)
    st_rest = R.render_frame_device(cam.to_c(0.01, 15), rest, sync=True, want_stats=True)
    at_rest = R.device_download(rest, H, W)
    R.device_free(rest)
    t0 = time.time()
    ref, ost = O.render(sd, oracle_camera(cam, 0.01), nthreads=os.cpu_count() or 8)
    mx, cnt = image_diff(img, ref)
    row = dict(pos=pos, yaw=round(yaw, 3), pitch=pitch, n_visible=int(st.n_visible), n_pairs=int(st.n_pairs),
               pairs_equal=bool(st.n_pairs == ost.n_tile_pairs and st.n_visible == ost.n_visible),
               max_tile_len=int(st.max_tile_len), binning_mode=R.binning_mode(), near_selection_tiles=int(st.n_near_tiles),
               near_selection_repaired=int(st.n_near_fallback), max_channel_diff_lsb=int(mx),
               pixels_differing=int(cnt), pixels=W * H, oracle_s=round(time.time() - t0, 2),
               frame_at_rest_equals_first_frame=bool(np.array_equal(at_rest, img)), scan_iterations_first_frame=int(st.n_iter_scan),
               scan_iterations_at_rest=int(st_rest.n_iter_scan), near_selection_repaired_at_rest=int(st_rest.n_near_fallback))
    rows.append(row)
    print(row, flush=True)
R.close()
summary = dict(workload=wl, poses=rows, worst_lsb=max(r["max_channel_diff_lsb"] for r in rows),
               worst_pixels=max(r["pixels_differing"] for r in rows), frames_at_rest_equal=all(r["frame_at_rest_equals_first_frame"] for r in rows))
print(json.dumps(summary))
if out:
    json.dump(summary, open(out, "w"), indent=1)
