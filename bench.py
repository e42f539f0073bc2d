#!/usr/bin/env python3
"""bench.py -- frames/s of the 3DGS raster hot path on MI355X (BASELINE.json metric).

A step = one frame: clear the device image + render_to_buffer's whole path (preprocess, binning,
per-tile sort, exact compositing) with scene and image resident in HBM -- the region
src/main.rs:71-75 times (splat_render_frame_device: the clear is fused into the compositor, which then does
not read the old pixels and zeroes the tiles nothing covers).  N=1 workload: C3, the 1.5M-Gaussian 'truck' stand-in at 1920x1080
(synthetic, seed 3: no real PLY ships with the reference).

N>1 ("scaling": "strong"): the same frame split into load-balanced tile-row slabs, one per GPU, with ONE
gather of slab rows to rank 0 per frame -- a grouped ncclSend / ncclRecv issued by the C ABI itself
(splat_comm_gather, include/splat_hip.h "Multi-GPU").  Launched as one process per GPU
(python -m torch.distributed.run ...): torch.distributed only carries the 128-byte RCCL id and the barrier.
`--single-process` runs the same decomposition from ONE process (splat_multi_*: a host thread and a
context per device, ncclCommInitAll).

Prints ONE JSON line on rank 0; exits non-zero if the frame is not the oracle's frame (> 1 LSB, or a
different (Gaussian, tile) pair count) or if the gathered frame differs from the single-GPU frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (n_gaussians, width, height, seed)       BASELINE.md section 3
    "C1": (10_000, 256, 256, 1),
    "C2": (281_498, 1280, 720, 2),
    "C3": (1_500_000, 1920, 1080, 3),
    "C5": (6_000_000, 3840, 2160, 5),
    # not a BASELINE config: the trained-like stand-in (flat anisotropic Gaussians on surfaces, heavy-tailed scales,
    # opacities near 0 / 1 -- splat_amd.synthetic_surface_raw) at C3's size; a parity case and an extra bench leg
    "C3s": (1_500_000, 1920, 1080, 13),
}
SURFACE_WORKLOADS = ("C3s",)


def make_scene(wl, via_ply=False):
    """The seeded scene of a workload.  via_ply: written as an INRIA PLY (62 floats per vertex) and read back through the
    C++ host mirror's load_from_ply -- the route SURVEY section 8(d) specifies (activations + recentring exercised)."""
    import tempfile
    import splat_amd
    n, _, _, seed = WORKLOADS[wl]
    raw_of = splat_amd.synthetic_surface_raw if wl in SURFACE_WORKLOADS else splat_amd.synthetic_raw
    if not via_ply:
        return (splat_amd.synthetic_surface_scene if wl in SURFACE_WORKLOADS else splat_amd.synthetic_scene)(n, seed)
    d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(d, "splat_bench_%s_%d.ply" % (wl, os.getpid()))
    try:
        splat_amd.write_ply(path, raw_of(n, seed), n)
        return splat_amd.load_from_ply(path)
    finally:
        if os.path.exists(path):
            os.remove(path)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ROW_OVERHEAD = 2000.0     # slab balancing: fixed cost of a tile row, in (Gaussian, tile) pairs
ROW_OVERHEAD_SWAP = 100000.0   # ... when consecutive frames composite side by side (two-image swap chain): tools/slab_try.py auto:8:<overhead>


def cpu_baseline(g, cam_c, threads):
    """The oracle (a restatement -- the Rust binary cannot be built here) timed on this box's host
    cores on ONE frame of the same workload.  Checker used as a baseline only; never as product."""
    from oracle import oracle as O
    oc = O.Camera()
    for f, _ in O.Camera._fields_:
        v = getattr(cam_c, f)
        if hasattr(v, "__len__"):
            getattr(oc, f)[:] = list(v)
        else:
            setattr(oc, f, v)
    scene = dict(pos4=g.positions, cov3d=g.cov3d, opacity=g.opacities, sh=g.sh)
    t0 = time.perf_counter()
    img, st = O.render(scene, oc, nthreads=threads)
    dt = time.perf_counter() - t0
    return img, st, dt


def orbit_poses(pipe, cam):
    poses = []
    for _ in range(36):
        poses.append(pipe.camera_constants())
        cam.update_yaw_angle(10.0 * np.pi / 180.0)        # Key::Right, src/main.rs:57-60
        cam.update_camera_pose()
    return poses


def single_process(args, n, W, H, seed):
    """One process, one host thread + context per device (splat_multi_*)."""
    import splat_amd
    t_setup = time.perf_counter()
    R0 = splat_amd.Renderer(device=0)
    g = splat_amd.synthetic_scene(n, seed)
    g.compute_cov3d(R0)
    cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0))
    cam.update_camera_pose()
    pipe = splat_amd.GaussianSplatPipeline01(g, cam, renderer=R0)
    cam_c = pipe.camera_constants()
    poses = orbit_poses(pipe, cam) if args.orbit else [cam_c]
    R0.upload(g)
    full = np.zeros((H, W), np.uint32)
    st = R0.render(cam_c, full)
    R0.close()
    # SPLAT_BENCH_SHARE_GPU=1 (testing only): every rank on device 0 (copy transport instead of RCCL)
    M = splat_amd.MultiRenderer([0] * args.gpus if os.environ.get("SPLAT_BENCH_SHARE_GPU") == "1" else list(range(args.gpus)))
    M.upload(g)
    swap_chain = os.environ.get("SPLAT_BENCH_SWAP_CHAIN", "1") != "0"
    if swap_chain:
        M.set_frame_overlap(2)        # two slab images per device in turn: consecutive frames composite side by side
    slabs = M.balance(cam_c)
    for k in range(SETTLE_FRAMES):                        # setup (see main): the devices out of their idle clocks
        M.render_frame(poses[k % len(poses)])
    M.sync()
    for k in range(args.warmup):
        M.render_frame(poses[k % len(poses)])
    M.sync()
    for r in range(args.gpus):
        M.rank_timing(r, reset=True)
    dropped0 = [M.rank_frames_dropped(r) for r in range(args.gpus)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        M.render_frame(poses[k % len(poses)])
    M.sync()
    dt = time.perf_counter() - t0
    kern = [M.rank_timing(r, reset=True) for r in range(args.gpus)]
    dropped = [M.rank_frames_dropped(r) - dropped0[r] for r in range(args.gpus)]     # frames a rank's device skipped inside the timed region
    M.render_frame(cam_c)
    M.sync()
    same = bool(np.array_equal(M.download(H, W), full))
    M.close()
    per = {k: max(ms[k] / max(fr, 1) for ms, fr in kern) for k in kern[0][0]}
    per_rank = [dict({k: v / max(fr, 1) for k, v in ms.items()}, rank=r, tile_rows=slabs[r][1] - slabs[r][0], frames_dropped=int(dropped[r]))
                for r, (ms, fr) in enumerate(kern)]
    t_frame = dt / args.steps * 1e3
    out = {
        "metric": "frames_per_sec", "value": args.steps / dt, "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "setup_frames_before_warmup": SETTLE_FRAMES, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d Gaussians @%dx%d, synthetic seed %d, Camera(0,0,5), Pipeline01 (lowpass 0.01, "
                               "sh_dim 15), exact mode" % (args.workload, n, W, H, seed),
                   "camera": "36-pose yaw orbit" if args.orbit else "fixed pose",
                   "partition": "ONE process, %d host threads + contexts (splat_multi_*), load-balanced tile-row slabs %s, "
                                "grouped ncclSend/ncclRecv to rank 0%s" % (args.gpus, [b - a for a, b in slabs],
                                                                         "; two images per device in turn (splat_multi_set_frame_overlap(2))" if swap_chain else ""),
                   "frame_overlap": 2 if swap_chain else 1,
                   "n_pairs": int(st.n_pairs), "n_visible": int(st.n_visible)},
        "roofline_frame": {"bytes_algorithmic": int(st.bytes_algorithmic), "t_frame_ms": t_frame,
                           "achieved": int(st.bytes_algorithmic) / (t_frame * 1e-3) / 1e9, "peak": HBM_PEAK_GBS * args.gpus, "unit": "GB/s",
                           "frac": int(st.bytes_algorithmic) / (t_frame * 1e-3) / 1e9 / (HBM_PEAK_GBS * args.gpus),
                           "note": "the FRAME's algorithmic bytes (one copy of the scene read, not one per rank) / wall time per frame, "
                                   "against the ranks' HBM peaks together"},
        "kernel_ms_slowest_rank": per,
        "kernel_ms_per_rank": per_rank,
        "frames_dropped": int(sum(dropped)),
        "multi_gpu_frame_equals_single_gpu_frame": same,
        "setup_s": time.perf_counter() - t_setup - dt,
    }
    print(json.dumps(out))
    sys.exit(3 if not same else (4 if sum(dropped) else 0))


SETTLE_FRAMES = 200     # see the timed loop


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it (no WORLD_SIZE): re-execute this very command line as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...`
    -- one rank per GPU over RCCL, the same processes the driver's own torchrun line starts.  A box with fewer GPUs than
    N runs only with SPLAT_BENCH_SHARE_GPU=1 (every rank on device 0, gloo: the decomposition, not a measurement)."""
    import socket
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("SPLAT_BENCH_SHARE_GPU") != "1":
        sys.exit("bench.py --gpus %d: this box has %d GPU(s) (SPLAT_BENCH_SHARE_GPU=1 runs the %d ranks on device 0 "
                 "over gloo, to exercise the decomposition)" % (args.gpus, have, args.gpus))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def surface_leg(R0, image, stream, device, args, counted, leg_drops):
    import torch
    import splat_amd
    n, W, H, seed = WORKLOADS["C3s"]
    t0 = time.perf_counter()
    g = make_scene("C3s", via_ply=True)
    t_load = time.perf_counter() - t0
    R = splat_amd.Renderer(device=device)
    out = {"workload": "C3s: %d Gaussians @%dx%d, synthetic surfaces seed %d, written as a PLY and loaded by load_from_ply (%.1f s)"
                       % (n, W, H, seed, t_load)}
    try:
        g.compute_cov3d(R)
        R.upload(g)
        R.set_stream(stream.cuda_stream)
        for name, pos, yaw in (("bench_pose", (0.0, 0.0, 5.0), 0.0), ("inside_pose", (0.3, 0.2, 0.4), 1.0)):
            cam = splat_amd.Camera(H, W, pos)
            if yaw:
                cam.update_yaw_angle(yaw)
            cam.update_camera_pose()
            cam_c = cam.to_c(0.01, 15)
            with torch.cuda.stream(stream):
                image.zero_()
                st = R.render_device(cam_c, image.data_ptr(), sync=True, want_stats=True)
            img = image.cpu().numpy().view(np.uint32).copy()
            for _ in range(30):
                R.render_frame_device(cam_c, image.data_ptr())
            torch.cuda.synchronize()
            with counted("c3s_" + name, R):
                t1 = time.perf_counter()
                for _ in range(100):
                    R.render_frame_device(cam_c, image.data_ptr())
                torch.cuda.synchronize()
                fps = 100 / (time.perf_counter() - t1)
            leg = {"frames_per_sec": fps, "frames_dropped": leg_drops["c3s_" + name], "n_visible": int(st.n_visible), "n_pairs": int(st.n_pairs),
                   "max_tile_len": int(st.max_tile_len), "early_out_fallback_waves": int(st.n_fallback),
                   "sort_fallback_tiles": int(st.n_sort_fallback), "key_buffer_entries_per_slot": int(R.binning_mode()),
                   "device_bytes_peak": int(R.device_bytes()[1])}
            if not args.no_cpu_baseline:
                ref, ost, cdt = cpu_baseline(g, cam_c, args.cpu_threads or (os.cpu_count() or 1))
                d = np.abs(np.stack([((img >> s) & 255).astype(np.int32) - ((ref >> s) & 255).astype(np.int32) for s in (0, 8, 16, 24)]))
                leg["parity"] = {"max_channel_diff_lsb": int(d.max()), "pixels_differing": int((d.max(0) > 0).sum()),
                                 "pairs_equal": bool(int(st.n_pairs) == int(ost.n_tile_pairs) and int(st.n_visible) == int(ost.n_visible)),
                                 "oracle_frames_per_sec": 1.0 / cdt}
            out[name] = leg
        # ... and in MOTION: yaw steps of 3 and 10 degrees a frame from the bench pose, 100 frames each,
        # device resident and asynchronous; every frame must be rendered (a skipped one costs nothing)
        for deg in (3.0, 10.0):
            cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0))
            poses = []
            for k in range(120):
                cam.update_camera_pose()
                poses.append(cam.to_c(0.01, 15))
                cam.update_yaw_angle(deg * np.pi / 180.0)
            for k in range(20):
                R.render_frame_device(poses[k], image.data_ptr())
            torch.cuda.synchronize()
            name = "c3s_orbit_%g_deg_per_frame" % deg
            with counted(name, R):
                t1 = time.perf_counter()
                for k in range(20, 120):
                    R.render_frame_device(poses[k], image.data_ptr())
                torch.cuda.synchronize()
                fps = 100 / (time.perf_counter() - t1)
            out["orbit_%g_deg_per_frame" % deg] = {"frames_per_sec": fps, "frames_dropped": leg_drops[name]}
    finally:
        R.close()
    return out


def live_pmc_traffic(args):
    """HBM bytes per launch of every kernel of THIS workload on THIS box: two short rocprofv3 passes of bench.py itself
    (--pmc FETCH_SIZE, then --pmc WRITE_SIZE -- separate passes, no trace domain beside them, as MI355X_MICROARCH.md's
    HBM section prescribes), corrected as that section says (both counters are KiB; gfx950's FETCH_SIZE tallies the
    128-B requests of a wide coalesced stream at 64 B: doubled; calibration in this repo's access pattern:
    profiles/README.md).  Returns ({kernel: bytes}, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 is not on this box"
    work = tempfile.mkdtemp(prefix="splat_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    sub = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--settle", "5", "--no-cpu-baseline",
           "--no-extra-legs", "--no-live-pmc", "--workload", args.workload, "--mode", args.mode]
    if args.scene:
        sub += ["--scene", os.path.abspath(args.scene)]
    if args.orbit:
        sub += ["--orbit"]
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            try:
                r = subprocess.run([exe, "--pmc", counter, "--kernel-include-regex", "splat", "--output-format", "csv", "-d", d, "--"] + sub,
                                   cwd=work, env=env, capture_output=True, text=True, timeout=420)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s timed out" % counter
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s: rc %d, %d counter files: %s" % (counter, r.returncode, len(files), (r.stderr or "")[-300:])
            acc = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name", counter) != counter:
                        continue
                    name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("splat::", "").strip()
                    acc.setdefault(name, []).append(float(row["Counter_Value"]))
            for name, vals in acc.items():
                per.setdefault(name, {})[counter] = (sum(vals) / len(vals), len(vals))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out, detail = {}, {}
    for name, c in per.items():
        f, nf = c.get("FETCH_SIZE", (0.0, 0))
        w, nw = c.get("WRITE_SIZE", (0.0, 0))
        out[name] = int(f * 1024.0 * 2.0 + w * 1024.0)
        detail[name] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w, "read_bytes_corrected": int(f * 2048.0),
                        "write_bytes": int(w * 1024.0), "launches_averaged": [nf, nw]}
    if args.pmc_json:
        json.dump({"workload": args.workload, "bytes_per_launch": out, "detail": detail}, open(args.pmc_json, "w"), indent=1, sort_keys=True)
    return out, detail


def cold_and_incoherent_legs(g, W, H, cam_c, device, stream, mode):
    """What the headline never shows (it runs one pose after 200 settled frames): a context's FIRST frame, frames at
    uncorrelated poses, and the first frame after the camera jumps.  One-pass binning sizes every tile's key region
    from the lists of earlier frames; a frame that outgrows them is skipped on the device and -- synchronous frames,
    as here -- redone inside the call with rebuilt regions, so these legs pay for every redo (`frames_redone`).
    Reference loop: src/main.rs:43-78 (one synchronous render_to_buffer per pose change)."""
    import torch
    import splat_amd
    out = {}
    R = splat_amd.Renderer(device=device, mode=mode)
    try:
        image = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        R.upload(g)
        R.set_stream(stream.cuda_stream)
        # (1) first frame after splat_upload_scene: key buffers are allocated, every slot counts its pairs first
        t0 = time.perf_counter()
        R.render_frame_device(cam_c, image.data_ptr(), sync=True)
        out["first_frame_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        R.render_frame_device(cam_c, image.data_ptr(), sync=True)
        out["second_frame_ms"] = (time.perf_counter() - t0) * 1e3
        # (2) 36 uncorrelated poses, a third of them inside the cloud, each a synchronous frame
        rng = np.random.default_rng(36)
        poses = []
        for k in range(36):
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            radius = rng.uniform(0.2, 1.2) if k % 3 == 0 else rng.uniform(2.5, 7.0)
            cam = splat_amd.Camera(H, W, tuple(float(v) for v in d * radius))
            cam.update_yaw_angle(float(rng.uniform(0.0, 2.0 * np.pi)))
            cam.update_pitch_angle(float(rng.uniform(-0.6, 0.6)))
            cam.update_camera_pose()
            poses.append(cam.to_c(0.01, 15))
        order = rng.permutation(36)
        d0 = R.frames_dropped()
        per = []
        t0 = time.perf_counter()
        for k in order:
            t1 = time.perf_counter()
            R.render_frame_device(poses[k], image.data_ptr(), sync=True)
            per.append((time.perf_counter() - t1) * 1e3)
        dt = time.perf_counter() - t0
        out["random_pose_sync_fps"] = 36 / dt
        out["random_pose_sync"] = {"frames": 36, "frames_redone": int(R.frames_dropped() - d0), "ms_mean": float(np.mean(per)),
                                   "ms_median": float(np.median(per)), "ms_max": float(np.max(per)),
                                   "what": "36 poses in random order: positions uniform in direction, radius 0.2-1.2 (every third pose: "
                                           "inside the cloud) or 2.5-7, random yaw and pitch; every frame synchronous "
                                           "(clear + render + wait), so a frame the device skipped is redone inside its call"}
        # (3) steady state at the bench pose, then the camera jumps (to the first inside pose, and back)
        for _ in range(40):
            R.render_frame_device(cam_c, image.data_ptr())
        try:
            R.sync()
        except Exception:
            pass
        torch.cuda.synchronize()
        d0 = R.frames_dropped()
        inside = poses[0]
        t0 = time.perf_counter()
        R.render_frame_device(inside, image.data_ptr(), sync=True)
        out["pose_jump_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        R.render_frame_device(inside, image.data_ptr(), sync=True)
        after = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        R.render_frame_device(cam_c, image.data_ptr(), sync=True)
        back = (time.perf_counter() - t0) * 1e3
        out["pose_jump"] = {"to_inside_ms": out["pose_jump_ms"], "same_pose_again_ms": after, "back_to_bench_pose_ms": back,
                            "frames_redone": int(R.frames_dropped() - d0),
                            "what": "40 asynchronous frames at the bench pose, then ONE synchronous frame from inside the cloud "
                                    "(radius %.2f): wall time until the complete frame" % float(np.linalg.norm(list(inside.cam_pos)))}
    finally:
        R.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--orbit", action="store_true",
                    help="step the camera yaw by 10 degrees every frame (the 36-pose orbit of src/main.rs:53-60) "
                         "instead of the fixed pose")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from one process (splat_multi_*: one host thread per device) instead of one rank per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the host-visible and orbit legs")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--mode", default="exact", choices=["exact", "fast"],
                    help="exact (default, what `value` is quoted on): the reference's frame; fast: SPLAT_MODE_FAST, every colour "
                         "byte within 1 of the exact frame's by construction")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the two short rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this workload after the timed "
                         "region; roofline.traffic then comes from the committed profiles/traffic.json, labelled so")
    ap.add_argument("--pmc-json", default=None, help="also write the live counter passes' per-kernel bytes to this file")
    ap.add_argument("--settle", type=int, default=SETTLE_FRAMES, help=argparse.SUPPRESS)
    ap.add_argument("--scene", default=None,
                    help="an INRIA 3DGS .ply (e.g. the real 'truck' / 'bicycle' / plush_sledge scene: none is in the repo, "
                         "src/main.rs:21 loads one): rendered at --workload's resolution and camera instead of the synthetic "
                         "stand-in of the same size; load_from_ply's activations and recentring included (src/gaussians.rs:375-405)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import splat_amd
    from splat_amd import dist as sdist

    n, W, H, seed = WORKLOADS[args.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.single_process and world == 1 and args.gpus > 1:
        return single_process(args, n, W, H, seed)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            return self_launch(args)          # plain `python bench.py --gpus N`: become the torchrun launch of N ranks
        args.gpus = world
    # SPLAT_BENCH_SHARE_GPU=1 (testing only): all ranks on cuda:0 over gloo, to exercise the N>1 code
    # path on a single-GPU box (RCCL refuses ranks that share a device); the real run is one rank per GPU over RCCL
    share = os.environ.get("SPLAT_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    # per-kernel HIP events ride on every 8th asynchronous frame by default (each record is a queue
    # bubble); a short timed region takes them on every 4th to have a few launches to average over (on every
    # frame they cost the 20-step run 4 % of its frame rate)
    os.environ.setdefault("SPLAT_TIMING_EVERY", "8" if args.steps >= 64 else "4")
    main_mode = splat_amd.MODE_FAST if args.mode == "fast" else splat_amd.MODE_EXACT
    R = splat_amd.Renderer(device=local, mode=main_mode)
    if args.scene:
        g = splat_amd.load_from_ply(args.scene)          # the C++ host mirror's loader (mmap + direct decode)
        n = len(g.opacities)
        data_kind, scene_name = "ply:" + os.path.basename(args.scene), "%s (%d Gaussians)" % (os.path.basename(args.scene), n)
    else:
        g = make_scene(args.workload)
        data_kind, scene_name = "synthetic", ("synthetic surfaces seed %d" if args.workload in SURFACE_WORKLOADS else "synthetic seed %d") % seed
    g.compute_cov3d(R)                                   # K0 on the GPU (load-time, not timed)
    cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0))        # src/main.rs:13,29
    cam.update_camera_pose()
    pipe = splat_amd.GaussianSplatPipeline01(g, cam, renderer=R)   # what the default binary uses
    cam_c = pipe.camera_constants()
    orbit = orbit_poses(pipe, cam)
    poses = orbit if args.orbit else [cam_c]
    R.upload(g)

    gather_kind = None
    swap_chain = False
    if world > 1:
        row_loads = R.tile_row_loads(cam_c)
        slabs = sdist.slab_partition(H, world)               # (placeholder until the transport is known: it decides the balance)
        # (SPLAT_BENCH_TRY_NATIVE=1 on the shared-GPU test path: attempt the native communicator anyway -- RCCL refuses ranks
        # that share a device -- so that the fallback and the reason it names are exercised)
        if not share or os.environ.get("SPLAT_BENCH_TRY_NATIVE") == "1":
            try:        # data plane: the C ABI's own communicator (RCCL); torch.distributed carries the id only
                box = [splat_amd.Renderer.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0, device=None if share else torch.device("cuda", local))
                R.comm_init(box[0], world, rank)
                gather_kind = "native"
            except Exception as e:      # keep the run alive on the other transport, and say so
                sys.stderr.write("bench.py rank %d: native RCCL gather unavailable (%s); using torch.distributed\n" % (rank, e))
        flag = torch.tensor([1 if gather_kind == "native" else 0], device="cpu" if share else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        # balanced contiguous slabs from per-tile-row pair counts; every rank derives the same partition.  Over RCCL the
        # ranks render into a two-image swap chain (splat_set_frame_overlap(2): the compositor of frame N+1 runs beside the
        # one of frame N and its gather, so a slab's frame costs its work rather than its densest tile's latency) and a
        # tile row's fixed cost weighs accordingly.
        if int(flag.item()) == 0:
            gather_kind = "torch.distributed"
            slabs = splat_amd.slab_partition_native(row_loads, world, ROW_OVERHEAD)
            R.set_slab(*slabs[rank])
        else:
            swap_chain = os.environ.get("SPLAT_BENCH_SWAP_CHAIN", "1") != "0"
            slabs = splat_amd.slab_partition_native(row_loads, world, ROW_OVERHEAD_SWAP if swap_chain else ROW_OVERHEAD)
            R.comm_set_slabs(slabs)
    else:
        slabs = sdist.slab_partition(H, world)
        R.set_slab(*slabs[rank])
    stream = torch.cuda.Stream()
    R.set_stream(stream.cuda_stream)
    image = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    images = [image]
    if swap_chain:
        images.append(torch.zeros((H, W), dtype=torch.int32, device="cuda"))
        R.set_frame_overlap(2)
    torch.cuda.synchronize()           # (the fills run on torch's default stream, the frames on `stream`)

    frame_no = [0]

    def step():
        pose = poses[frame_no[0] % len(poses)]
        image = images[frame_no[0] % len(images)]
        frame_no[0] += 1
        with torch.cuda.stream(stream):
            # color.clear(0) + render_to_buffer (src/main.rs:73-74): one call, the clear fused into the compositor
            R.render_frame_device(pose, image.data_ptr())        # enqueue only
            if gather_kind == "native":
                R.comm_gather(image.data_ptr(), W, H, 0)         # grouped ncclSend / ncclRecv behind the frame, on its stream
            elif world > 1:
                if share:
                    stream.synchronize()     # gloo's CUDA receive is not ordered after this stream's work (NCCL's is)
                sdist.gather_slabs(image, slabs, rank)

    from splat_amd.renderer import SplatError
    from splat_amd import _lib as _abi

    def settle(r):
        """splat_sync: every frame's status is harvested, so splat_frames_dropped() is final.  A frame that outgrew storage
        sized from earlier frames is SKIPPED on the device (it costs ~0 ms and leaves the image alone: with a fixed pose
        the parity check cannot see it) -- the loss surfaces here once as SPLAT_ERR_CAPACITY; the legs count them."""
        try:
            r.sync()
        except SplatError as e:
            if e.code != _abi.ERR_CAPACITY:
                raise

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        settle(R)

    leg_drops = {}

    class counted:
        """with counted(name): ... -- frames the device skipped inside a timed leg (must be 0 for the fps to mean anything)"""
        def __init__(self, name, r=None):
            self.name, self.r = name, r or R
        def __enter__(self):
            settle(self.r)
            self.d0 = self.r.frames_dropped()
        def __exit__(self, *a):
            settle(self.r)
            leg_drops[self.name] = leg_drops.get(self.name, 0) + self.r.frames_dropped() - self.d0

    # one synchronous frame first: settles the pair-buffer capacity and gives the frame's statistics
    with torch.cuda.stream(stream):
        image.zero_()
        st = R.render_device(cam_c, image.data_ptr(), sync=True, want_stats=True)
    # setup, untimed and not counted as warm-up steps: the same frames until the device is out of its idle clocks and
    # every lazily created queue exists (a GPU fresh from idle runs its first few dozen frames ~10 % slower)
    for _ in range(0 if share else args.settle):         # (not on the CPU-transport test path: nothing to settle there)
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    R.timing(reset=True)
    dropped0 = R.frames_dropped()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()                                              # barrier + device fence + splat_sync (statuses harvested)
    dt = time.perf_counter() - t0
    dropped_timed = R.frames_dropped() - dropped0        # frames the device skipped inside the timed region: must be 0
    kern_ms, frames = R.timing(reset=True)               # HIP events on the kernels' own stream
    dev_peak = R.device_bytes()[1]                       # device memory the context has held at most (scene, frame slots, key buffers)
    key_entries = R.binning_mode()
    last_pose = poses[(frame_no[0] - 1) % len(poses)]
    final = images[(frame_no[0] - 1) % len(images)].clone()
    # the compositor's (wave, record) iteration counts of a frame LIKE THE TIMED ONES (the first frame above scanned for its
    # walks' starts; the frames of a camera at rest start where the previous frame did: SPLAT_OPT_START_HINTS)
    with torch.cuda.stream(stream):
        st_steady = R.render_frame_device(last_pose, images[(frame_no[0] - 1) % len(images)].data_ptr(), sync=True, want_stats=True)
    R.timing(reset=True)
    # The context overlaps the binning + sort of frame N+1 (its own stream) with the compositor of
    # frame N, so the durations above include time shared with a neighbouring frame's kernels.  For
    # reference, a few frames with a sync after each (nothing overlaps): every kernel alone on the chip.
    iso_ms, iso_frames = None, 0
    legs = {}
    if world == 1:
        with torch.cuda.stream(stream):
            for k in range(10):
                image.zero_()
                R.render_device(poses[k % len(poses)], image.data_ptr(), sync=True)
        iso_ms, iso_frames = R.timing(reset=True)
        if not args.no_extra_legs:
            K = 72
            # (1) the 36-pose yaw orbit of src/main.rs:53-60, device resident like `value`
            if not args.orbit:
                for k in range(4):
                    R.render_frame_device(orbit[k], image.data_ptr())
                torch.cuda.synchronize()
                with counted("orbit_36_poses_device_resident_fps"):
                    t1 = time.perf_counter()
                    for k in range(K):
                        R.render_frame_device(orbit[k % 36], image.data_ptr())
                    torch.cuda.synchronize()
                    legs["orbit_36_poses_device_resident_fps"] = K / (time.perf_counter() - t1)
            # (1b) a short timed region (the driver's --steps 20) is mostly pipeline fill and a GPU coming out of idle:
            # the same step, 400 frames back to back, for the steady state
            if args.steps < 100:
                for k in range(10):
                    R.render_frame_device(cam_c, image.data_ptr())
                torch.cuda.synchronize()
                with counted("steady_state_400_frames_device_resident_fps"):
                    t1 = time.perf_counter()
                    for k in range(400):
                        R.render_frame_device(cam_c, image.data_ptr())
                    torch.cuda.synchronize()
                    legs["steady_state_400_frames_device_resident_fps"] = 400 / (time.perf_counter() - t1)
            # (1d) `value`'s step with the start hints off (SPLAT_OPT_START_HINTS 0): with a camera at rest the compositor's
            # exact walks start where the previous frame's did instead of scanning their lists for the place -- this is the
            # same fixed pose with the scan in every frame, as every frame of a fast-moving camera has it
            from splat_amd import _lib as _L
            hints_before = R.get_option(_L.OPT_START_HINTS)
            R.set_option(_L.OPT_START_HINTS, 0)
            for k in range(10):
                R.render_frame_device(cam_c, image.data_ptr())
            torch.cuda.synchronize()
            with counted("fixed_pose_scanning_every_frame_fps"):
                t1 = time.perf_counter()
                for k in range(200):
                    R.render_frame_device(cam_c, image.data_ptr())
                torch.cuda.synchronize()
                legs["fixed_pose_scanning_every_frame_fps"] = 200 / (time.perf_counter() - t1)
            # (the exact modes render the same bytes wherever a walk starts; SPLAT_MODE_FAST is within 1 of the exact frame from any start)
            legs["fixed_pose_scanning_frame_max_channel_diff_vs_value_frame"] = (
                int(np.abs(((image.cpu().numpy().view(np.uint32)[..., None] >> np.array([0, 8, 16, 24], np.uint32)) & 255).astype(np.int16)
                           - ((final.cpu().numpy().view(np.uint32)[..., None] >> np.array([0, 8, 16, 24], np.uint32)) & 255).astype(np.int16)).max())
                if len(poses) == 1 else None)
            R.set_option(_L.OPT_START_HINTS, hints_before)
            # (1e) a camera in slow motion: a yaw of 0.1 degrees a frame (a 90 degrees/s pan at 900 frames/s; the 10-degree steps
            # of leg (1) are key presses, src/main.rs:57-60)
            if not args.orbit:
                slow_cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0))
                slow_cam.update_camera_pose()
                slow = []
                for k in range(210):
                    slow.append(slow_cam.to_c(pipe.LOWPASS, pipe.SH_DIM))
                    slow_cam.update_yaw_angle(0.1 * np.pi / 180.0)
                    slow_cam.update_camera_pose()
                for k in range(10):
                    R.render_frame_device(slow[k], image.data_ptr())
                torch.cuda.synchronize()
                with counted("slow_pan_0p1_deg_per_frame_fps"):
                    t1 = time.perf_counter()
                    for k in range(10, 210):
                        R.render_frame_device(slow[k], image.data_ptr())
                    torch.cuda.synchronize()
                    legs["slow_pan_0p1_deg_per_frame_fps"] = 200 / (time.perf_counter() - t1)
            # (1c) a two-image swap chain (splat_set_frame_overlap(2)): the compositors of consecutive frames share the chip.
            # Nothing for a frame that fills the chip (C3, C5); a frame bound by its densest tile's lone wave (C1, C2, a
            # multi-GPU slab) runs a third faster.  Both images must hold the frame `value` rendered.
            image2 = torch.zeros_like(image)
            torch.cuda.synchronize()   # (fill on the default stream, frames on the context's)
            R.set_frame_overlap(2)
            pair = (image, image2)
            for k in range(20):
                R.render_frame_device(last_pose, pair[k % 2].data_ptr())
            torch.cuda.synchronize()
            with counted("swap_chain_two_device_images_fps"):
                t1 = time.perf_counter()
                for k in range(200):
                    R.render_frame_device(last_pose, pair[k % 2].data_ptr())
                settle(R)
                torch.cuda.synchronize()
                legs["swap_chain_two_device_images_fps"] = 200 / (time.perf_counter() - t1)
            legs["swap_chain_frames_equal_value_frame"] = bool(torch.equal(image, final) and torch.equal(image2, final))
            R.set_frame_overlap(1)
            del image2
            # (2) host-visible: the literal render_to_buffer -- host image in and out, synchronous (src/main.rs:71-75)
            himg = np.zeros((H, W), np.uint32)
            R.render(cam_c, himg)
            with counted("host_visible_splat_render_fps"):
                t1 = time.perf_counter()
                for _ in range(20):
                    himg[:] = 0
                    R.render(cam_c, himg)
                legs["host_visible_splat_render_fps"] = 20 / (time.perf_counter() - t1)
            # (2b) the same with the caller's image page-locked once (splat_host_register: what a host would do with the
            # reference's long-lived `color` buffer, src/main.rs:62): the copies run at the PCIe rate instead of through
            # the driver's staging buffers
            try:
                splat_amd.Renderer.host_register(himg)
                R.render(cam_c, himg)
                with counted("host_visible_splat_render_registered_fps"):
                    t1 = time.perf_counter()
                    for _ in range(20):
                        himg[:] = 0
                        R.render(cam_c, himg)
                    legs["host_visible_splat_render_registered_fps"] = 20 / (time.perf_counter() - t1)
                splat_amd.Renderer.host_unregister(himg)
            except Exception as e:      # (a driver that will not pin this allocation: the leg is informative only)
                legs["host_visible_splat_render_registered_fps"] = "unavailable: %s" % e
            # (2c) splat_render_frame: the same frame as ONE call that ships no zeros (clear fused, pixels cross PCIe once; a
            # page-locked image is written by the compositor itself) -- fixed pose, then (2d) THE REFERENCE'S LOOP, literally
            # (src/main.rs:43-78): for each of the 36 orbit poses a pose update (C++ host mirror: splat::Camera, as the Rust
            # loop's update_camera_pose), one synchronous host-visible frame, wait.  Three forms: render_frame into the
            # caller's page-locked image, into a pageable one, and the literal pair `clear; render_to_buffer` (splat_render).
            import ctypes as _C
            hostlib = _C.CDLL(os.path.join(ROOT, "splat_amd", "libsplat_host.so"))
            hostlib.splat_host_camera.argtypes = [_C.c_float, _C.c_float, _C.POINTER(_C.c_float), _C.c_float, _C.c_float, _C.c_int,
                                                  _C.c_float, _C.POINTER(_abi.CameraC)]
            cam_pos = (_C.c_float * 3)(0.0, 0.0, 5.0)
            loop_cam = _abi.CameraC()
            step_rad = np.float32(10.0 * np.pi / 180.0)

            def pose_update(k):      # Camera::update_yaw_angle x k + update_camera_pose (src/camera.rs:41-68, 94-126), in C++
                hostlib.splat_host_camera(float(H), float(W), cam_pos, float(np.float32(k % 36) * step_rad), 0.0, 1, pipe.LOWPASS,
                                          _C.byref(loop_cam))
                return loop_cam
            pinned_img = R.host_image(H, W)
            page_img = np.zeros((H, W), np.uint32)
            for name, buf, literal, fixed in (("host_visible_splat_render_frame_fps", pinned_img, False, True),
                                              ("reference_loop_fps", pinned_img, False, False),
                                              ("reference_loop_pageable_image_fps", page_img, False, False),
                                              ("reference_loop_literal_clear_and_render_to_buffer_fps", page_img, True, False)):
                for k in range(8):
                    R.render_frame(cam_c if fixed else pose_update(k), buf)
                with counted(name):
                    t1 = time.perf_counter()
                    for k in range(K):
                        c_k = cam_c if fixed else pose_update(k)
                        if literal:
                            buf[:] = 0                  # color.clear(0), src/main.rs:73
                            R.render(c_k, buf, want_stats=False)
                        else:
                            R.render_frame(c_k, buf)
                    legs[name] = K / (time.perf_counter() - t1)
            # the loop's last frame (pose 71 % 36 = 35) must be the device-resident frame of that pose
            R.render_frame(orbit[35], pinned_img)
            R.render_frame_device(orbit[35], image.data_ptr(), sync=True)
            dev_img = image.cpu().numpy().view(np.uint32)
            if args.mode == "fast":     # (every fast-mode frame is within 1 of the exact frame per colour byte; WHICH one depends on where its walks started)
                sh_ = np.array([0, 8, 16], np.uint32)
                dch = np.abs(((np.asarray(pinned_img)[..., None] >> sh_) & 255).astype(np.int16) - ((dev_img[..., None] >> sh_) & 255).astype(np.int16))
                legs["reference_loop_frame_equals_device_frame"] = bool(dch.max() <= 2 and np.array_equal(np.asarray(pinned_img) >> 24, dev_img >> 24))
            else:
                legs["reference_loop_frame_equals_device_frame"] = bool(np.array_equal(pinned_img, dev_img))
            legs["reference_loop_zero_copy"] = int(R.get_option(_abi.OPT_HOST_ZERO_COPY))
            # (3) host-visible: the viewer loop (clear, render, present) with pinned frames in flight: frame k is presented
            # (waited for) while frames k+1.. render and cross PCIe.  Two buffers are what a double-buffered window has;
            # four keep the device's frame pipeline (two binning chains + a compositor) full
            for nb in (2, 4):
                bufs = [R.host_image(H, W) for _ in range(nb)]
                for k in range(80):                              # the first few dozen frames into fresh pinned buffers run at a
                    R.render_stream(cam_c, bufs[k % nb])         # third of the rate (one-time mapping of the copy path): warm-up
                    if k >= nb - 1:
                        R.stream_wait(bufs[(k - (nb - 1)) % nb])
                for k in range(80 - (nb - 1), 80):
                    R.stream_wait(bufs[k % nb])
                leg = "host_visible_splat_render_stream_fps" if nb == 2 else "host_visible_splat_render_stream_4_in_flight_fps"
                with counted(leg):
                    t1 = time.perf_counter()
                    for k in range(K):
                        R.render_stream(cam_c, bufs[k % nb])
                        if k >= nb - 1:
                            R.stream_wait(bufs[(k - (nb - 1)) % nb])
                    for k in range(max(0, K - (nb - 1)), K):
                        R.stream_wait(bufs[k % nb])
                    legs[leg] = K / (time.perf_counter() - t1)
                last_buf = bufs[(K - 1) % nb]
            legs["host_visible_frames_equal_device_frame"] = bool(np.array_equal(last_buf, himg))
            # (4) the trained-like stand-in (C3s: flat anisotropic Gaussians on surfaces) at C3's size, loaded through
            # write_ply -> load_from_ply, bench pose and a pose inside the sphere: fps, counters, parity
            if args.workload == "C3" and not args.scene:
                legs["c3s_surface_scene"] = surface_leg(R, image, stream, local, args, counted, leg_drops)
            # (5) cold and incoherent frames: first frame, 36 uncorrelated poses, a pose jump
            legs.update(cold_and_incoherent_legs(g, W, H, cam_c, local, stream, main_mode))
            legs["frames_dropped"] = dict(leg_drops)
            legs["what"] = ("host-visible = pixels delivered to host memory (PCIe inclusive); never `value`.  %d frames each "
                            "(splat_render: 20).  frames_dropped = frames the device skipped inside each leg's timed loop "
                            "(splat_frames_dropped() around it, statuses harvested by splat_sync): a skipped frame costs ~0 ms, "
                            "so a leg with drops is not a measurement (exit code 4)" % K)
            R.timing(reset=True)

    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dr = torch.tensor([dropped_timed], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(dr, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    dropped_all = int(dr.item())             # frames any rank's device skipped inside the timed region

    # per-rank stats -> whole-frame totals
    tot = torch.tensor([st.n_visible, st.n_pairs, st.bytes_algorithmic, st.flops_algorithmic, st_steady.n_iter_scan, st_steady.n_iter_blend],
                       dtype=torch.int64, device="cuda")
    comp = torch.tensor([kern_ms["composite"] / max(frames, 1)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(comp, op=dist.ReduceOp.MAX)

    # every rank's per-kernel averages (HIP events on its own streams) travel to rank 0 for the line
    per_rank = None
    if world > 1:
        mine = {k: v / max(frames, 1) for k, v in kern_ms.items()}
        mine.update(rank=rank, tile_rows=slabs[rank][1] - slabs[rank][0], n_pairs=int(st.n_pairs), n_visible=int(st.n_visible),
                    k1_blocks_culled=int(st.n_blocks_culled), frames_dropped=int(dropped_timed))
        box = [None] * world
        dist.all_gather_object(box, mine)
        per_rank = box

    slab_check = None
    exit_code, parity_ok = 0, True
    if world > 1 and rank == 0:
        # the gathered frame must equal the frame this rank renders alone, byte for byte
        R.set_slab(0, -1)
        full = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()       # (the fill runs on torch's default stream, the frame on `stream`: without this the fill
                                       # may land in the middle of the frame -- seen once in ~30 runs with eight ranks on one GPU)
        with torch.cuda.stream(stream):
            R.render_device(last_pose, full.data_ptr(), sync=True)
        torch.cuda.synchronize()
        slab_check = bool(torch.equal(full, final))
        if not slab_check:      # which rows? (a slab's rows that did not arrive look different from pixels rendered differently)
            bad = (full != final)
            rows = torch.nonzero(bad.any(dim=1)).flatten().tolist()
            print("[bench] gathered frame differs from the single-GPU frame: %d pixels in rows %d..%d (%d rows); slabs (tile rows) %s; "
                  "gathered image all zero in %d of those rows" % (int(bad.sum()), rows[0], rows[-1], len(rows), slabs,
                                                                    sum(1 for r_ in rows if not bool(final[r_].any()))), file=sys.stderr, flush=True)
        R.set_slab(*slabs[rank])
    if rank == 0:
        per = {k: v / max(frames, 1) for k, v in kern_ms.items()}
        t_gpu = sum(per[k] for k in ("preprocess", "scan", "emit", "sort", "composite"))
        t_frame = dt / args.steps * 1e3                      # wall time per frame: what the overlapped kernels add up to
        iso = {k: v / max(iso_frames, 1) for k, v in iso_ms.items()} if iso_ms else None
        # dominant kernel: the compositor.  Algorithmic bytes per launch (DESIGN.md "Roofline"):
        # D*(12 sorted key/index + 36 record) read once per tile + 4 B/pixel written.
        slab_px = (sdist.slab_pixel_rows(slabs[0], H)[1] - sdist.slab_pixel_rows(slabs[0], H)[0]) * W
        comp_bytes = st.n_pairs * 48 + slab_px * 4
        achieved = comp_bytes / (per["composite"] * 1e-3) / 1e9 if per["composite"] > 0 else 0.0
        traffic, valu_util, prof_src, valu_insts = None, None, None, None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                wl = tj.get(args.workload, {})
                kname = next((k for k in sorted(wl) if k.startswith("composite_exact_kernel") and not k.endswith(":detail")), None)
                traffic = wl.get(kname) if kname else None
                valu_util = wl.get(kname + ":detail", {}).get("valu_issue_util") if kname else None
                valu_insts = wl.get(kname + ":detail", {}).get("valu_insts") if kname else None
                prof_src = tj.get(args.workload + ":source")
            except Exception:
                traffic = None
        # the frame on the axis that bounds it: VALU wave-instructions of the frame's four kernels (committed
        # counter pass) against this run's frame time on the chip's 1024 SIMDs
        frame_valu = None
        try:
            wlj = json.load(open(tp)).get(args.workload, {}) if os.path.exists(tp) else {}
            parts = {}
            for kn, short in (("composite_exact_kernel<false, false, 2>", "composite"), ("preprocess_kernel<true, false, false>", "preprocess"),
                              ("select_near_kernel", "select"), ("scan_bucket_kernel<256>", "scan")):
                v = wlj.get(kn + ":detail", {}).get("valu_insts")
                if v:
                    parts[short] = float(v)
            if "composite" in parts and "preprocess" in parts and world == 1 and args.mode == "exact" and not args.orbit:
                tot_valu = sum(parts.values())
                frame_valu = {"wave_instructions_per_frame": tot_valu, "by_kernel": parts, "simds": 1024, "clock_ghz": 2.4,
                              "cycles_per_wave_instruction_per_simd": (dt / args.steps) * 2.4e9 * 1024 / tot_valu,
                              "issue_cost_cycles": {"full_rate": "2.4-2.8", "half_rate": "4.1-4.4", "quarter_rate": "8.1", "compositor_mix": "~3.0"},
                              "what": "SQ_INSTS_VALU per launch of the frame's kernels (profiles/%s_pmc_SQ.csv, the fixed-pose frame of this "
                                      "workload) / this run's frame time on 1024 SIMDs at 2.4 GHz, beside the measured issue cost per instruction "
                                      "class (tools/lab/valu_probe.hip, LAB_NOTEBOOK.md section 3): the pipelined frame runs at the chip's VALU "
                                      "issue rate for its instruction mix" % (prof_src or "rNN")}
        except Exception:
            frame_valu = None
        traffic_live, traffic_note, live_all = False, None, None
        if world == 1 and not args.no_live_pmc:
            # bytes measured in THIS run: the context is idle now (the timed region and the legs are over); the two counter
            # passes are processes of their own on the same GPU
            live, detail = live_pmc_traffic(args)
            if live is None:
                traffic_note = "live counter passes unavailable (%s): committed profile used" % detail
            else:
                kname = next((k for k in sorted(live) if k.startswith("composite_exact_kernel")), None)
                if kname:
                    traffic, traffic_live, live_all = live[kname], True, live
                    traffic_note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench.py on this box, after the timed "
                                    "region (3 steps each; KiB -> bytes, FETCH_SIZE doubled: MI355X_MICROARCH.md HBM section)")
        out = {
            "metric": "frames_per_sec", "value": args.steps / dt, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "setup_frames_before_warmup": args.settle,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": data_kind,
            "config": {"workload": "%s: %d Gaussians @%dx%d, %s, Camera(0,0,5), Pipeline01 "
                                   "(lowpass 0.01, sh_dim 15), %s mode" % (args.workload, n, W, H, scene_name, args.mode),
                       "camera": "36-pose yaw orbit, 10 degrees per frame" if args.orbit else "fixed pose",
                       "partition": ("one rank per GPU, load-balanced tile-row slabs %s, one gather of slab rows per frame: %s"
                                     % ([b - a for a, b in slabs],
                                        {"native": "grouped ncclSend/ncclRecv issued by the C ABI (splat_comm_gather)",
                                         "torch.distributed": "torch.distributed batch_isend_irecv"}[gather_kind])
                                     + ("; frames alternate between two device images (splat_set_frame_overlap(2): the compositors of "
                                        "consecutive frames share the chip)" if swap_chain else ""))
                                    if world > 1 else "single GPU",
                       "frame_overlap": 2 if swap_chain else 1,
                       "start_hints": int(R.get_option(_abi.OPT_START_HINTS)),
                       "n_visible": int(tot[0]), "n_pairs": int(tot[1]), "max_tile_len": int(st.max_tile_len),
                       "early_out_fallback_waves": int(st.n_fallback), "sort_fallback_tiles": int(st.n_sort_fallback),
                       "k1_blocks_culled": int(st.n_blocks_culled),
                       "frames_dropped": int(dropped_all),
                       "device_bytes_peak": int(dev_peak), "key_buffer_entries_per_slot": int(key_entries)},
            "roofline": {"bound": "hbm", "kernel": "composite_exact_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "avg_launch_ms": per["composite"], "bytes_per_launch": comp_bytes,
                         "isolated": ({"avg_launch_ms": iso["composite"], "achieved": comp_bytes / (iso["composite"] * 1e-3) / 1e9,
                                       "frac": comp_bytes / (iso["composite"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "what": "the same kernel with a sync after every frame: in the timed region it shares "
                                               "the chip with the next frame's preprocess/scan/sort (cross-frame overlap)"}
                                      if iso and iso["composite"] > 0 else None),
                         "traffic_source": traffic_note,
                         "traffic_all_kernels": live_all,
                         "from_committed_profile": {"traffic": (traffic is not None) and not traffic_live, "valu_issue_util": valu_util,
                                                    "counters": ("profiles/%s_pmc_*.csv" % prof_src) if prof_src else None,
                                                    "what": "VALU issue utilisation (and the HBM bytes per launch when the live counter "
                                                            "passes could not run: `traffic` true here) come from the committed "
                                                            "rocprofv3 --pmc passes of this build; everything else in this object is "
                                                            "measured in this run"},
                         "compositor_work": {"flops_alg": int(tot[3]),
                                             "gflops_alg_per_s": (int(tot[3]) / (per["composite"] * 1e-3) / 1e9) if per["composite"] > 0 else None,
                                             "wave_record_iterations_scan": int(tot[4]), "wave_record_iterations_blend": int(tot[5]),
                                             "valu_per_wave_record_iteration": (valu_insts / (int(tot[4]) + int(tot[5]))) if (valu_insts and int(tot[4]) + int(tot[5]) > 0) else None,
                                             "what": "flops_alg = sum over pixels of its tile's list length x 25 (BASELINE.md section 4); "
                                                     "iterations = (wave, record) steps the compositor actually took on the "
                                                     "statistics frame (early-out + coverage compaction), 64 pixels each; "
                                                     "valu_per_wave_record_iteration = the launch's SQ_INSTS_VALU (committed "
                                                     "profile) / those iterations: every VALU instruction of the kernel -- "
                                                     "staging, alpha pass, sort of the short lists -- charged to the walks "
                                                     "(the blend loop itself is 47 per step, 72 with the bracket's second state)"},
                         "note": "the compositor is bound by VALU issue and LDS latency, not by HBM (DESIGN.md section 4)"},
            "roofline_frame": {"bytes_algorithmic": int(tot[2]), "t_frame_ms": t_frame,
                               "sum_of_kernel_ms": t_gpu,
                               "achieved": int(tot[2]) / (t_frame * 1e-3) / 1e9,
                               "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                               "frac": int(tot[2]) / (t_frame * 1e-3) / 1e9 / (HBM_PEAK_GBS * world),
                               "valu": frame_valu,
                               "note": "B_alg / wall time per frame; kernels of consecutive frames overlap, so the sum of "
                                       "their durations exceeds the frame time"
                                       + ("; N > 1: the ranks' algorithmic bytes summed (every rank reads the whole scene in its "
                                          "K1) against the ranks' HBM peaks together" if world > 1 else "")},
            "kernel_ms": per,
            "kernel_ms_isolated": iso,
        }
        if legs:
            out["extra_legs"] = legs
            # the frame of a MOVING camera -- the only frame the reference renders (src/main.rs:69: `if is_pose_dirty`) -- beside
            # `value`: device resident and asynchronous like it (the 10-degree steps of src/main.rs:57-60), and the reference's
            # own loop (pose update + one synchronous host-visible frame)
            out["value_orbit"] = legs.get("orbit_36_poses_device_resident_fps")
            out["value_reference_loop"] = legs.get("reference_loop_fps")
            out["device_bytes_peak_over_scene_bytes"] = dev_peak / float(n * 276)
        if per_rank is not None:
            out["kernel_ms_per_rank"] = per_rank
            out["kernel_ms_slowest_rank"] = {k: max(r[k] for r in per_rank) for k in per}
        if slab_check is not None:
            out["multi_gpu_frame_equals_single_gpu_frame"] = slab_check
            if not slab_check:
                exit_code = 3
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or (os.cpu_count() or 1)
            ref, ost, cdt = cpu_baseline(g, last_pose, threads)
            gpu_img = final.cpu().numpy().view(np.uint32)
            d = np.abs(np.stack([((gpu_img >> s) & 255).astype(np.int32) - ((ref >> s) & 255).astype(np.int32)
                                 for s in (0, 8, 16, 24)]))
            out["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "frames/s", "cores": threads, "kind": "port",
                                   "sample": "1 frame of the same workload (oracle restatement, row-band threads; "
                                             "preprocess %.0f ms, sort %.0f ms, raster %.0f ms)" %
                                             (ost.ms_preprocess, ost.ms_sort, ost.ms_raster)}
            out["parity"] = {"max_channel_diff_lsb": int(d.max()), "pixels_differing": int((d.max(0) > 0).sum()),
                             "pixels": int(W * H), "fragments": int(ost.n_fragments),
                             "pairs_equal": bool(int(tot[1]) == int(ost.n_tile_pairs) and int(tot[0]) == int(ost.n_visible)),
                             "tolerance_lsb": 1 if args.mode == "exact" else 2,
                             "against": "oracle/ (C++ restatement of src/gaussians.rs + src/pipelines.rs + src/camera.rs); "
                                        "euc conventions ASSUMED, not pinned: y_up=1 from notes/screenshot.png and from the Python "
                                        "prototype the Rust was ported from, which maps NDC to pixels with (1 - y) * height / 2 "
                                        "(notes/util.py:101-113) -- two inferences, no pin (SURVEY appendix B recollects "
                                        "CoordinateMode::VULKAN = y down); pixel-centre "
                                        "samples, z-clip [0,1], inclusive rectangle -- euc@290e14c is not in the image"}
            # (--mode fast: within 1 of the exact frame by construction -- checked below -- and the exact frame within 1 of the
            # oracle through the exponential's last place: 2 in principle, 1 in every frame measured)
            parity_ok = out["parity"]["max_channel_diff_lsb"] <= (1 if args.mode == "exact" else 2) and out["parity"]["pairs_equal"]
            # the verification mode: fragment()'s exp computed as the host libm does -> the oracle's frame bit for bit
            R.close()
            R2 = splat_amd.Renderer(device=local, mode=splat_amd.MODE_LIBM_EXP)
            R2.upload(g)
            R2.set_stream(stream.cuda_stream)
            with torch.cuda.stream(stream):
                image.zero_()
                R2.render_device(last_pose, image.data_ptr(), sync=True)
            torch.cuda.synchronize()
            exact_img = image.cpu().numpy().view(np.uint32)
            for _ in range(5):
                R2.render_frame_device(last_pose, image.data_ptr())
            torch.cuda.synchronize()
            with counted("parity.libm_exp_mode", R2):
                t1 = time.perf_counter()
                for _ in range(60):
                    R2.render_frame_device(last_pose, image.data_ptr())
                torch.cuda.synchronize()
                libm_fps = 60 / (time.perf_counter() - t1)
            out["parity"]["libm_exp_mode"] = {"pixels_differing": int((exact_img != ref).sum()),
                                              "frames_per_sec": libm_fps, "frames_dropped": leg_drops["parity.libm_exp_mode"],
                                              "what": "SPLAT_MODE_LIBM_EXP: exp as glibc computes it (double, table + cubic); every "
                                                      "other operation is already the reference's, so the frame must be the "
                                                      "oracle's bit for bit -- the default mode's differing pixels are the "
                                                      "exponential's last place and nothing else"}
            parity_ok = parity_ok and out["parity"]["libm_exp_mode"]["pixels_differing"] == 0
            R2.close()
            # the fast mode: the same frame to within one count per colour byte, by construction (include/splat_hip.h)
            # (with --mode fast the roles swap: the timed frame is the fast one, this renderer supplies the exact frame)
            R3 = splat_amd.Renderer(device=local, mode=splat_amd.MODE_EXACT if args.mode == "fast" else splat_amd.MODE_FAST)
            R3.upload(g)
            R3.set_stream(stream.cuda_stream)
            with torch.cuda.stream(stream):
                for _ in range(3):
                    image.zero_()
                    R3.render_device(last_pose, image.data_ptr(), sync=True)
            torch.cuda.synchronize()
            fast_img = image.cpu().numpy().view(np.uint32)
            for _ in range(10):
                R3.render_frame_device(last_pose, image.data_ptr())
            torch.cuda.synchronize()
            with counted("parity.other_mode", R3):
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    R3.render_frame_device(last_pose, image.data_ptr())
                torch.cuda.synchronize()
                fast_fps = args.steps / (time.perf_counter() - t1)
            R3.close()

            def ch(a):
                return np.stack([((a >> s) & 255).astype(np.int32) for s in (24, 16, 8, 0)])
            dfe, dfo = np.abs(ch(fast_img) - ch(gpu_img)), np.abs(ch(fast_img) - ch(ref))
            out["parity"]["fast_mode" if args.mode == "exact" else "exact_mode"] = {"frames_per_sec": fast_fps, "frames_dropped": leg_drops["parity.other_mode"], "max_channel_diff_vs_exact_frame": int(dfe.max()),
                                          "alpha_bytes_differing": int((dfe[0] > 0).sum()),
                                          "pixels_differing_vs_exact_frame": int((dfe.max(0) > 0).sum()),
                                          "max_channel_diff_vs_oracle": int(dfo.max()),
                                          "pixels_beyond_1_lsb_vs_oracle": int((dfo.max(0) > 1).sum()),
                                          "what": "SPLAT_MODE_FAST (opt-in, never `value`): the early-out's bracket closes at "
                                                  "hi - lo <= 2 and the walk continues from its middle; blend() never expands a "
                                                  "difference of states, so every colour byte is within 1 of the exact frame's "
                                                  "(must hold: exit code 3 otherwise), alpha bytes equal; %d frames" % args.steps}
            parity_ok = parity_ok and int(dfe.max()) <= 1 and int((dfe[0] > 0).sum()) == 0
        if dropped_all or any(leg_drops.values()):
            sys.stderr.write("bench.py: the device SKIPPED frames inside a timed region (headline %d, legs %s): the rate is not "
                             "a measurement\n" % (dropped_all, json.dumps(leg_drops)))
            exit_code = 4
        for leg in (out.get("extra_legs", {}).get("c3s_surface_scene", {}) or {}).values():
            if isinstance(leg, dict) and "parity" in leg and not (leg["parity"]["max_channel_diff_lsb"] <= 1 and leg["parity"]["pairs_equal"]):
                parity_ok = False
        if out.get("extra_legs", {}).get("swap_chain_frames_equal_value_frame") is False:
            parity_ok = False                  # (frames composited side by side must be the frame composited alone)
        if out.get("extra_legs", {}).get("reference_loop_frame_equals_device_frame") is False:
            parity_ok = False                  # (a frame the compositor stored into host memory must be the device image's frame)
        print(json.dumps(out))
        if not parity_ok:
            sys.stderr.write("bench.py: PARITY MISS against the oracle: %s\n" % json.dumps(out["parity"]))
            exit_code = 3
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    R.close()
    if exit_code:
        sys.exit(exit_code)       # a fast frame that is not the reference's frame is not a result


if __name__ == "__main__":
    main()
