#!/usr/bin/env python3
"""Fuzz the frame pipeline: random sequences of poses (near, far, inside, orbit jumps) rendered ASYNCHRONOUSLY, several
frames in flight with storage and launch sizes derived from EARLIER frames of the sequence, each into its own
device image; then every image is compared with the same pose rendered synchronously.  A frame may only differ if it
was skipped on the device (then its image is untouched and splat_frames_dropped() accounts for it); also slabs
(2..4 tile-row slabs == the full frame) and the streamed path.   usage: python tools/fuzz_async.py [n_cases] [seed]
  --determinism [n_cases] [seed]: every case's sequence (asynchronous frames + their synchronous references) is rendered by
  THREE processes -- the default pipeline, SPLAT_PIPELINE=1 (one stream, nothing overlaps) and AMD_SERIALIZE_KERNEL=3 (the
  runtime waits before and after every launch) -- and the frames' digests are compared: a race between streams, or a fill the
  launches do not wait for (the allocation-time race of round 5 lived through two rounds), shows up as bytes that depend on
  the schedule."""
import hashlib, json, os, subprocess
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import splat_amd
from splat_amd import _lib
from splat_amd.renderer import SplatError
from helpers import make_camera

def sequence(rng, H, W):
    """jumps, and between them stretches of a camera at rest and of one that creeps (0.03-0.35 degrees a frame)"""
    poses = []
    pos, yaw, pitch, lp = (0, 0, 5.0), 0.0, 0.0, 0.01
    for k in range(18):
        r = rng.uniform()
        if k == 0 or r >= 0.55:
            pos = [(0, 0, 5.0), (0, 0, 1.5), (0.3, 0.2, 0.4), (0, 0, 25.0), (2.0, -1.0, 3.0), (0, 0, 9.0)][int(rng.integers(0, 6))]
            yaw, pitch, lp = float(rng.uniform(0, 6.28)), float(rng.uniform(-0.5, 0.5)), float(rng.choice([0.01, 0.3]))
        elif r >= 0.3:
            yaw += float(rng.uniform(0.0005, 0.006))
        poses.append(make_camera(H, W, pos, yaw=yaw, pitch=pitch).to_c(lp, 15))
    return poses


def emit(seed):
    """child of --determinism: one case under this process's environment; prints the frames' digests"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([3000, 20000, 80000, 200000]))
    g = splat_amd.synthetic_scene(n, seed)
    if rng.integers(0, 3) == 0: g.positions[:, :3] *= rng.choice([0.1, 0.3])
    H, W = int(rng.choice([96, 160, 240])), int(rng.choice([128, 200, 320]))
    R = splat_amd.Renderer()
    g.compute_cov3d(R); R.upload(g)
    poses = sequence(rng, H, W)
    garbage = rng.integers(1, 2**32, (H, W), dtype=np.uint64).astype(np.uint32)
    imgs = [R.device_image(garbage) for _ in poses]
    R.render_device(poses[0], imgs[0], sync=True)
    R.device_free(imgs[0]); imgs[0] = R.device_image(garbage)
    for p_, im in zip(poses, imgs):
        R.render_frame_device(p_, im)
    try:
        R.sync()
    except SplatError:
        pass
    out = {"async": [], "sync": []}
    for im in imgs:
        a = R.device_download(im, H, W)
        out["async"].append("skipped" if np.array_equal(a, garbage) else hashlib.sha256(a.tobytes()).hexdigest())
    for p_ in poses:
        ref = np.zeros((H, W), np.uint32)
        R.render(p_, ref)
        out["sync"].append(hashlib.sha256(ref.tobytes()).hexdigest())
    R.close()
    print("DIGEST " + json.dumps(out))


def determinism(ncases, seed0):
    envs = [("default", {}), ("SPLAT_PIPELINE=1", {"SPLAT_PIPELINE": "1"}), ("AMD_SERIALIZE_KERNEL=3", {"AMD_SERIALIZE_KERNEL": "3"})]
    bad = 0
    t0 = time.time()
    for case in range(ncases):
        got = []
        for name, extra in envs:
            env = dict(os.environ); env.update(extra)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--emit", str(seed0 + case)], capture_output=True, text=True, env=env, timeout=600)
            line = next((l for l in r.stdout.splitlines() if l.startswith("DIGEST ")), None)
            if r.returncode != 0 or line is None:
                bad += 1; print("CASE seed %d under %s: child failed (rc %d)\n%s" % (seed0 + case, name, r.returncode, r.stderr[-400:])); got.append(None); continue
            got.append(json.loads(line[7:]))
        base = got[0]
        for (name, _), d in zip(envs[1:], got[1:]):
            if base is None or d is None: continue
            if d["sync"] != base["sync"]:
                bad += 1; print("CASE seed %d: synchronous frames differ between the default schedule and %s: frames %s" % (seed0 + case, name, [k for k, (a, b) in enumerate(zip(base["sync"], d["sync"])) if a != b]))
            diff = [k for k, (a, b) in enumerate(zip(base["async"], d["async"])) if a != b and "skipped" not in (a, b)]
            if diff:
                bad += 1; print("CASE seed %d: asynchronous frames %s differ between the default schedule and %s" % (seed0 + case, diff, name))
        if base is not None:
            wrong = [k for k, (a, b) in enumerate(zip(base["async"], base["sync"])) if a != b and a != "skipped"]
            if wrong:
                bad += 1; print("CASE seed %d: asynchronous frames %s are not their synchronous renders" % (seed0 + case, wrong))
    print("fuzz_async --determinism: %d cases x %d schedules, %d failures, %.0f s" % (ncases, len(envs), bad, time.time() - t0))
    return bad


if len(sys.argv) > 1 and sys.argv[1] == "--emit":
    emit(int(sys.argv[2])); sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--determinism":
    sys.exit(1 if determinism(int(sys.argv[2]) if len(sys.argv) > 2 else 12, int(sys.argv[3]) if len(sys.argv) > 3 else 500) else 0)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
tot_dropped = tot_frames = 0
t0 = time.time()
for case in range(ncases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([3000, 20000, 80000, 200000]))
    g = splat_amd.synthetic_scene(n, seed0 + case)
    if rng.integers(0, 3) == 0: g.positions[:, :3] *= rng.choice([0.1, 0.3])
    H, W = int(rng.choice([96, 160, 240])), int(rng.choice([128, 200, 320]))
    R = splat_amd.Renderer()
    g.compute_cov3d(R); R.upload(g)
    # the frames whose exact walks start from the previous frames' hints (SPLAT_OPT_START_HINTS) and whose selections are sized tightly
    poses = sequence(rng, H, W)
    garbage = rng.integers(1, 2**32, (H, W), dtype=np.uint64).astype(np.uint32)
    imgs = [R.device_image(garbage) for _ in poses]
    d0 = R.frames_dropped()
    R.render_device(poses[0], imgs[0], sync=True)          # (sizes storage for pose 0 only)
    R.device_free(imgs[0]); imgs[0] = R.device_image(garbage)
    for p, im in zip(poses, imgs):
        R.render_frame_device(p, im)                       # clear + render, asynchronous
    try:
        R.sync()
    except SplatError:
        pass                                               # some frame outgrew its storage: reported here, counted below
    dropped = R.frames_dropped() - d0
    tot_dropped += dropped; tot_frames += len(poses)
    got = [R.device_download(im, H, W) for im in imgs]
    wrong = 0
    refs = []
    hints = R.get_option(_lib.OPT_START_HINTS)
    R.set_option(_lib.OPT_START_HINTS, 0)                   # (the references: every walk scans for its start)
    for k, p in enumerate(poses):
        ref = np.zeros((H, W), np.uint32)
        R.render(p, ref)
        refs.append(ref)
        if not np.array_equal(got[k], ref):
            if np.array_equal(got[k], garbage): wrong += 1           # skipped: untouched
            else:
                bad += 1
                print("CASE %d seed %d: frame %d differs from its synchronous render (%d px) and is not an untouched skip" % (case, seed0 + case, k, int((got[k] != ref).sum())))
    R.set_option(_lib.OPT_START_HINTS, hints)
    if wrong > dropped:
        bad += 1
        print("CASE %d seed %d: %d frames untouched but only %d reported dropped" % (case, seed0 + case, wrong, dropped))
    # the same frames into THREE images in rotation (a swap chain: with SPLAT_FRAME_OVERLAP=2 consecutive frames composite
    # side by side, and a frame must still follow the earlier frames to ITS image): every image ends up holding the last
    # frame rendered to it -- or an earlier one of its frames / nothing, if later ones were skipped and reported
    rot = [R.device_image(garbage) for _ in range(3)]
    d1 = R.frames_dropped()
    for k, p in enumerate(poses):
        R.render_frame_device(p, rot[k % 3])
    try:
        R.sync()
    except SplatError:
        pass
    dropped_rot = R.frames_dropped() - d1
    tot_dropped += dropped_rot; tot_frames += len(poses)
    stale = 0
    for j in range(3):
        mine = [k for k in range(len(poses)) if k % 3 == j]
        have = R.device_download(rot[j], H, W)
        if np.array_equal(have, refs[mine[-1]]): continue
        if np.array_equal(have, garbage) or any(np.array_equal(have, refs[k]) for k in mine[:-1]): stale += 1
        else:
            bad += 1; print("CASE %d seed %d: image %d of the rotation holds none of its frames (%d px off its last one)" % (case, seed0 + case, j, int((have != refs[mine[-1]]).sum())))
    if stale > dropped_rot:
        bad += 1; print("CASE %d seed %d: %d images of the rotation hold an older frame but only %d frames were reported dropped" % (case, seed0 + case, stale, dropped_rot))
    for im in rot: R.device_free(im)
    # slabs == full frame
    p = poses[int(rng.integers(0, len(poses)))]
    full = np.zeros((H, W), np.uint32); R.render(p, full)
    rows = (H + 15) // 16
    cuts = sorted(set([0, rows] + [int(x) for x in rng.integers(0, rows + 1, int(rng.integers(1, 4)))]))
    acc = np.zeros((H, W), np.uint32)
    for a, b in zip(cuts[:-1], cuts[1:]):
        R.set_slab(a, b)
        part = np.zeros((H, W), np.uint32); R.render(p, part)
        acc[a * 16:min(b * 16, H)] = part[a * 16:min(b * 16, H)]
        if part[:a * 16].any() or part[min(b * 16, H):].any():
            bad += 1; print("CASE %d seed %d: slab (%d,%d) wrote outside its rows" % (case, seed0 + case, a, b))
    R.set_slab(0, -1)
    if not np.array_equal(acc, full):
        bad += 1; print("CASE %d seed %d: slabs %s != full frame (%d px)" % (case, seed0 + case, cuts, int((acc != full).sum())))
    # streamed frames == synchronous frames, four in flight
    bufs = [R.host_image(H, W) for _ in range(4)]
    outs = []
    for k, p in enumerate(poses[:8]):
        if k >= 4:
            R.stream_wait(bufs[k % 4]); outs.append(np.array(bufs[k % 4]).copy())
        R.render_stream(p, bufs[k % 4])
    for k in range(4, 8):
        R.stream_wait(bufs[k % 4]); outs.append(np.array(bufs[k % 4]).copy())
    for k, p in enumerate(poses[:8]):
        ref = np.zeros((H, W), np.uint32); R.render(p, ref)
        if not np.array_equal(outs[k], ref):
            bad += 1; print("CASE %d seed %d: streamed frame %d differs (%d px)" % (case, seed0 + case, k, int((outs[k] != ref).sum())))
    for im in imgs: R.device_free(im)
    R.close()
print("fuzz_async: %d cases, %d failures, %d of %d asynchronous frames skipped on the device and reported, %.0f s" % (ncases, bad, tot_dropped, tot_frames, time.time() - t0))
sys.exit(1 if bad else 0)
