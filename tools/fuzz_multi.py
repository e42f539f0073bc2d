#!/usr/bin/env python3
"""Fuzz the native multi-GPU layer on ONE GPU (the device listed k times: copy transport): random scenes, target sizes
whose last tile row is partial, 1..6 ranks, balanced and equal slabs, host in/out frames onto a random image and
viewer-loop frames queued back to back over changing poses -- the gathered frame must be the single-context frame
byte for byte.   usage: python tools/fuzz_multi.py [n_cases] [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import splat_amd
from splat_amd.renderer import SplatError
from helpers import make_camera

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bad = 0
t0 = time.time()
for case in range(ncases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([500, 8000, 60000, 150000]))
    g = splat_amd.synthetic_scene(n, seed0 + case)
    if rng.integers(0, 3) == 0: g.positions[:, :3] *= 0.3
    H, W = int(rng.choice([40, 100, 250, 392])), int(rng.choice([64, 200, 520]))
    k = int(rng.integers(1, 7))
    R = splat_amd.Renderer()
    g.compute_cov3d(R); R.upload(g)
    poses = [make_camera(H, W, [(0, 0, 5.0), (0.1, 0, 4.0), (0.3, 0.2, 0.4), (0, 0, 12.0)][int(rng.integers(0, 4))],
                         yaw=float(rng.uniform(0, 6.28)), pitch=float(rng.uniform(-0.4, 0.4))).to_c(float(rng.choice([0.01, 0.3])), 15) for _ in range(6)]
    init = rng.integers(0, 2**32, (H, W), dtype=np.uint64).astype(np.uint32)
    want_onto = init.copy(); R.render(poses[0], want_onto)
    want_clear = []
    for p in poses:
        im = np.zeros((H, W), np.uint32); R.render(p, im); want_clear.append(im)
    R.close()
    M = splat_amd.MultiRenderer([0] * k)
    try:
        M.upload(g)
        if case % 2: M.set_frame_overlap(2)               # every other case: two slab images per rank in turn, frames side by side
        if rng.integers(0, 2): M.balance(poses[0])
        got = init.copy(); M.render(poses[0], got)
        if not np.array_equal(got, want_onto):
            bad += 1; print("CASE %d seed %d: %d ranks, host in/out frame differs (%d px), slabs %s" % (case, seed0 + case, k, int((got != want_onto).sum()), M.slabs()))
        def synced_frame(p):
            # a queued frame whose slab outgrew storage sized for another pose is skipped on that rank and reported by the
            # sync (storage regrown): render it again, as a caller would
            for attempt in range(6):
                try:
                    M.sync()
                    return True
                except SplatError as e:
                    if e.code != -4: raise
                    M.render_frame(p)
            return False
        for p in poses: M.render_frame(p)                 # queued back to back; the image holds the last one
        if not synced_frame(poses[-1]):
            bad += 1; print("CASE %d seed %d: %d ranks: still out of capacity after six attempts" % (case, seed0 + case, k))
        last = M.download(H, W)
        if not np.array_equal(last, want_clear[-1]):
            bad += 1; print("CASE %d seed %d: %d ranks, last of the queued viewer-loop frames differs (%d px)" % (case, seed0 + case, k, int((last != want_clear[-1]).sum())))
        j = int(rng.integers(0, len(poses)))
        M.balance(poses[j]); M.render_frame(poses[j]); synced_frame(poses[j])
        if not np.array_equal(M.download(H, W), want_clear[j]):
            bad += 1; print("CASE %d seed %d: %d ranks, frame after re-balancing differs" % (case, seed0 + case, k))
    finally:
        M.close()
print("fuzz_multi: %d cases, %d failures, %.0f s" % (ncases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
