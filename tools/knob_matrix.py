#!/usr/bin/env python3
"""The schedule options that carry heuristics, held against fixed settings: for every cell of
    {C2, C3, C3s, C5} x {rest, 1 deg/frame, 10 deg/frame, inside (3 deg/frame from inside the cloud), random poses}
the frame rate (device-resident asynchronous frames, as bench.py's `value`) with every option on its default / automatic
setting, and with ONE option at a time forced to each of its other settings:
    pair walk (auto | 0 | 1), sort in compositor (auto | 0 | 1), near selection (2048 | 0), early-out min list (768 | 384 | 1536),
    overflow redo (adaptive | 0 | 2), start hints (2 | 0 | 1), count first (default | the other two of 0, 1, 2),
    large-splat list threshold (128 tiles | no list | window only | 512), large list kept from (256 large splats | always).
A setting under which the device skipped frames inside the timed loop is no alternative (a skipped frame costs nothing).
Writes the table as JSON (profiles/r07_knob_matrix.json is this tool's output); tests/test_gpu_knobs.py holds a subset live.
usage: knob_matrix.py [--scenes C2,C3] [--motions rest,10deg] [--frames 100] [--out file.json]"""
import json, math, sys, time
sys.path.insert(0, ".")
import numpy as np, torch, splat_amd
from splat_amd import _lib as L
from splat_amd.renderer import SplatError
from bench import WORKLOADS, make_scene

KNOBS = [("pair_walk", L.OPT_PAIR_WALK, -1, (0, 1)), ("sort_in_compositor", L.OPT_SORT_IN_COMPOSITOR, -1, (0, 1)),
         ("near_select_keys", L.OPT_NEAR_SELECT_KEYS, 2048, (0,)), ("early_out_min_list", L.OPT_EARLY_OUT_MIN_LIST, 768, (384, 1536)),
         ("overflow_redo", L.OPT_OVERFLOW_REDO, 1, (0, 2)), ("start_hints", L.OPT_START_HINTS, 2, (0, 1)),
         ("count_first", L.OPT_COUNT_FIRST, None, (0, 1, 2)),
         ("large_splat_tiles", L.OPT_LARGE_SPLAT_TILES, 128, (-1, 0, 512)), ("large_list_min", L.OPT_LARGE_LIST_MIN, 256, (0,))]
MOTIONS = ("rest", "1deg", "10deg", "inside", "random")


def poses_for(motion, H, W, n):
    if motion == "random":
        rng = np.random.default_rng(11)
        out = []
        for k in range(n):
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            r = rng.uniform(0.2, 1.2) if k % 3 == 0 else rng.uniform(2.5, 7.0)
            cam = splat_amd.Camera(H, W, tuple(float(x) for x in d * r))
            cam.update_yaw_angle(float(rng.uniform(0, 2 * math.pi))); cam.update_pitch_angle(float(rng.uniform(-0.5, 0.5)))
            cam.update_camera_pose()
            out.append(cam.to_c(0.01, 15))
        return out
    pos, step, yaw0 = {"rest": ((0.0, 0.0, 5.0), 0.0, 0.0), "1deg": ((0.0, 0.0, 5.0), 1.0, 0.0), "10deg": ((0.0, 0.0, 5.0), 10.0, 0.0),
                       "inside": ((0.3, 0.2, 0.4), 3.0, 1.0)}[motion]
    cam = splat_amd.Camera(H, W, pos)
    if yaw0: cam.update_yaw_angle(yaw0)
    out = []
    for k in range(n):
        cam.update_camera_pose()
        out.append(cam.to_c(0.01, 15))
        if step: cam.update_yaw_angle(math.radians(step))
    return out


def measure(R, poses, img, warm, frames):
    def settle():
        try: R.sync()
        except SplatError as e:
            if e.code != L.ERR_CAPACITY: raise
    best, dropped = 0.0, 0
    for rep in range(3):                 # (the best of three: a few per cent of noise would otherwise decide cells)
        for attempt in range(4):         # warm-up, again while it still outgrows storage (buffers grow at the sync behind it)
            d0 = R.frames_dropped()
            for k in range(warm): R.render_frame_device(poses[k % len(poses)], img.data_ptr())
            settle(); torch.cuda.synchronize()
            if R.frames_dropped() == d0: break
        d0 = R.frames_dropped()
        t0 = time.perf_counter()
        for k in range(warm, warm + frames): R.render_frame_device(poses[k % len(poses)], img.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        settle()
        if R.frames_dropped() - d0: dropped = max(dropped, R.frames_dropped() - d0)
        else: best = max(best, frames / dt)
    return best, (dropped if best == 0.0 else 0)


def run(scenes=("C2", "C3", "C3s", "C5"), motions=MOTIONS, frames=100, out=None):
    table = {"what": __doc__.split("usage:")[0].strip(), "frames_per_measurement": frames, "cells": {}}
    worst = (1.0, None)
    for wl in scenes:
        n, W, H, seed = WORKLOADS[wl]
        g = make_scene(wl)
        R = splat_amd.Renderer()
        g.compute_cov3d(R); R.upload(g)
        img = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        warm = 30
        for motion in motions:
            poses = poses_for(motion, H, W, warm + frames)
            auto, d_auto = measure(R, poses, img, warm, frames)
            cell = {"auto_fps": round(auto, 1), "auto_frames_dropped": d_auto, "forced": {}}
            best_fixed, best_name = 0.0, None
            for name, opt, default, others in KNOBS:
                if default is None: default = int(R.get_option(opt))
                for v in others:
                    if v == default: continue
                    R.set_option(opt, v)
                    fps, dropped = measure(R, poses, img, warm, frames)
                    R.set_option(opt, default)
                    cell["forced"]["%s=%d" % (name, v)] = {"fps": round(fps, 1), "frames_dropped": dropped}
                    if dropped == 0 and fps > best_fixed: best_fixed, best_name = fps, "%s=%d" % (name, v)
            again, d_again = measure(R, poses, img, warm, frames)       # (the defaults once more, last: whatever the order of measurement gives or takes)
            if d_again == 0 and again > auto: auto = again; cell["auto_fps"] = round(auto, 1)
            cell["best_forced"] = best_name
            cell["auto_over_best_forced"] = round(auto / best_fixed, 4) if best_fixed else None
            table["cells"]["%s/%s" % (wl, motion)] = cell
            if best_fixed and auto / best_fixed < worst[0]: worst = (auto / best_fixed, "%s/%s vs %s" % (wl, motion, best_name))
            print("%-12s auto %7.0f frames/s (%d dropped); best forced %-24s %7.0f  -> auto/best %.3f" %
                  (wl + "/" + motion, auto, d_auto, best_name, best_fixed, auto / best_fixed if best_fixed else float("nan")), flush=True)
        R.close(); del img
    table["worst_cell"] = {"auto_over_best_forced": round(worst[0], 4), "where": worst[1]}
    print("worst cell: auto / best forced = %.3f (%s)" % worst)
    if out:
        json.dump(table, open(out, "w"), indent=1)
    return table


if __name__ == "__main__":
    argv = sys.argv[1:]
    kw = {}
    while argv:
        if argv[0] == "--scenes": kw["scenes"] = argv[1].split(",")
        elif argv[0] == "--motions": kw["motions"] = argv[1].split(",")
        elif argv[0] == "--frames": kw["frames"] = int(argv[1])
        elif argv[0] == "--out": kw["out"] = argv[1]
        argv = argv[2:]
    run(**kw)
