"""Shared test helpers: scenes as the oracle wants them, image comparison."""
import numpy as np

import splat_amd
from oracle import oracle as O


def scene_dict(g):
    """GaussianList -> the dict the oracle wrapper takes."""
    return dict(pos4=g.positions, cov3d=g.cov3d, opacity=g.opacities, sh=g.sh)


def with_oracle_cov3d(g):
    """CPU-only tests: fill cov3d with the oracle (the product computes it on the GPU)."""
    g.cov3d = O.compute_cov3d(g.scales, g.rotations)
    return g


def oracle_camera(cam, lowpass, sh_dim=15):
    """Build the oracle's camera struct from the PRODUCT camera's constants (same inputs to both)."""
    c = cam.to_c(lowpass, sh_dim)
    oc = O.Camera()
    oc.view[:] = list(c.view)
    oc.proj[:] = list(c.proj)
    oc.w, oc.h = c.w, c.h
    oc.htanx, oc.htany, oc.focal = c.htanx, c.htany, c.focal
    oc.cam_pos[:] = list(c.cam_pos)
    oc.lowpass, oc.sh_dim = c.lowpass, c.sh_dim
    return oc


def channels(img):
    img = np.asarray(img, np.uint32)
    return np.stack([(img >> s) & 0xFF for s in (24, 16, 8, 0)], -1).astype(np.int32)   # A R G B


def image_diff(a, b):
    """(max per-channel |diff|, number of differing pixels)"""
    d = np.abs(channels(a) - channels(b))
    return int(d.max()) if d.size else 0, int((d.max(-1) > 0).sum()) if d.size else 0


def make_camera(h, w, pos=(0.0, 0.0, 5.0), yaw=0.0, pitch=0.0):
    cam = splat_amd.Camera(h, w, pos)
    if yaw:
        cam.update_yaw_angle(yaw)
    if pitch:
        cam.update_pitch_angle(pitch)
    cam.update_camera_pose()
    return cam
