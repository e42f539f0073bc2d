#!/usr/bin/env python3
"""Regression anchor: a 64-Gaussian scene @64x64 rendered by the ORACLE (both pipelines' low-pass
values), plus an exhaustive blend() table.  Inputs and outputs only; regenerate with
`python tests/golden/make_anchor.py` (from the repo root) if the oracle is deliberately changed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import splat_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402
from helpers import scene_dict, oracle_camera, make_camera, with_oracle_cov3d  # noqa: E402


def main():
    g = splat_amd.synthetic_scene(64, 42)
    g.scales *= np.float32(6.0)                 # splats a few pixels wide at 64x64
    with_oracle_cov3d(g)
    cam = make_camera(64, 64, (0.0, 0.0, 4.0), yaw=0.3, pitch=-0.2)
    out = dict(positions=g.positions, cov3d=g.cov3d, opacities=g.opacities, sh=g.sh,
               view=np.array(cam.to_c(0.01).view[:], np.float32), proj=np.array(cam.to_c(0.01).proj[:], np.float32),
               cam_pos=cam.position)
    for name, lp in (("img_lowpass_0p01", 0.01), ("img_lowpass_0p3", 0.3)):
        img, st = O.render(scene_dict(g), oracle_camera(cam, lp))
        out[name] = img
        assert st.n_visible > 40 and img.any()
    # blend(): old byte x alpha x colour
    olds = np.arange(256, dtype=np.uint32)
    alphas = np.array([1 / 255, 0.1, 0.5, 0.99], np.float32)
    cols = np.array([-0.25, 0.0, 0.37, 1.0, 1.6], np.float32)
    tab = np.zeros((256, len(alphas), len(cols)), np.uint32)
    for o in olds:
        px = (0x7F << 24) | (int(o) << 16) | (int(255 - o) << 8) | int(o ^ 0xA5)
        for i, a in enumerate(alphas):
            for j, c in enumerate(cols):
                tab[o, i, j] = O.blend(px, [c, c, c, a])
    out.update(blend_alphas=alphas, blend_cols=cols, blend_table=tab)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "anchor_64.npz"), **out)
    print("wrote anchor_64.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
