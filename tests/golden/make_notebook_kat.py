#!/usr/bin/env python3
"""Generate tests/golden/notebook_kat.json by running the REFERENCE's own Python prototype
(/root/reference/notes/{util,util_gau}.py + the Gaussian class of 00_Gaussian_Projection.ipynb
cell 1) in the authoring container.  The reference source is imported from where it lies and is
never copied; only numeric inputs/outputs are committed.

The prototype needs PyGLM / PyOpenGL / plyfile, which are absent: they are replaced by stubs
(the prototype only uses glm.lookAt / glm.perspective; a 20-line numpy stand-in follows the
published GLM formulas).  Float64, low-pass 0.3.  Run:  python tests/golden/make_notebook_kat.py
"""
import json
import os
import sys
import types

import numpy as np
import scipy as sp
import scipy.spatial.transform  # noqa: F401

REF = "/root/reference/notes"


def _glm_stub():
    glm = types.ModuleType("glm")

    def lookAt(eye, center, up):
        eye, center, up = (np.asarray(v, np.float64) for v in (eye, center, up))
        f = center - eye
        f = f / np.linalg.norm(f)
        s = np.cross(f, up)
        s = s / np.linalg.norm(s)
        u = np.cross(s, f)
        m = np.eye(4)
        m[0, :3], m[1, :3], m[2, :3] = s, u, -f
        m[0, 3], m[1, 3], m[2, 3] = -s @ eye, -u @ eye, f @ eye
        return m

    def perspective(fovy, aspect, n, f):
        t = np.tan(fovy / 2.0)
        m = np.zeros((4, 4))
        m[0, 0] = 1.0 / (aspect * t)
        m[1, 1] = 1.0 / t
        m[2, 2] = -(f + n) / (f - n)
        m[2, 3] = -2.0 * f * n / (f - n)
        m[3, 2] = -1.0
        return m

    glm.lookAt, glm.perspective = lookAt, perspective
    return glm


def main():
    for name in ("OpenGL", "OpenGL.GL", "OpenGL.GL.shaders"):
        sys.modules[name] = types.ModuleType(name)
    pf = types.ModuleType("plyfile")
    pf.PlyData = object
    sys.modules["plyfile"] = pf
    sys.modules["glm"] = _glm_stub()
    sys.path.insert(0, REF)
    import util
    import util_gau

    nb = json.load(open(os.path.join(REF, "00_Gaussian_Projection.ipynb")))
    src = "".join(nb["cells"][1]["source"])
    ns = {"np": np, "sp": sp, "util": util, "Camera": util.Camera, "naive_gaussian": util_gau.naive_gaussian,
          "plt": None}
    # cell 1 ends with plotting-free object construction; keep only what defines the class + objects
    exec(src, ns)
    objs = ns["gaussian_objects"]
    g = util_gau.naive_gaussian()
    out = {"source": "notes/00_Gaussian_Projection.ipynb cell 1 + notes/util.py + notes/util_gau.py",
           "recorded_conics_ipynb_212_215": [[0.07541478, 0.0, 0.07541478], [0.00173521, 0.0, 0.07541478],
                                              [0.07541478, 0.0, 0.00173521], [0.03394433, 0.0, 0.03394433]],
           "naive": {"xyz": g.xyz.tolist(), "rot_wxyz": g.rot.tolist(), "scale": g.scale.tolist(),
                     "opacity": g.opacity.tolist(), "sh": g.sh.tolist()},
           "cases": []}
    for (h, w, pos) in ((720, 1280, (0.0, 0.0, 3.0)), (600, 800, (0.0, 0.0, 5.0)),
                        (720, 1280, (-0.57651054, 2.99040512, -0.03924271))):
        cam = util.Camera(h, w, position=pos)
        case = {"h": h, "w": w, "pos": list(pos), "view": np.array(cam.get_view_matrix()).tolist(),
                "proj": np.array(cam.get_projection_matrix(), np.float64).tolist(),
                "htanfovxy_focal": [float(v) for v in cam.get_htanfovxy_focal()], "gaussians": []}
        for o in objs:
            conic, bbox_cam, bbox_ndc = o.get_conic_and_bb(cam)
            d = o.pos - cam.position
            d = d / np.linalg.norm(d)
            case["gaussians"].append({
                "cov3d": o.cov3D.tolist(), "cov2d": o.get_cov2d(cam).tolist(), "depth": float(o.get_depth(cam)),
                "conic": conic.tolist(), "bboxsize_cam_corner0": bbox_cam[0].tolist(),
                "bbox_ndc_corner0": bbox_ndc[0].tolist(), "color_clipped": o.get_color(d).tolist()})
        out["cases"].append(case)
    # SH known-answer: degree <= 3 on random coefficients (the notebook's get_color keys on len(sh))
    rng = np.random.default_rng(7)
    sh = rng.standard_normal(48) * 0.3
    dirs = [[0.0, 0.0, 1.0], [0.6, -0.48, 0.64], [-0.2672612419124244, 0.5345224838248488, -0.8017837257372732]]
    shk = {"sh48": sh.tolist(), "dirs": dirs, "dims": {}}
    G = ns["Gaussian"]
    for dim in (3, 12, 27, 48):
        o = G(np.zeros(3), np.ones(3), np.array([1.0, 0, 0, 0]), np.array([1.0]), sh[:dim])
        # undo the clip by evaluating on scaled-down coefficients: colour-0.5 is linear in sh
        o.sh = sh[:dim] * 1e-3
        shk["dims"][str(dim)] = [((o.get_color(np.array(d)) - 0.5) * 1e3).tolist() for d in dirs]
    out["sh_kat"] = shk
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "notebook_kat.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
