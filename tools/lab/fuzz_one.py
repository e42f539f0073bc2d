#!/usr/bin/env python3
"""Re-run chosen seeds of tools/fuzz_parity.py several times under an environment override and say which mode's frame is off
and where.  usage: fuzz_one.py REPEATS seed [seed ...]   (environment: e.g. SPLAT_NEAR_KEYS=0)"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np
import splat_amd
from oracle import oracle as O
from helpers import scene_dict, oracle_camera, image_diff
import fuzz_parity as F
reps = int(sys.argv[1])
forced = {k: v for k, v in os.environ.items() if k.startswith("SPLAT_")}
for seed in [int(a) for a in sys.argv[2:]]:
    g, cam, lp, variant, init, desc = F.make_case(seed)
    shd = F.SH_DIMS[seed % len(F.SH_DIMS)]
    conv0 = dict(F.CONVS[(seed // 7) % len(F.CONVS)])
    for k in F.KEYS: os.environ.pop(k, None)
    os.environ.update(variant); os.environ.update(forced)
    sd = scene_dict(g)
    keep = np.isfinite(g.positions).all(axis=1)
    if not g.cov3d.any():
        r0 = splat_amd.Renderer(); g.compute_cov3d(r0); r0.close(); sd = scene_dict(g)
    if not keep.all():
        sd = {k: np.ascontiguousarray(v[keep]) for k, v in sd.items()}
    ref, ost = O.render(sd, oracle_camera(cam, lp, shd), O.default_conventions(**{k: v for k, v in conv0.items() if k != "corrected_projection"}), init.copy(), nthreads=32)
    H, W = int(cam.h), int(cam.w)
    bad = 0
    for rep in range(reps):
        for mode in (0, splat_amd.MODE_LIBM_EXP, splat_amd.MODE_FAST):
            conv = dict(conv0); corrected = splat_amd.MODE_CORRECTED_PROJECTION if conv.pop("corrected_projection", 0) else 0
            R = splat_amd.Renderer(mode=mode | corrected, **conv)
            try:
                R.upload(g)
                img = init.copy()
                st = R.render(cam.to_c(lp, shd), img)
            finally:
                R.close()
            mx, cnt = image_diff(img, ref)
            lim = 0 if mode == splat_amd.MODE_LIBM_EXP else 1
            if mx > lim:
                bad += 1
                d = np.abs(np.stack([((img >> s) & 255).astype(np.int32) - ((ref >> s) & 255).astype(np.int32) for s in (24, 16, 8, 0)])).max(0) > lim
                ty, tx = np.nonzero(d)
                tiles = sorted({(int(y) // 16, int(x) // 16) for y, x in zip(ty, tx)})
                blocks = sorted({(int(y) // 8, int(x) // 8) for y, x in zip(ty, tx)})
                print("seed %d rep %d mode %d: max diff %d, %d px, tiles (row, col) %s, %d 8x8 blocks; near tiles %d repaired %d retries %d longest %d"
                      % (seed, rep, mode, mx, cnt, tiles[:6], len(blocks), st.n_near_tiles, st.n_near_fallback, st.n_fallback, st.max_tile_len))
    print("seed %d (%s): %d bad frames of %d" % (seed, desc, bad, reps * 3))
