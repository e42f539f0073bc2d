import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import splat_amd
from oracle import oracle as O
from helpers import scene_dict, oracle_camera, make_camera, channels
R = splat_amd.Renderer()
n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 256, 256
g = splat_amd.synthetic_scene(n, 1); g.compute_cov3d(R)
cam = make_camera(h, w)
R.upload(g)
img = np.zeros((h, w), np.uint32); st = R.render(cam.to_c(0.01), img)
ref, ost = O.render(scene_dict(g), oracle_camera(cam, 0.01), nthreads=8)
d = np.abs(channels(img) - channels(ref)).max(-1)
print('max', d.max(), 'count', (d > 0).sum(), 'pairs', st.n_pairs, ost.n_tile_pairs, 'maxlen', st.max_tile_len)
ys, xs = np.nonzero(d)
print('rows mod 16 hist', np.bincount(ys % 16, minlength=16))
print('cols mod 16 hist', np.bincount(xs % 16, minlength=16))
tiles = {}
for y, x in zip(ys, xs): tiles[(y // 16, x // 16)] = tiles.get((y // 16, x // 16), 0) + 1
print('tiles affected', len(tiles), list(tiles.items())[:10])
off, order = R.tile_lists(256, st.n_pairs)
lens = np.diff(off)
for (ty, tx), c in list(tiles.items())[:10]: print((ty, tx), 'len', lens[ty * 16 + tx], 'bad px', c)
