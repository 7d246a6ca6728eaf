"""Exact-sum LK against the oracle's exact-sum mode: bit equality of positions, status and error on texture, corridor
and white-noise frames (run on a GPU box).  usage: lk_exact_check.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402
from slam_amd import lib as K, synth  # noqa: E402

rng = np.random.default_rng(5)
cases = []
a, b = synth.texture_pair(1241, 376, 5, 3.3, -1.7, noise=True)
cases.append(("texture 1241x376", a, b))
sc = synth.CorridorScene(1241, 376, seed=0x5EED0002)
cases.append(("corridor", sc.render(0), sc.render(1)))
cases.append(("white noise 164x140", rng.integers(0, 256, (140, 164), dtype=np.uint8), rng.integers(0, 256, (140, 164), dtype=np.uint8)))
n = rng.integers(0, 256, (300, 400), dtype=np.uint8)
n[:100, :200] = 255
n[150:, 200:] = 0
cases.append(("shifted noise, saturated", n, np.roll(n, (2, -2), (0, 1))))
bad = 0
for name, a, b in cases:
    h, w = a.shape
    ctx = K.Context(K.default_config(width=w, height=h, num_slots=2, max_points=4096))
    ctx.upload_image(0, np.ascontiguousarray(a)); ctx.upload_image(1, np.ascontiguousarray(b))
    ctx.build_pyramids([0, 1])
    pts = np.stack([rng.uniform(-30, w + 30, 3000), rng.uniform(-30, h + 30, 3000)], 1).astype(np.float32)
    for sums, mode in ((K.LK_SUMS_EXACT, 3), (K.LK_SUMS_TREE, 0)):
        go, gs, ge = ctx.pyrlk(0, 1, pts, K.lk_params(sums=sums))
        oo, os_, oe = O.pyrlk(a, b, pts, nthreads=8, sums=mode)
        same = np.array_equal(go.view(np.uint32), oo.view(np.uint32)) and np.array_equal(gs, os_) and np.array_equal(ge.view(np.uint32), oe.view(np.uint32))
        nd = int((go.view(np.uint32) != oo.view(np.uint32)).any(1).sum())
        print("%-28s sums=%d vs oracle mode %d: bit-equal=%s  points differing=%d  status differing=%d  max|d|=%.6f" % (
            name, sums, mode, same, nd, int((gs != os_).sum()), float(np.abs(go - oo).max())))
        if sums == K.LK_SUMS_EXACT and not same:
            bad += 1
            i = np.nonzero((go.view(np.uint32) != oo.view(np.uint32)).any(1) | (gs != os_))[0][:5]
            for k in i:
                print("   pt", k, pts[k], "gpu", go[k], gs[k], ge[k], "oracle", oo[k], os_[k], oe[k])
    ctx.close()
sys.exit(1 if bad else 0)
