"""Lists the points of the randomised LK sweep (tests/test_gpu_fuzz.py, smooth-texture cases) on which GPU and
oracle differ by more than 0.02 px, with their iteration counts per level; run on a GPU box.
usage: fuzz_case.py seed cases"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle as O
from slam_amd import lib as K
import test_gpu_fuzz as T
seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for i, (w, h) in enumerate(T._sizes(rng, cases)):
    a, b = T._images(rng, w, h, i % 3)
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    n = 600
    pts = np.stack([rng.uniform(-30, w + 30, n), rng.uniform(-30, h + 30, n)], 1).astype(np.float32)
    pts[:8] = [[0, 0], [w - 1, h - 1], [w, h], [-21, -21], [w - 0.5, 0.25], [0.25, h - 0.5], [w / 2, h / 2], [10.5, 10.5]]
    if i % 3:
        continue
    ctx = K.Context(K.default_config(width=w, height=h, num_slots=2, max_points=1024))
    ctx.upload_image(0, a); ctx.upload_image(1, b); ctx.build_pyramids([0, 1])
    go, gs, ge = ctx.pyrlk(0, 1, pts)
    oo, os_, oe = O.pyrlk(a, b, pts, nthreads=8)
    both = (gs == os_) & (gs == 1)
    d = np.abs(go - oo).max(1)
    bad = np.flatnonzero(both & (d > 0.02))
    if len(bad):
        pa, pb = O.Pyramid(a, 4, 21, True), O.Pyramid(b, 4, 21, False)
        it, _ = O.lk_iterations(pa, pb, pts[bad], max_level=pa.max_level)
        for k, j in enumerate(bad):
            print("case %d %dx%d pt %s gpu %s oracle %s d %.4f err %.3f/%.3f iterations per level %s" % (
                i, w, h, pts[j], go[j], oo[j], d[j], ge[j], oe[j], it[:, k]))
    ctx.close()
