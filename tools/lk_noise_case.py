#!/usr/bin/env python3
"""Replays tests/test_gpu_fuzz.py::test_pyramid_scharr_lk_random_sizes for a seed and prints, for every noise frame whose
tracks disagree beyond 0.05 px at the 95th percentile, how far the kernel is from the serial oracle and how far the
oracle is from itself with pairwise sums.  usage: lk_noise_case.py SEED CASES"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from slam_amd import lib as K  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
import oracle  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]))
for i, (w, h) in enumerate(F._sizes(rng, int(sys.argv[2]))):
    a, b = F._images(rng, w, h, i % 3)
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    n = 600
    pts = np.stack([rng.uniform(-30, w + 30, n), rng.uniform(-30, h + 30, n)], 1).astype(np.float32)
    pts[:8] = [[0, 0], [w - 1, h - 1], [w, h], [-21, -21], [w - 0.5, 0.25], [0.25, h - 0.5], [w / 2, h / 2], [10.5, 10.5]]
    if i % 3 == 0:
        continue
    ctx = K.Context(K.default_config(width=w, height=h, num_slots=2, max_points=1024))
    ctx.upload_image(0, a); ctx.upload_image(1, b); ctx.build_pyramids([0, 1])
    go, gs, ge = ctx.pyrlk(0, 1, pts)
    oo, os_, oe = oracle.pyrlk(a, b, pts, nthreads=8)
    both = (gs == os_) & (gs == 1)
    d = np.abs(go - oo).max(1)[both]
    if both.any() and np.quantile(d, 0.95) > 0.05:
        oracle.set_lk_sum_mode(2)
        try:
            po_, ps_, _ = oracle.pyrlk(a, b, pts, nthreads=8)
        finally:
            oracle.set_lk_sum_mode(0)
        ok = both & (ps_ == 1)
        dg, do = np.abs(go - oo).max(1)[ok], np.abs(po_ - oo).max(1)[ok]
        dp = np.abs(go - po_).max(1)[ok]
        q = lambda v: " ".join("%.3f" % np.quantile(v, p) for p in (0.5, 0.9, 0.95, 0.99))
        print("case %d %dx%d kind %d points %d: kernel-serial q50/90/95/99 %s (>0.05: %d) | pairwise-serial %s (>0.05: %d) | kernel-pairwise %s (>0.05: %d)"
              % (i, w, h, i % 3, ok.sum(), q(dg), (dg > 0.05).sum(), q(do), (do > 0.05).sum(), q(dp), (dp > 0.05).sum()), flush=True)
    ctx.close()
