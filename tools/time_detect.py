#!/usr/bin/env python3
"""Device time of the corner kernels for one whole-frame detection (2000 Shi-Tomasi corners,
1241x376) and for the reference's 160-ROI key-frame grid; run on a GPU box."""
import os
import sys


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slam_amd import lib as K, synth  # noqa: E402

W, H = 1241, 376
img = synth.CorridorScene(W, H, seed=0x5EED0002).render(3)
ctx = K.Context(K.default_config(width=W, height=H, num_slots=1, max_points=4096))
ctx.upload_image(0, img)
grid = [(30 + i * 59, 30 + j * 39, 59, 39) for j in range(8) for i in range(20)]
for name, rois, gp in (("whole frame", [(0, 0, W, H)], K.gftt_params(2000, 0.01, 10.0, 3, 0, 0.0)),
                       ("160 ROIs", grid, K.gftt_params(100, 0.01, 10.0, 3, 1, 0.0))):
    for _ in range(3):
        ctx.detect_gftt(0, rois, gp)
    ctx.profile_enable(True); ctx.profile_reset()
    import time
    t0 = time.perf_counter()
    for _ in range(20):
        out = ctx.detect_gftt(0, rois, gp)
    wall = (time.perf_counter() - t0) / 20
    ms, n = ctx.profile_get(K.K_CORNER)
    ctx.profile_enable(False)
    print("%s: %d corners, corner kernels %.1f us per call (%d launches), wall %.3f ms per call" % (
        name, sum(len(x) for x, _ in out), ms / 20 * 1e3, n // 20, wall * 1e3))
