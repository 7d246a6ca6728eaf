#!/usr/bin/env python3
"""Whole-frame Shi-Tomasi detection (2000 corners, 1241x376, the benchmark's detector setting) a few times, and
the reference's 160-sub-region key-frame detection: the workload of the corner kernels' counter passes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slam_amd import lib as K, synth  # noqa: E402

W, H = 1241, 376
img = synth.CorridorScene(W, H, seed=0x5EED0002).render(3)
ctx = K.Context(K.default_config(width=W, height=H, num_slots=1, max_points=4096))
ctx.upload_image(0, img)
sw, sh = (W - 60) / 20.0, (H - 60) / 8.0
grid = [(30 + int(sw * j), 30 + int(sh * i), int(sw), int(sh)) for i in range(8) for j in range(20)]
mode = sys.argv[1] if len(sys.argv) > 1 else "frame"
for _ in range(5):
    if mode == "frame":
        out = ctx.detect_gftt(0, [(0, 0, W, H)], K.gftt_params(2000, 0.01, 10.0, 3, 0, 0.0))
    else:
        out = ctx.detect_gftt(0, grid, K.gftt_params(100, 0.01, 10.0, 3, 1, 0.0))
print(mode, sum(len(x) for x, _ in out), "corners")
ctx.close()
