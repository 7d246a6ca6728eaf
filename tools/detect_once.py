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
if mode == "batch":  # 254 frames of the sequence in one ktlvo_detect_gftt_frames call, as the benchmark times it
    import time
    import numpy as np
    nf = 254
    frames = synth.render_corridor_frames(W, H, 0x5EED0002, 128, workers=8)
    ctx.close()
    ctx = K.Context(K.default_config(width=W, height=H, num_slots=nf, max_points=4096))
    ctx.upload_images(list(range(nf)), np.stack([frames[i % 128] for i in range(nf)]))
    gp = K.gftt_params(2000, 0.01, 10.0, 3, 0, 0.0)
    for _ in range(4):
        t = time.perf_counter()
        out = ctx.detect_gftt_frames(list(range(nf)), gp)
        dt = time.perf_counter() - t
    print("batch of %d frames: %.3f ms per call (results to the host)" % (nf, dt * 1e3))
for _ in range(0 if mode == "batch" else 5):
    if mode == "frame":
        out = ctx.detect_gftt(0, [(0, 0, W, H)], K.gftt_params(2000, 0.01, 10.0, 3, 0, 0.0))
    else:
        out = ctx.detect_gftt(0, grid, K.gftt_params(100, 0.01, 10.0, 3, 1, 0.0))
print(mode, sum(len(x) for x, _ in out), "corners")
ctx.close()
