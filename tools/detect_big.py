"""Batched whole-frame detection on big frames, for timing and kernel traces:
  python tools/detect_big.py 1080|4k|kitti [frames per call]   (default 31 frames of 1920x1080, 15 of 3840x2160, 16 of 1241x376)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slam_amd import lib as K, synth
W, H, nf, nc, md = {"1080": (1920, 1080, 31, 8000, 5.0), "4k": (3840, 2160, 15, 16384, 8.0), "kitti": (1241, 376, 16, 2000, 10.0)}[sys.argv[1]]
if len(sys.argv) > 2:
    nf = int(sys.argv[2])
frames = synth.render_corridor_frames(W, H, 0x5EED0002, 4, workers=4)
ctx = K.Context(K.default_config(width=W, height=H, num_slots=nf, max_points=max(4096, nc)))
ctx.upload_images(list(range(nf)), np.stack([frames[i % 4] for i in range(nf)]))
gp = K.gftt_params(nc, 0.01, md, 3, 0, 0.0)
for _ in range(5):
    t = time.perf_counter(); out = ctx.detect_gftt_frames(list(range(nf)), gp); dt = time.perf_counter() - t
print("batch of %d frames %dx%d: %.3f ms per call (results to the host), %d corners" % (nf, W, H, dt * 1e3, sum(len(x) for x, _ in out)))
ctx.close()
