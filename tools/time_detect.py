#!/usr/bin/env python3
"""Wall and device time of corner detection through ktlvo_detect_gftt: whole-frame Shi-Tomasi at the three
benchmark shapes and the reference's 160-sub-region key-frame grid; run on a GPU box.
usage: time_detect.py [label]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slam_amd import lib as K, synth  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else ""
cases = [("1241x376 whole frame, 2000 @ minDist 10", 1241, 376, 2000, 10.0, None),
         ("1241x376 160 sub-regions, 100 each", 1241, 376, 100, 10.0, "grid"),
         ("1920x1080 whole frame, 8000 @ minDist 5", 1920, 1080, 8000, 5.0, None),
         ("3840x2160 whole frame, 16384 @ minDist 8", 3840, 2160, 16384, 8.0, None)]
for name, W, H, nc, md, mode in cases:
    img = synth.CorridorScene(W, H, seed=0x5EED0002).render(3)
    ctx = K.Context(K.default_config(width=W, height=H, num_slots=1, max_points=max(4096, nc)))
    ctx.upload_image(0, img)
    if mode == "grid":
        sw, sh = (W - 60) / 20.0, (H - 60) / 8.0
        rois = [(30 + int(sw * j), 30 + int(sh * i), int(sw), int(sh)) for i in range(8) for j in range(20)]
        gp = K.gftt_params(nc, 0.01, md, 3, 1, 0.0)
    else:
        rois, gp = [(0, 0, W, H)], K.gftt_params(nc, 0.01, md, 3, 0, 0.0)
    for _ in range(3):
        out = ctx.detect_gftt(0, rois, gp)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ctx.detect_gftt(0, rois, gp)
    wall = (time.perf_counter() - t0) / reps
    # the C entry point alone, on arrays made once (the wrapper above also builds the ROI table and cuts the
    # output into one pair of arrays per ROI on every call)
    n = len(rois)
    arr = (K.Roi * n)(*[K.Roi(*r) for r in rois])
    xy, resp, counts = np.zeros((n, nc, 2), np.float32), np.zeros((n, nc), np.float32), np.zeros(n, np.int32)
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = K.lib.ktlvo_detect_gftt(ctx._h, 0, arr, n, C.byref(gp), K._ptr(xy), K._ptr(resp), K._ptr(counts), nc)
    raw = (time.perf_counter() - t0) / reps
    assert rc == 0 and int(counts.sum()) == sum(len(x) for x, _ in out)
    print("%s %-42s %6d corners  %8.3f ms per call, %8.3f ms in ktlvo_detect_gftt itself" %
          (label, name, sum(len(x) for x, _ in out), wall * 1e3, raw * 1e3), flush=True)
    ctx.close()
